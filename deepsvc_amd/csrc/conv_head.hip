// conv_head.hip -- narrow output heads (Cout <= 4) of large f16 images as packed dot products on the vector ALU:
// the frame / flow / mask outputs of the reference (modules.py:127-131 ME_Spynet's last 7x7 conv 16 -> 2, 307-311 / 322-340 / 382-418
// the 3x3 convs 64 -> 3, 32 -> 3, 64 -> 2 and 64+64 -> 3 behind the reconstruction and refinement stacks).
//
// Why not the MFMA kernels: a 32-wide (or 16-wide) output tile computes 8-16x the needed columns, and with so few useful MFMAs per
// staged byte the two-group 3x3 kernel's HEAD variant was bound by its staging phases (127-139 us for 64 -> 3 @1088x1920, where the
// tensor itself streams from HBM in ~35 us; the 7x7 16 -> 2 conv took 336 us in the generic kernel: tools/conv_census.py).
// Here:
//   * one workgroup = one 8x32-pixel tile, one thread = one output pixel, all Cout accumulators in fp32 registers;
//   * the halo patch of ALL input channels is staged once into LDS with 16-byte bounds-checked buffer loads (padding reads zeros),
//     row pitch = channels * 2 + 16 bytes, so the per-pixel ds_read_b128 of 32 neighbouring lanes are conflict-free;
//   * the weights are read with SCALAR loads straight from the MFMA-packed buffer ([chunk][tap][Np rows] x 32 halfs: the rows of output
//     channels 0..Cout-1 of a (chunk, tap) are adjacent) and enter v_dot2_f32_f16 as the SGPR operand: no weight staging at all;
//   * 2 MACs per lane and instruction with fp32 accumulation (3x3 x 64 -> 3: 864 v_dot2 per pixel, ~50 us of VALU time per launch
//     at 1088x1920 -- with three workgroups per CU the staging of one overlaps the arithmetic of the others).
// Summation order differs from the MFMA kernels' (fp32 sums either way): f16-mode tolerance, like every choice between f16 kernels.
#include <climits>
#include "conv_common.h"

using namespace dsvc_conv;

namespace {

typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));

constexpr int HD_TH = 8, HD_TW = 32;

// KS: 3 or 7 (square, stride 1, pad KS/2).  CV0 / CV1: 16-byte vectors (8 channels) per pixel of input segment 0 / 1.  NCO: outputs.
template <int KS, int CV0, int CV1, int NCO>
__global__ void __launch_bounds__(256) conv_head_kernel(const ConvParams p) {
    DSVC_F16_SAT();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CVT = CV0 + CV1, PITCH = CVT * 16 + 16;
    constexpr int PH = HD_TH + KS - 1, PW = HD_TW + KS - 1, NPIX = PH * PW, NV = NPIX * CVT, NTAPS = KS * KS;
    static_assert((CVT & (CVT - 1)) == 0, "vectors per pixel: a power of two");
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx0 = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty0 = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int oy0 = ty0 * HD_TH, ox0 = tx0 * HD_TW;

    // ---- stage the patch: vector v -> patch pixel v / CVT, 16-byte piece v % CVT (8 lanes cover one pixel's 128 bytes) ----
    {
        const rsrc_t r0 = make_rsrc(p.seg[0].ptr, p.seg[0].bytes);
        const rsrc_t r1 = make_rsrc(p.seg[CV1 ? 1 : 0].ptr, p.seg[CV1 ? 1 : 0].bytes);
        const unsigned pitch0 = (unsigned)p.seg[0].ld * 2u, pitch1 = (unsigned)p.seg[CV1 ? 1 : 0].ld * 2u;
        const int iy0 = oy0 - KS / 2, ix0 = ox0 - KS / 2;
        constexpr int SB = 8;
        for (int base = tid; base < NV; base += 256 * SB) {
            uint4 v[SB];
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int vi = base + 256 * j;
                const int pix = vi / CVT, piece = vi % CVT;
                const int py = pix / PW, px = pix - py * PW;
                const int iy = iy0 + py, ix = ix0 + px;
                const bool inb = vi < NV && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                const unsigned lin = (unsigned)((b * p.H + iy) * p.W + ix);
                if (CV1 == 0 || piece < CV0) v[j] = buf_load16(r0, (lin * pitch0 + (unsigned)piece * 16u) | (inb ? 0u : OOB));
                else v[j] = buf_load16(r1, (lin * pitch1 + (unsigned)(piece - CV0) * 16u) | (inb ? 0u : OOB));
            }
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int vi = base + 256 * j;
                if (vi < NV) *(uint4*)(smem + (vi / CVT) * PITCH + (vi % CVT) * 16) = v[j];
            }
        }
    }
    __syncthreads();

    // ---- one output pixel per thread ----
    const int ty = tid >> 5, tx = tid & 31;
    float acc[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) acc[co] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    // the packed weights through the CONSTANT address space: uniform addresses there become s_load_dword (as plain global pointers the
    // compiler issued one vector load per weight: it cannot prove that no other wave's store aliases them)
    typedef const unsigned __attribute__((address_space(4))) cu32_t;
    cu32_t* wq = (cu32_t*)(uintptr_t)p.w;
    const int np16 = p.Np * 16;                                  // dwords per (chunk, tap): Np rows of 32 halfs
    const char* pbase = smem + (ty * PW + tx) * PITCH;
#pragma unroll 1
    for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int t = ky * KS + kx;
            const char* px = pbase + (ky * PW + kx) * PITCH;
#pragma unroll
            for (int piece = 0; piece < CVT; ++piece) {
                // segment s starts at chunk seg[s].chunk0; piece q of a segment = channels 8q .. 8q+7 = pairs 4*(q % 4) .. +3 of chunk q / 4
                const int q = piece < CV0 ? piece : piece - CV0;
                const int chunk = (piece < CV0 ? 0 : p.seg[CV1 ? 1 : 0].chunk0) + q / 4;
                cu32_t* wrow = wq + (size_t)(chunk * NTAPS + t) * np16 + (q % 4) * 4;
                const uint4 xv = *(const uint4*)(px + piece * 16);
                const unsigned xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int co = 0; co < NCO; ++co)
                        acc[co] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2_t, xs[j]), __builtin_bit_cast(h16x2_t, wrow[co * 16 + j]), acc[co], false);
            }
        }
    }

    // ---- epilogue: activation, optional residual (either type), element stores ----
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < p.Ho && ox < p.Wo) {
        const int64_t opix = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
#pragma unroll
        for (int co = 0; co < NCO; ++co)
            if (co < p.Cout) {
                float val = apply_act(acc[co], p.act, p.slope);
                if (p.res1) val += load_elem(p.res1, p.res1_dtype, opix * p.res1_ld + co);
                store_elem(p.out, p.out_dtype, opix * p.out_ld + co, val);
            }
    }
}

template <int KS, int CV0, int CV1, int NCO>
int launch_head(hipStream_t st, const ConvParams& p, int ntiles) {
    constexpr int CVT = CV0 + CV1;
    const size_t lds = (size_t)(HD_TH + KS - 1) * (HD_TW + KS - 1) * (CVT * 16 + 16);
    static bool attr_set = false;
    auto kern = conv_head_kernel<KS, CV0, CV1, NCO>;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { dsvc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return DSVC_E_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), lds, st, p);
    DSVC_CHECK_LAUNCH("conv_head_kernel");
    return DSVC_OK;
}

template <int KS, int CV0, int CV1>
int launch_head_nco(hipStream_t st, const ConvParams& p, int ntiles, int cout) {
    if (cout <= 2) return launch_head<KS, CV0, CV1, 2>(st, p, ntiles);
    if (cout == 3) return launch_head<KS, CV0, CV1, 3>(st, p, ntiles);
    return launch_head<KS, CV0, CV1, 4>(st, p, ntiles);
}

}  // namespace

// returns 1 when the shape is not covered (the caller goes on to the other kernels), else DSVC_OK / error
int dsvc_conv_head_try(hipStream_t st, const ConvParams& p_in, const dsvc_conv_desc* d) {
    if (d->compute != DSVC_F16 || d->kh != d->kw || (d->kh != 3 && d->kh != 7) || d->stride != 1 || d->pad != d->kh / 2 || d->shuffle != 1) return 1;
    if (d->Cout > 4 || p_in.nseg > 2 || d->res2 || d->res1_scale || d->pre_act != DSVC_ACT_NONE || d->post_weight || p_in.s2d) return 1;
    if (p_in.w_bs != 0 || p_in.out_bs != 0 || (p_in.b_inner != INT_MAX && d->B > 1)) return 1;        // dense batches with shared weights only
    for (int s = 0; s < p_in.nseg; ++s) {
        const ConvSeg& sg = p_in.seg[s];
        if (sg.dtype != DSVC_F16 || (sg.C % 8) != 0 || ((uintptr_t)sg.ptr & 15) != 0 || (sg.ld % 8) != 0) return 1;
        if ((int64_t)d->B * d->H * d->W * sg.ld * 2 >= (1ll << 31)) return 1;           // 32-bit byte offsets (pixel index * pitch)
    }
    const int c0 = p_in.seg[0].C, c1 = p_in.nseg > 1 ? p_in.seg[1].C : 0;
    ConvParams p = p_in;
    p.tiles_x = (p.Wo + HD_TW - 1) / HD_TW;
    p.tiles_y = (p.Ho + HD_TH - 1) / HD_TH;
    const int64_t nt = (int64_t)p.B * p.tiles_x * p.tiles_y;
    if (nt < 256 || nt > INT_MAX) return 1;        // small images stay on the MFMA kernels
    const int ntiles = (int)nt;
    // v_dot2c_f32_f16 issues at HALF rate (8 cycles per wave: 3x3 32 -> 3 = 432 dot products per pixel runs in 49.6 us, exactly that bound),
    // so the kernel wins where the MFMA kernels waste most: measured at 1088x1920 (tools/bench_head.py), dot products vs MFMA:
    // 7x7 16 -> 2: 73 vs 321 us; 3x3 32 -> 3: 50 vs 62; 3x3 64 -> 2 + residual: 96 vs 116; 3x3 64 -> 3: 120 vs 98; 64+64 -> 3 (one
    // 92 KB workgroup per CU): 438 vs 259.  Option "head" = 2 takes the losing shapes too (A/B runs).
    const bool all = g_dsvc_opt.head >= 2;
    if (d->kh == 3) {
        if (c1 == 0 && c0 == 64 && (d->Cout <= 2 || all)) return launch_head_nco<3, 8, 0>(st, p, ntiles, d->Cout);
        if (c1 == 0 && c0 == 32) return launch_head_nco<3, 4, 0>(st, p, ntiles, d->Cout);
        if (c1 == 64 && c0 == 64 && all) return launch_head_nco<3, 8, 8>(st, p, ntiles, d->Cout);
        return 1;
    }
    if (c1 == 0 && c0 == 16) return launch_head_nco<7, 2, 0>(st, p, ntiles, d->Cout);
    return 1;
}

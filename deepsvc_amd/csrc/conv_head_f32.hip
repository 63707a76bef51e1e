// conv_head_f32.hip -- fp32 (parity) mode's narrow output heads (Cout <= 4) of large images on the vector ALU: the frame / flow / mask
// outputs of the reference (modules.py:127-131 ME_Spynet's last 7x7 conv 16 -> 2; 307-311 / 322-340 / 382-418 the 3x3 convs 64 -> 3,
// 32 -> 3, 64 -> 2 and 64+64 -> 3 behind the refinement and reconstruction stacks).
//
// The MFMA kernel computes a 32-wide output tile for these: 8-16x the needed columns, 0.7 TB/s on a tensor that streams in ~150 us
// (tools/conv_time_census.py with CENSUS_MODE=fp32: 64 -> 3 takes 747 us at 1088x1920, the time of a 64 -> 32 conv).
//
// Bit-identical to that kernel all the same, because v_mfma_f32_32x32x2_f32 IS an fma chain -- acc = fma(a1, b1, fma(a0, b0, acc)), one
// rounding per product-add, k slot 0 first (tools/mfma_f32_order_probe.py: 352 of 352 outputs against exact arithmetic; the three other
// candidate orders match ~46 %) -- so v_fma_f32 in the same order reproduces it: per 16-channel chunk (segments in order), per tap
// (row-major), channels 0, 8, 1, 9, ... 7, 15 (conv_mfma_kernel<float>'s pairing of k slots), accumulators starting at zero, then the
// general epilogue's arithmetic (+ bias, activation, + residuals).  tests/test_gpu_ops.py::test_fp32_head_kernel_is_bit_identical.
//
// Structure: one workgroup = one 8x32-pixel tile, one thread = one output pixel with all Cout accumulators; the halo patch of ONE
// 16-channel chunk at a time in LDS (80-byte pixel pitch: conflict-free ds_read_b128), the next chunk's patch in registers while this
// chunk's taps run; weights by SCALAR loads straight from the MFMA-packed buffer
// ([chunk][tap][Np rows] x 16 floats) as the SGPR operand of v_fma_f32.
#include <climits>
#include "conv_common.h"

using namespace dsvc_conv;

namespace {

constexpr int HF_TH = 8, HF_TW = 32;

template <int ACT, int NCO>
__device__ __forceinline__ void hf_store(const ConvParams& p, const float (&acc)[NCO], int64_t opix) {
#pragma unroll
    for (int co = 0; co < NCO; ++co)
        if (co < p.Cout) {
            float v = act_t<ACT>(acc[co] + (p.bias ? p.bias[co] : 0.f), p.slope);
            if (p.res1) v += load_elem(p.res1, p.res1_dtype, opix * p.res1_ld + co);
            if (p.res2) v += load_elem(p.res2, p.res2_dtype, opix * p.res2_ld + co);
            store_elem(p.out, p.out_dtype, opix * p.out_ld + co, v);
        }
}

// KS: 3 or 7 (square, stride 1, pad KS / 2).  NCO: accumulators per pixel (>= Cout).
template <int KS, int NCO, int NB>
__global__ void __launch_bounds__(256) conv_head_f32_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PH = HF_TH + KS - 1, PW = HF_TW + KS - 1, NPIX = PH * PW, NV = NPIX * 4, NTAPS = KS * KS;
    constexpr int SB = (NV + 255) / 256;                              // 16-byte pieces per thread and chunk
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx0 = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty0 = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int oy0 = ty0 * HF_TH, ox0 = tx0 * HF_TW;
    const int iy0 = oy0 - KS / 2, ix0 = ox0 - KS / 2;

    uint4 pv[SB];
    auto issue = [&](int ch) {                                        // chunk ch of the virtual concatenation -> registers (padding reads zeros)
        int s = 0;
        if (p.nseg > 1 && ch >= p.seg[1].chunk0) s = 1;
        if (p.nseg > 2 && ch >= p.seg[2].chunk0) s = 2;
        const ConvSeg& sg = p.seg[s];
        const rsrc_t rs = make_rsrc(sg.ptr, sg.bytes);
        const unsigned pitch = (unsigned)sg.ld * 4u, c0b = (unsigned)(ch - sg.chunk0) * 64u;
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const int vi = tid + 256 * j;
            const int pix = vi >> 2, piece = vi & 3;
            const int py = pix / PW, px = pix - py * PW;
            const int iy = iy0 + py, ix = ix0 + px;
            const bool inb = vi < NV && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const unsigned lin = (unsigned)((b * p.H + iy) * p.W + ix);
            pv[j] = buf_load16(rs, (lin * pitch + c0b + (unsigned)piece * 16u) | (inb ? 0u : OOB));
        }
    };
    auto commit = [&](char* buf) {
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const int vi = tid + 256 * j;
            if (vi < NV) *(uint4*)(buf + (vi >> 2) * ROWB + (vi & 3) * 16) = pv[j];
        }
    };

    const int ty = tid >> 5, tx = tid & 31;
    float acc[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) acc[co] = 0.f;
    // the packed weights through the CONSTANT address space: uniform addresses there become s_load_dword
    typedef const float __attribute__((address_space(4))) cf32_t;
    cf32_t* wq = (cf32_t*)(uintptr_t)p.w;
    const int np16 = p.Np * 16;                                      // floats per (chunk, tap): Np rows of 16

    issue(0);
    commit(smem);
    for (int ch = 0; ch < p.nchunks; ++ch) {
        __syncthreads();                                            // chunk ch committed by everyone (two buffers: ... and everyone done with the other one)
        if (ch + 1 < p.nchunks) issue(ch + 1);
        const char* pbase = smem + (NB == 2 ? (ch & 1) * (NPIX * ROWB) : 0) + (ty * PW + tx) * ROWB;
#pragma unroll 1
        for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const char* px = pbase + (ky * PW + kx) * ROWB;
                cf32_t* wrow = wq + (size_t)(ch * NTAPS + ky * KS + kx) * np16;
                float x[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t = *(const float4*)(px + q * 16);
                    x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {                       // the MFMA's k slots: channel k, then channel k + 8
#pragma unroll
                    for (int co = 0; co < NCO; ++co) acc[co] = __builtin_fmaf(x[k], wrow[co * 16 + k], acc[co]);
#pragma unroll
                    for (int co = 0; co < NCO; ++co) acc[co] = __builtin_fmaf(x[k + 8], wrow[co * 16 + k + 8], acc[co]);
                }
            }
        }
        if constexpr (NB == 1) __syncthreads();                     // one buffer: everyone done reading it
        if (ch + 1 < p.nchunks) commit(smem + (NB == 2 ? ((ch + 1) & 1) * (NPIX * ROWB) : 0));
    }

    // ---- epilogue: the general kernel's arithmetic (conv_common.h: epilogue_pass) ----
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < p.Ho && ox < p.Wo) {
        const int64_t opix = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
        switch (p.act) {
            case DSVC_ACT_RELU: hf_store<DSVC_ACT_RELU>(p, acc, opix); break;
            case DSVC_ACT_LRELU: hf_store<DSVC_ACT_LRELU>(p, acc, opix); break;
            case DSVC_ACT_GELU: hf_store<DSVC_ACT_GELU>(p, acc, opix); break;
            case DSVC_ACT_SIGMOID: hf_store<DSVC_ACT_SIGMOID>(p, acc, opix); break;
            case DSVC_ACT_HALF_TANH: hf_store<DSVC_ACT_HALF_TANH>(p, acc, opix); break;
            default: hf_store<DSVC_ACT_NONE>(p, acc, opix); break;
        }
    }
}

template <int KS, int NCO>
int launch_hf(hipStream_t st, const ConvParams& p, int ntiles) {
    // one patch buffer (27 KB for 3x3: four workgroups per CU, bounded by registers) beats two (two workgroups per CU): 205 vs 270 us for
    // 64 -> 3, 145 vs 285 us for 7x7 16 -> 2 at 1088x1920 (tools/bench_head_f32.py) -- other workgroups hide the second barrier
    constexpr int NB = 1;
    const size_t lds = (size_t)NB * (HF_TH + KS - 1) * (HF_TW + KS - 1) * ROWB;
    static bool attr_set = false;
    auto kern = conv_head_f32_kernel<KS, NCO, NB>;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { dsvc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return DSVC_E_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), lds, st, p);
    DSVC_CHECK_LAUNCH("conv_head_f32_kernel");
    return DSVC_OK;
}

template <int KS>
int launch_hf_nco(hipStream_t st, const ConvParams& p, int ntiles, int cout) {
    if (cout <= 2) return launch_hf<KS, 2>(st, p, ntiles);
    if (cout == 3) return launch_hf<KS, 3>(st, p, ntiles);
    return launch_hf<KS, 4>(st, p, ntiles);
}

}  // namespace

// returns 1 when the shape is not covered (the caller goes on to the other kernels), else DSVC_OK / error
int dsvc_conv_head_f32_try(hipStream_t st, const ConvParams& p_in, const dsvc_conv_desc* d) {
    if (d->compute != DSVC_F32 || d->kh != d->kw || (d->kh != 3 && d->kh != 7) || d->stride != 1 || d->pad != d->kh / 2 || d->shuffle != 1) return 1;
    if (d->Cout > 4 || d->res1_scale || d->pre_act != DSVC_ACT_NONE || d->post_weight || p_in.s2d) return 1;
    if (p_in.w_bs != 0 || p_in.out_bs != 0 || (p_in.b_inner != INT_MAX && d->B > 1)) return 1;        // dense batches with shared weights only
    for (int s = 0; s < p_in.nseg; ++s) {
        const ConvSeg& sg = p_in.seg[s];
        if (sg.dtype != DSVC_F32 || (sg.C % 16) != 0 || ((uintptr_t)sg.ptr & 15) != 0 || (sg.ld % 4) != 0) return 1;   // whole chunks of aligned rows
        if ((int64_t)d->B * d->H * d->W * sg.ld * 4 >= (1ll << 31)) return 1;           // 32-bit byte offsets (pixel index * pitch)
    }
    ConvParams p = p_in;
    p.tiles_x = (p.Wo + HF_TW - 1) / HF_TW;
    p.tiles_y = (p.Ho + HF_TH - 1) / HF_TH;
    const int64_t nt = (int64_t)p.B * p.tiles_x * p.tiles_y;
    if (nt < 256 || nt > INT_MAX) return 1;        // small images stay on the MFMA kernel
    if (d->kh == 3) return launch_hf_nco<3>(st, p, (int)nt, d->Cout);
    return launch_hf_nco<7>(st, p, (int)nt, d->Cout);
}

// conv_headtap.hip -- 3x3 output heads with Cout <= 3 of large f16 images on the matrix pipe WITHOUT the padding waste: the frame / flow / mask
// outputs of the reference (modules.py:307-311 / 322-340 / 382-418: the 3x3 convs 64 -> 3, 32 -> 3, 64 -> 2 and 64+64 -> 3 behind the
// reconstruction and refinement stacks).
//
// A 3x3 conv with three outputs is 27 dot products per input pixel: out[y][x][co] = sum_t  < in[y+dy_t][x+dx_t][:], W[t][:][co] >.  Computed per
// OUTPUT pixel on the MFMA kernels it fills 3 of a 32-wide tile's columns (the two-group kernel's HEAD variant: 98-144 us for 64 -> 3 at
// 1088x1920, where the tensor streams from HBM in ~50 us); per output pixel on the vector ALU (conv_head.hip) it is 864 half-rate v_dot2 per
// pixel (120 us).  Here the products are taken per INPUT pixel: one 32x32x16 MFMA tile = 32 patch pixels x (tap, output) columns
// t * Cout + co (27 of 32 columns used), i.e. a pointwise GEMM [pixels x C] x [C x 9 Cout] over the halo patch -- 4 MFMAs per 32 pixels and
// 64 channels --, and the 3x3 structure becomes a 9-point gather-sum over fp32 partial products in LDS:
//   1. the halo patch of a segment's channels is staged with 16-byte bounds-checked buffer loads, 8 lanes per 128-byte pixel row (padding
//      reads zeros), row pitch = channels * 2 + 16 bytes: the per-lane ds_read_b128 of the A fragments are conflict-free (conv_head.hip's staging);
//   2. the B fragments (lane (r, h): column r = (tap, output), channels 16 ks + 8 h .. + 7) are 16-byte loads straight from the MFMA-packed
//      weight buffer ([chunk][tap][Np rows] x 32 halfs) into registers, once per workgroup: no weight staging;
//   3. wave w multiplies the patch's M-tiles w, w + 4, w + 8 (the patch is a flat list of (8 + 2) x (32 + 2) = 340 pixels = 11 M-tiles);
//      a second input segment (64 + 64 -> 3) is staged into the same patch buffer and accumulated into the same registers;
//   4. the partial products P[pixel][column] go to LDS (over the patch, behind a barrier; pitch 33 floats: conflict-free both ways) and every
//      thread sums the 9 x Cout values of its output pixel, adds bias / activation / residual and stores.
// Workgroups are persistent (three per CU, XCD-aware tile walk) and prefetch the next stage's patch vectors into registers.
// Summation order differs from the other f16 kernels' (channels first, then taps; fp32 sums either way): f16-mode tolerance, like every choice
// between f16 kernels; deterministic.
#include <climits>
#include "conv_common.h"

using namespace dsvc_conv;

namespace {

constexpr int HT_TH = 8, HT_TW = 32, HT_PH = HT_TH + 2, HT_PW = HT_TW + 2, HT_NPIX = HT_PH * HT_PW;
constexpr int HT_NMT = (HT_NPIX + 31) / 32, HT_NPIXP = HT_NMT * 32, HT_PP = 33;
constexpr int HT_MTW = (HT_NMT + 3) / 4;                 // M-tiles per wave

// CV: 16-byte vectors (8 channels) per pixel of EACH input segment (4: 32 channels, 8: 64).  NSEG: 1 or 2 segments.  NCO: outputs (2 or 3).
// Persistent workgroups (three per CU): the patch vectors of the NEXT stage -- the tile's second segment, or the next tile's first -- are loaded
// into registers right after the current stage's vectors have been written to LDS, so they fly during the MFMA, exchange and gather phases
// (one-tile-per-workgroup launches ran at 3.5 TB/s of fetched bytes: every phase after the barrier had no load in flight).
template <int CV, int NSEG, int NCO>
__global__ void __launch_bounds__(256, NSEG == 1 ? 3 : 2) conv_headtap_kernel(const ConvParams p, int ntiles) {
    DSVC_F16_SAT();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PITCH = CV * 16 + 16, KS = CV / 2, NV = HT_NPIX * CV, NCOL = 9 * NCO, NLD = (NV + 255) / 256;
    static_assert(NCOL <= 32 && (CV == 4 || CV == 8), "one N-tile of (tap, output) columns; whole 32-channel chunks");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware walk: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the blocks with equal blockIdx % 8 share an
    // L2.  Each such group takes one contiguous row-major range of tiles: vertically adjacent tiles -- whose patches share 2 of 10 rows -- run
    // on the same XCD a tile row apart in time, and the halo re-reads (a third of the bytes a tile stages) can hit in L2.
    const int xq = ntiles >> 3, xrem = ntiles & 7;
    auto tile_of = [&](int vb, int& b, int& oy0, int& ox0) {      // virtual block id (vb % 8 = this workgroup's XCD label) -> tile
        const int xcd = vb & 7;
        int t = (xcd < xrem ? xcd * (xq + 1) : xrem * (xq + 1) + (xcd - xrem) * xq) + (vb >> 3);
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        b = t / p.tiles_y; oy0 = ty * HT_TH; ox0 = tx * HT_TW;
    };
    const rsrc_t rs0 = make_rsrc(p.seg[0].ptr, p.seg[0].bytes), rs1 = make_rsrc(p.seg[NSEG - 1].ptr, p.seg[NSEG - 1].bytes);
    const unsigned pitch0 = (unsigned)p.seg[0].ld * 2u, pitch1 = (unsigned)p.seg[NSEG - 1].ld * 2u;
    // patch vector v = tid + 256 j -> patch pixel v / CV, 16-byte piece v % CV (8 or 4 lanes cover one pixel's row; padding reads zeros)
    auto issue = [&](int vb, int s, uint4 (&v)[NLD]) {
        int b, oy0, ox0;
        tile_of(vb < ntiles ? vb : 0, b, oy0, ox0);
        const bool live = vb < ntiles;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int vi = tid + 256 * j;
            const int pix = vi / CV, piece = vi % CV;
            const int py = pix / HT_PW, px = pix - py * HT_PW;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            const bool inb = live && vi < NV && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const unsigned lin = (unsigned)((b * p.H + iy) * p.W + ix);
            v[j] = (NSEG == 1 || s == 0) ? buf_load16(rs0, (lin * pitch0 + (unsigned)piece * 16u) | (inb ? 0u : OOB))
                                         : buf_load16(rs1, (lin * pitch1 + (unsigned)piece * 16u) | (inb ? 0u : OOB));
        }
    };
    // ---- B fragments: column r = t * NCO + co <- packed row (chunk, t, co), halfs (ks & 1) * 16 + 8 h .. + 7; once per workgroup ----
    uint4 bw[NSEG][KS];
    {
        const int t = r / NCO, co = r - t * NCO;
        const bool colok = r < NCOL && co < p.Cout;
#pragma unroll
        for (int s = 0; s < NSEG; ++s)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int chunk = p.seg[s].chunk0 + (ks >> 1);
                const size_t off = ((size_t)(chunk * 9 + t) * p.Np + co) * 64 + (ks & 1) * 32 + h * 16;
                bw[s][ks] = colok ? *(const uint4*)((const char*)p.w + off) : make_uint4(0u, 0u, 0u, 0u);
            }
    }
    float bias[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) bias[co] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    float* P = (float*)smem;
    const int ty = tid >> 5, tx = tid & 31;

    uint4 v[NLD];
    issue((int)blockIdx.x, 0, v);
    for (int vb = (int)blockIdx.x; vb < ntiles; vb += (int)gridDim.x) {
        f32x16 acc[HT_MTW];
#pragma unroll
        for (int j = 0; j < HT_MTW; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int vi = tid + 256 * j;
                if (vi < NV) *(uint4*)(smem + (vi / CV) * PITCH + (vi % CV) * 16) = v[j];
            }
            lds_barrier();
            if (s + 1 < NSEG) issue(vb, s + 1, v);
            else issue(vb + (int)gridDim.x, 0, v);
            // ---- partial products of this wave's M-tiles (patch pixels 340 .. 351 of the last tile are never gathered) ----
#pragma unroll
            for (int j = 0; j < HT_MTW; ++j) {
                const int mt = wave + 4 * j;
                if (mt < HT_NMT) {
                    const char* arow = smem + (mt * 32 + r) * PITCH + h * 16;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const uint4 a = *(const uint4*)(arow + ks * 32);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, bw[s][ks]), acc[j], 0, 0, 0);
                    }
                }
            }
            lds_barrier();                                 // every wave has read this segment's patch: the next segment / the partial products replace it
        }
#pragma unroll
        for (int j = 0; j < HT_MTW; ++j) {
            const int mt = wave + 4 * j;
            if (mt < HT_NMT) {
#pragma unroll
                for (int i = 0; i < 16; ++i) P[(mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * HT_PP + r] = acc[j][i];     // accumulator row = patch pixel, lane r = column
            }
        }
        lds_barrier();
        // ---- one output pixel per thread: the 9-point gather-sum, then activation, optional residual (either type), element stores ----
        float o[NCO];
#pragma unroll
        for (int co = 0; co < NCO; ++co) o[co] = bias[co];
        const float* pb = P + (ty * HT_PW + tx) * HT_PP;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* q = pb + (ky * HT_PW + kx) * HT_PP + (ky * 3 + kx) * NCO;
#pragma unroll
                for (int co = 0; co < NCO; ++co) o[co] += q[co];
            }
        lds_barrier();                                     // the next tile's patch replaces the partial products
        int b, oy0, ox0;
        tile_of(vb, b, oy0, ox0);
        const int oy = oy0 + ty, ox = ox0 + tx;
        if (oy < p.Ho && ox < p.Wo) {
            const int64_t opix = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
#pragma unroll
            for (int co = 0; co < NCO; ++co)
                if (co < p.Cout) {
                    float val = apply_act(o[co], p.act, p.slope);
                    if (p.res1) val += load_elem(p.res1, p.res1_dtype, opix * p.res1_ld + co);
                    store_elem(p.out, p.out_dtype, opix * p.out_ld + co, val);
                }
        }
    }
}

template <int CV, int NSEG, int NCO>
int launch_headtap(hipStream_t st, const ConvParams& p, int ntiles) {
    constexpr size_t patch = (size_t)HT_NPIXP * (CV * 16 + 16), part = (size_t)HT_NPIXP * HT_PP * sizeof(float);
    constexpr size_t lds = patch > part ? patch : part;     // 50.7 KB (64 channels per segment) / 46.5 KB: three workgroups per CU
    constexpr int WGCU = (CV == 8 && NSEG == 2) ? 2 : 3;     // resident workgroups per CU: LDS allows three, the 64 + 64 variant's registers two
    const int grid = ntiles < 256 * WGCU ? ntiles : 256 * WGCU;
    hipLaunchKernelGGL((conv_headtap_kernel<CV, NSEG, NCO>), dim3(grid), dim3(256), lds, st, p, ntiles);
    DSVC_CHECK_LAUNCH("conv_headtap_kernel");
    return DSVC_OK;
}

template <int CV, int NSEG>
int launch_headtap_nco(hipStream_t st, const ConvParams& p, int ntiles, int cout) {
    if (cout <= 2) return launch_headtap<CV, NSEG, 2>(st, p, ntiles);
    return launch_headtap<CV, NSEG, 3>(st, p, ntiles);
}

}  // namespace

// returns 1 when the shape is not covered (the caller goes on to the other kernels), else DSVC_OK / error
int dsvc_conv_headtap_try(hipStream_t st, const ConvParams& p_in, const dsvc_conv_desc* d) {
    if (d->compute != DSVC_F16 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->shuffle != 1) return 1;
    if (d->Cout > 3 || p_in.nseg > 2 || d->res2 || d->res1_scale || d->pre_act != DSVC_ACT_NONE || d->post_weight || p_in.s2d) return 1;
    if (p_in.w_bs != 0 || p_in.out_bs != 0 || (p_in.b_inner != INT_MAX && d->B > 1)) return 1;        // dense batches with shared weights only
    for (int s = 0; s < p_in.nseg; ++s) {
        const ConvSeg& sg = p_in.seg[s];
        if (sg.dtype != DSVC_F16 || (sg.C != 32 && sg.C != 64) || sg.C != p_in.seg[0].C || ((uintptr_t)sg.ptr & 15) != 0 || (sg.ld % 8) != 0) return 1;
        if ((int64_t)d->B * d->H * d->W * sg.ld * 2 >= (1ll << 31)) return 1;           // 32-bit byte offsets (pixel index * pitch)
    }
    ConvParams p = p_in;
    p.tiles_x = (p.Wo + HT_TW - 1) / HT_TW;
    p.tiles_y = (p.Ho + HT_TH - 1) / HT_TH;
    const int64_t nt = (int64_t)p.B * p.tiles_x * p.tiles_y;
    if (nt < 256 || nt > INT_MAX) return 1;        // small images stay on the MFMA kernels
    const int ntiles = (int)nt;
    const int c0 = p.seg[0].C;
    if (p.nseg == 1) return c0 == 64 ? launch_headtap_nco<8, 1>(st, p, ntiles, d->Cout) : launch_headtap_nco<4, 1>(st, p, ntiles, d->Cout);
    return c0 == 64 ? launch_headtap_nco<8, 2>(st, p, ntiles, d->Cout) : launch_headtap_nco<4, 2>(st, p, ntiles, d->Cout);
}

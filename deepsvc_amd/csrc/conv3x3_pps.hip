// conv3x3_pps.hip -- the two-group ("ping-pong") f16 3x3 stride-1 convolution of conv3x3_pp.hip for MORE THAN TWO 32-channel input
// chunks and 64 output channels, with the weights STREAMED through LDS chunk by chunk instead of resident:
//   modules.py:401 (fea_embd[0]: cat[fea, cur] 64+64 -> 64), modules.py:262-277 (UNet.up_conv3: cat[x2, up3(x3)] 64+64 -> 64 + LeakyReLU),
//   modules.py:322 (Reconstruction.first_conv: cat[res_fea, warp_fea] 64+32 -> 64), modules.py:307 (RefineNet.refine[0]: cat[x, ref] 64+3 -> 64).
// These shapes ran on the general kernel (conv_mfma.hip) at 0.23-0.29 of the f16 MFMA peak (profiles/r03_conv_census_f16.txt: 64+64 -> 64
// @1088x1920 510 us): its workgroup is a serial stage -> multiply -> store chain per chunk and tap group (phase probe: MFMA 30 %, weight
// staging 26 %, patch 15 %, epilogue 15 %, barriers 14 % of a workgroup's clocks).  All weights of a 4-chunk conv (184 KB at the 80-byte
// pitch) do not fit beside two halo patches, but TWO chunks do (2 x 46 080 + 2 x 27 200 B = the footprint of conv3x3_pp.hip), and the
// two-group schedule leaves exactly the slots in which the next chunk's weights can be written while nobody reads that buffer:
//
//   stream index s = tile * NCH + chunk (one M phase each); weights(s) live in LDS buffer s & 1; group 1 runs one phase behind group 0
//
//     global phase     2s          2s+1        2s+2        2s+3
//     group 0          M(s)        S(s+1)      M(s+1)      S(s+2)          M(s): 72 MFMAs on patch(s) x weights(s)
//     group 1          S(s)        M(s)        S(s+1)      M(s+1)          S(u): write the prefetched patch(u) to the group's LDS patch,
//                                                                                [after the last chunk of a tile: its epilogue first]
//   buffer (s+1)&1 held weights(s-1), last read by group 1 in global phase 2s-1, and is next read (as weights(s+1)) by group 0 in global
//   phase 2s+2: it is free during global phases 2s and 2s+1, in which group 1 (its S phase after M(s-1)) and group 0 (its S phase after
//   M(s)) each write HALF of weights(s+1).  In its S phase after M(t), group g therefore commits half g of weights(t+1+g) -- loaded into
//   registers one S phase earlier (L2 hits) -- and requests half g of weights(t+2+g).  Patches are prefetched THREE stream steps ahead
//   (two register buffers, as in conv3x3_pp.hip: a load issued one M phase ahead still stalled its consumer).
// Everything else -- tile shape 8x32, 80-byte LDS pitch, fragment pipeline, swapped operand roles (A = weights, B = pixels), the LDS-free
// epilogue with two swap levels, the XCD-banded tile walk, wave priority around the MFMA phase -- is conv3x3_pp.hip's, and so is the
// arithmetic: per output element the same MFMAs in the same order as the general kernel (chunk, tap, 16-deep k-step), bias added
// after the accumulation as there, so results are bit-identical to it (tests/test_gpu_ops.py::test_streamed_weights_pp_kernel_matches_general_kernel).
#include <cstdlib>
#include "conv_common.h"

using namespace dsvc_conv;

namespace {

constexpr int PS_TH = 8, PS_PW = 34, PS_NPATCH = (PS_TH + 2) * PS_PW;   // 340 patch pixels
constexpr int PS_PFV = (PS_NPATCH * 4 + 255) / 256;                     // 6 patch vectors per thread
constexpr int PS_NTN = 64, PS_WROWS = 9 * PS_NTN;                       // rows (tap-major) of one chunk's weights
constexpr int PS_WBYTES = PS_WROWS * ROWB;                              // 46 080 B per LDS weight buffer
constexpr int PS_HALF = PS_WROWS / 2;                                   // rows a group stages per chunk
constexpr int PS_WV = (PS_HALF * 4 + 255) / 256;                        // 5 weight vectors per thread and half chunk

__device__ __forceinline__ void ps_store16(rsrc_t r, unsigned off, uint4 v) {
    v4u_t t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(t, r, off, 0, 0);
}
__device__ __forceinline__ float ps_lrelu(float x, float slope) { return __builtin_amdgcn_fmed3f(x, x * slope, __builtin_inff()); }

template <bool ACT>
__global__ void __launch_bounds__(512, 2) conv3x3_pps_kernel(const ConvParams p, int ntiles, int iters, float act_slope) {
    DSVC_F16_SAT();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MTW = 2, NT = 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wg = wave & 3, gtid = tid & 255;      // group, wave in group, thread in group
    const int r = lane & 31, h = lane >> 5;
    char* lds_w = smem;                                                                   // [2][PS_WROWS] rows of 80 B
    char* lds_a = smem + 2 * PS_WBYTES + g * (PS_NPATCH * ROWB);                           // this group's patch
    float* lds_bias = (float*)(smem + 2 * PS_WBYTES + 2 * PS_NPATCH * ROWB);               // 64 floats
    const rsrc_t rout = make_rsrc(p.out, p.out_bytes);
    const int NCH = p.nchunks;
    const rsrc_t rw = make_rsrc(p.w, (unsigned)NCH * PS_WROWS * 64u);

    const int H = p.H, W = p.W, tiles_x = p.tiles_x, tiles_y = p.tiles_y;
    struct TileOrg { int b, oy0, ox0; };
    auto tile_origin = [&](int tile) -> TileOrg {
        const int tx = tile % tiles_x;
        const int rest = tile / tiles_x;
        const int ty = rest % tiles_y;
        return TileOrg{rest / tiles_y, ty * PS_TH, tx * 32};
    };
    // ---- patch staging (as conv3x3_pp.hip): vector v = gtid + 256 j -> patch pixel v >> 2, 16-byte quarter v & 3 ----
    const int q = gtid & 3;
    int ppy[PS_PFV], ppx[PS_PFV], pp_pix[PS_PFV];
#pragma unroll
    for (int j = 0; j < PS_PFV; ++j) {
        const int pix = (gtid >> 2) + 64 * j;
        ppy[j] = pix / PS_PW; ppx[j] = pix - ppy[j] * PS_PW;
        pp_pix[j] = ppy[j] * W + ppx[j];
    }
    const bool last_ok = (gtid >> 2) + 64 * (PS_PFV - 1) < PS_NPATCH;
    auto load_chunk = [&](const TileOrg& t, int ch, uint4 (&pf)[PS_PFV]) {
        int si = 0;
        if (p.nseg > 1 && ch >= p.seg[1].chunk0) si = 1;
        if (p.nseg > 2 && ch >= p.seg[2].chunk0) si = 2;
        const ConvSeg& sg = p.seg[si];
        const rsrc_t rin = make_rsrc(sg.ptr, sg.bytes);
        const unsigned ldb = (unsigned)sg.ld * 2, cb = (unsigned)(ch - sg.chunk0) * 64 + q * 16;
        const bool cok = (ch - sg.chunk0) * 32 + q * 8 < sg.C;             // short chunk (C % 8 == 0): the tail reads zeros
        const int base_pix = (t.b * H + t.oy0 - 1) * W + t.ox0 - 1;
#pragma unroll
        for (int j = 0; j < PS_PFV; ++j) {
            const unsigned iy = (unsigned)(t.oy0 - 1 + ppy[j]), ix = (unsigned)(t.ox0 - 1 + ppx[j]);
            const bool inb = iy < (unsigned)H && ix < (unsigned)W && (j < PS_PFV - 1 || last_ok) && cok;
            pf[j] = buf_load16(rin, (__umul24((unsigned)(base_pix + pp_pix[j]) & 0xFFFFFFu, ldb) + cb) | (inb ? 0u : OOB));
        }
    };
    auto store_chunk = [&](uint4 (&pf)[PS_PFV]) {
#pragma unroll
        for (int j = 0; j < PS_PFV; ++j) {
            const int pix = (gtid >> 2) + 64 * j;
            if (pix >= PS_NPATCH) continue;
            *(uint4*)(lds_a + pix * ROWB + q * 16) = pf[j];
        }
    };
    // ---- weight staging: half `g` of a chunk = rows [288 g, 288 g + 288), vector v = gtid + 256 j -> row v >> 2, quarter v & 3 ----
    auto load_whalf = [&](int ch, uint4 (&wv)[PS_WV]) {
        const unsigned base = ((unsigned)ch * PS_WROWS + (unsigned)g * PS_HALF) * 64u + q * 16u;
#pragma unroll
        for (int j = 0; j < PS_WV; ++j) {
            const unsigned row = (unsigned)(gtid >> 2) + 64u * j;
            wv[j] = buf_load16(rw, (base + row * 64u) | (row < (unsigned)PS_HALF ? 0u : OOB));
        }
    };
    auto store_whalf = [&](int buf, uint4 (&wv)[PS_WV]) {
        char* dst = lds_w + buf * PS_WBYTES + g * PS_HALF * ROWB + q * 16;
#pragma unroll
        for (int j = 0; j < PS_WV; ++j) {
            const int row = (gtid >> 2) + 64 * j;
            if (row < PS_HALF) *(uint4*)(dst + row * ROWB) = wv[j];
        }
    };

    int a_off[MTW], b_off[NT];
#pragma unroll
    for (int m = 0; m < MTW; ++m) a_off[m] = ((wg * MTW + m) * PS_PW + r) * ROWB + h * 32;
#pragma unroll
    for (int n = 0; n < NT; ++n) b_off[n] = (n * 32 + r) * ROWB + h * 32;

    f32x16 acc[MTW][NT];
    auto mfma_phase = [&](int wsel) {          // 18 k-steps (9 taps x 2 halves of the 64-byte chunk), fragments read two k-steps ahead
        const char* wt = lds_w + wsel * PS_WBYTES;
        auto read_frags = [&](int s, uint4 (&af)[MTW], uint4 (&bf)[NT]) {
            const int t = s >> 1, ks = s & 1;
            const int tap_off = ((t / 3) * PS_PW + (t % 3)) * ROWB;
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = *(const uint4*)(wt + t * PS_NTN * ROWB + b_off[n] + ks * 16);
#pragma unroll
            for (int m = 0; m < MTW; ++m) af[m] = *(const uint4*)(lds_a + a_off[m] + tap_off + ks * 16);
        };
        uint4 af[3][MTW], bf[3][NT];
        read_frags(0, af[0], bf[0]);
        read_frags(1, af[1], bf[1]);
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int s2 = 0; s2 < 18; ++s2) {
            if (s2 + 2 < 18) read_frags(s2 + 2, af[(s2 + 2) % 3], bf[(s2 + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, bf[s2 % 3][n]),   // D[channel][pixel]
                                                                         __builtin_bit_cast(h16x8, af[s2 % 3][m]), acc[m][n], 0, 0, 0);
            asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));   // pin the accumulation order (conv3x3_pp.hip)
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto init_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    };
    // epilogue lane roles: see conv3x3_pp.hip (row_off / PP_TILE): after the two swap levels lane l of vector `half` holds pixel
    // (l & 15) + 16 half of the tile row and channels 16 ((l >> 4) & 1) + 8 (l >> 5) .. +7 of the N-tile
    auto row_off = [&](const TileOrg& t, int m, bool valid, int half) -> unsigned {
        const int oy = t.oy0 + wg * MTW + m;
        const int px = (r & 15) + 16 * half;
        const unsigned pix = (unsigned)((t.b * p.Ho + oy) * p.Wo + t.ox0) + px;
        return (__umul24(pix & 0xFFFFFFu, (unsigned)p.out_ld * 2) + 32 * (r >> 4) + 16 * h) | ((valid && oy < p.Ho && t.ox0 + px < p.Wo) ? 0u : OOB);
    };
#define PS_SWAP32(X, Y, LO, HI)                                                                                     \
    { const float xq = (X), yq = (Y);   /* scalars first: hipcc 7.2 mis-casts vector elements */                    \
      const v2u_t sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false); \
      const unsigned lo_ = sw.x, hi_ = sw.y; (LO) = __builtin_bit_cast(float, lo_); (HI) = __builtin_bit_cast(float, hi_); }
#define PS_SWAP16(X, Y, A, B)                                                                                       \
    { const v2u_t sw = __builtin_amdgcn_permlane16_swap((X), (Y), false, false); (A) = sw.x; (B) = sw.y; }
#define PS_TILE(M, N)                                                                                               \
    {                                                                                                               \
        float v0[8], v1[8];                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
            PS_SWAP32(acc[M][N][j], acc[M][N][4 + j], v0[j], v0[4 + j])                                             \
            PS_SWAP32(acc[M][N][8 + j], acc[M][N][12 + j], v1[j], v1[4 + j])                                        \
        }                                                                                                           \
        {   /* + bias AFTER the accumulation, as the general kernel does (bit-identical sums): v0 = channels 8h .., v1 = 16 + 8h .. of the N-tile */ \
            const float4 ba = *(const float4*)(lds_bias + N * 32 + 8 * h), bb = *(const float4*)(lds_bias + N * 32 + 8 * h + 4);          \
            const float4 bc = *(const float4*)(lds_bias + N * 32 + 16 + 8 * h), bd = *(const float4*)(lds_bias + N * 32 + 20 + 8 * h);    \
            v0[0] += ba.x; v0[1] += ba.y; v0[2] += ba.z; v0[3] += ba.w; v0[4] += bb.x; v0[5] += bb.y; v0[6] += bb.z; v0[7] += bb.w;       \
            v1[0] += bc.x; v1[1] += bc.y; v1[2] += bc.z; v1[3] += bc.w; v1[4] += bd.x; v1[5] += bd.y; v1[6] += bd.z; v1[7] += bd.w;       \
        }                                                                                                           \
        if constexpr (ACT) {                                                                                        \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) { v0[j] = ps_lrelu(v0[j], act_slope); v1[j] = ps_lrelu(v1[j], act_slope); } \
        }                                                                                                           \
        unsigned s0[4], s1[4];                                                                                      \
        _Pragma("unroll") for (int d = 0; d < 4; ++d)                                                               \
            PS_SWAP16(pk_h16(v0[2 * d], v0[2 * d + 1]), pk_h16(v1[2 * d], v1[2 * d + 1]), s0[d], s1[d])             \
        ps_store16(rout, o_out[M][0] + 64 * N, make_uint4(s0[0], s0[1], s0[2], s0[3]));                             \
        ps_store16(rout, o_out[M][1] + 64 * N, make_uint4(s1[0], s1[1], s1[2], s1[3]));                             \
    }

    // ---- XCD-aware tile walk (conv3x3_pp.hip) ----
    const int nband = (gridDim.x % 8 == 0) ? 8 : 1, per = (int)gridDim.x / nband;
    const int band = (int)blockIdx.x % nband, band_tiles = (ntiles + nband - 1) / nband;
    const int lt0 = ((int)blockIdx.x / nband) * 2 + g;
    auto tile_of = [&](int k) {            // tile index, or ntiles when this (group, k) has no tile
        const int lt = lt0 + k * 2 * per;
        const int t = band * band_tiles + lt;
        return (lt < band_tiles && t < ntiles) ? t : ntiles;
    };
    auto org_of = [&](int k) { return tile_origin(min(tile_of(k), ntiles - 1)); };

    // ---- prologue ----
    // stream position of the patch LOAD pointer: (lk, lc) = tile iteration and chunk of the next patch to request
    uint4 pfA[PS_PFV], pfB[PS_PFV], wv[PS_WV];
    TileOrg cur = org_of(0);
    int lk = 0, lc = 0;
    TileOrg lorg = cur;
    auto advance_load = [&]() { if (++lc == NCH) { lc = 0; ++lk; lorg = org_of(lk); } };
    load_chunk(lorg, lc, pfA); advance_load();            // patch(0)
    load_chunk(lorg, lc, pfB); advance_load();            // patch(1)
    {   // weights(0): all of it, by everyone; group 1 also stages its half of weights(1) (what it would have written "after M(-1)")
        constexpr int NWV = (PS_WROWS * 4 + 511) / 512;     // 5 vectors per thread over 512 threads
        const int q4 = tid & 3;
        uint4 w0[NWV];
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const unsigned row = (unsigned)(tid >> 2) + 128u * j;
            w0[j] = buf_load16(rw, (row * 64u + q4 * 16u) | (row < (unsigned)PS_WROWS ? 0u : OOB));
        }
        load_whalf(1 % NCH, wv);
        if (tid < PS_NTN) lds_bias[tid] = (p.bias && tid < p.Cout) ? p.bias[tid] : 0.f;
        store_chunk(pfA);                                  // the patch loads were issued first and return first
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const int row = (tid >> 2) + 128 * j;
            if (row < PS_WROWS) *(uint4*)(lds_w + row * ROWB + q4 * 16) = w0[j];
        }
        if (g == 1) store_whalf(1, wv);                    // (wave-uniform branch; group 0 keeps its half of weights(1) in wv)
    }
    load_chunk(lorg, lc, pfA); advance_load();            // patch(2)
    // weight stream position of this group: after M(t) it commits half g of weights(t + 1 + g) and requests weights(t + 2 + g)
    int wc = (1 + g) % NCH;                                // chunk of the half held in wv (group 0) / to hold next
    if (g == 1) { wc = 2 % NCH; load_whalf(wc, wv); }      // group 1: wv <- its half of weights(2)
    lds_barrier();                       // weights(0), bias, group 1's half of weights(1) and both groups' first patches are in place
    if (g) lds_barrier();                // group 1 runs one phase behind group 0

    int c = 0, k = 0;
    const int steps = iters * NCH;
    // one stream step; PFX = the register buffer that holds patch(t + 1) (and is refilled with patch(t + 3))
    auto step = [&](int t, uint4 (&pf_next)[PS_PFV]) {
        if (c == 0) init_acc();
        mfma_phase(t & 1);
        lds_barrier();
        if (c == NCH - 1) {              // ---- epilogue of the finished tile (wave-uniform branch) ----
            const bool valid = tile_of(k) < ntiles;
            unsigned o_out[MTW][2];
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int half = 0; half < 2; ++half) o_out[m][half] = row_off(cur, m, valid, half);
            PS_TILE(0, 0) PS_TILE(0, 1) PS_TILE(1, 0) PS_TILE(1, 1)
            c = 0; ++k;
            cur = org_of(k);
        } else {
            ++c;
        }
        // ---- S: patch(t + 1) into the group's LDS patch, request patch(t + 3); commit / request this group's weight halves ----
        store_chunk(pf_next);
        load_chunk(lorg, lc, pf_next); advance_load();
        store_whalf((t + 1 + g) & 1, wv);
        if (++wc == NCH) wc = 0;
        load_whalf(wc, wv);
        lds_barrier();
    };
    for (int t = 0; t < steps; t += 2) {
        step(t, pfB);
        if (t + 1 < steps) step(t + 1, pfA);
    }
    if (!g) lds_barrier();               // matches group 1's leading barrier
#undef PS_TILE
#undef PS_SWAP16
#undef PS_SWAP32
}

template <bool ACT>
int launch_pps(hipStream_t st, const ConvParams& p, int ntiles, float act_slope) {
    const size_t lds = (size_t)2 * PS_WBYTES + (size_t)2 * PS_NPATCH * ROWB + (size_t)PS_NTN * sizeof(float);
    static bool attr_set = false;
    auto kern = conv3x3_pps_kernel<ACT>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { dsvc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return DSVC_E_LAUNCH; }
        attr_set = true;
    }
    int grid = (ntiles + 1) / 2;
    if (grid > 256) grid = 256;                                   // one persistent workgroup per CU, two tiles in flight each
    const int nband = grid % 8 == 0 ? 8 : 1, band_tiles = (ntiles + nband - 1) / nband;   // as in the kernel
    const int iters = (band_tiles + 2 * (grid / nband) - 1) / (2 * (grid / nband));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, ntiles, iters, act_slope);
    DSVC_CHECK_LAUNCH("conv3x3_pps_kernel");
    return DSVC_OK;
}

bool ps_vec16_ok(const void* ptr, int ld) { return ((uintptr_t)ptr & 15) == 0 && (ld % 8) == 0; }

}  // namespace

// returns 1 when the shape is not covered (the caller falls through to the general kernel), else DSVC_OK / error
int dsvc_conv3x3_pps_try(hipStream_t st, const ConvParams& p_in, const dsvc_conv_desc* d) {
    if (d->compute != DSVC_F16 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->shuffle != 1) return 1;
    if (p_in.nchunks < 3 || p_in.nchunks > 16 || d->Cout != 64) return 1;
    if (d->res1 || d->res2 || d->res1_scale || d->post_weight || d->pre_act != DSVC_ACT_NONE) return 1;
    for (int s = 0; s < p_in.nseg; ++s) {                          // every pixel's chunk row is an aligned run of 16-byte vectors
        const ConvSeg& sg = p_in.seg[s];
        if (sg.dtype != DSVC_F16 || (sg.C % 8) != 0 || !ps_vec16_ok(sg.ptr, sg.ld)) return 1;
    }
    if (d->out_dtype != DSVC_F16 || !ps_vec16_ok(d->out, d->out_ld)) return 1;
    float act_slope;
    if (d->act == DSVC_ACT_NONE) act_slope = 1.f;
    else if (d->act == DSVC_ACT_RELU) act_slope = 0.f;
    else if (d->act == DSVC_ACT_LRELU && d->slope >= 0.f && d->slope <= 1.f) act_slope = d->slope;
    else return 1;
    ConvParams p = p_in;
    p.tiles_x = (p.Wo + 31) / 32;
    p.tiles_y = (p.Ho + PS_TH - 1) / PS_TH;
    const int ntiles = p.B * p.tiles_x * p.tiles_y;
    if ((int64_t)p.B * p.H * p.W >= (1 << 24)) return 1;          // pixel indices go through the 24-bit multiplier
    if (ntiles < 1024) return 1;                                   // small images: the general kernel's finer tiling fills the chip better
    return d->act == DSVC_ACT_NONE ? launch_pps<false>(st, p, ntiles, 1.f) : launch_pps<true>(st, p, ntiles, act_slope);
}

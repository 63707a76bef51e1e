// conv1x1.hip -- f16 pointwise (1x1) convolution as a pure streaming kernel (ResBottleneck conv1/conv3,
// ConvBlockResidual.up_dim: modules.py:176,178,242; image_model.py:25,27).
//
// A 1x1 conv has no halo, so nothing is reused across pixels except the weights: HBM-bound (2 bytes in + 2 bytes
// out per MAC-column).  Structure for that regime:
//   * the image is a flat list of pixels; a wave owns 32 consecutive pixels at a time (grid-stride, persistent waves);
//   * the MFMA A fragments are loaded STRAIGHT from HBM in operand layout -- lane (r,h) reads the 16 B holding channels
//     [16ks+8h, +8) of pixel r -- with bounds-checked buffer loads (the ragged tail reads zeros); no LDS staging, no
//     workgroup barriers; the next unit's fragments are in flight while the current unit's MFMAs and stores run;
//   * the weights (<= 128x128, <= 40 KiB) are staged into LDS once per workgroup and read as B fragments per MFMA
//     (keeping them in registers cost occupancy: 245 VGPRs at 96->96); output channels go in pairs of N-tiles;
//   * the epilogue is the shared fused pass (bias, activation, gated residual, second residual, cast) through a
//     wave-private LDS slab so that stores are 16-B vectors along the NHWC channel axis -- or, for the common case
//     (f16 out, activation in {none, ReLU, LeakyReLU}, at most one f16 residual: conv1 / conv3 of every
//     ResBottleneckBlock), a register epilogue: operand roles swapped (D[channel][pixel]), one v_permlane32_swap per
//     register so that each lane holds 8 consecutive channels of its pixel, bias preloaded into the accumulators, the
//     residual prefetched one unit ahead as 16-B vectors.  The slab path ran the residual case at 2.6 TB/s (118 us for
//     96->96 @544x960 against 46 us without residual).
#include <cstdlib>
#include "conv_common.h"

using namespace dsvc_conv;

namespace {


// register-epilogue variant (see the header): act(x) = max(x, x * act_slope) (slope 1: none, 0: ReLU)
// Up to two input segments of whole 32-channel chunks (the virtual torch.cat of the up-path ConvBlockResidual shortcuts);
// HAS_GATE: out = act(conv) + gate[c] * res1 (the SE gate of modules.py:226-247, batch 1).
// COAL: global memory is touched in whole pixel rows -- every load / store instruction moves 1 KB made of complete channel rows (8 lanes per
// 128-byte row) instead of 32-byte pieces of 32 different rows -- and a wave-private LDS tile (pitch = row + 16 bytes: conflict-free both
// ways) transposes between that layout and the MFMA operand layout.  Same arithmetic, same rounding: only the route of the bytes changes.
template <int KS, int NT, bool HAS_R1, bool HAS_GATE, bool COAL = false>
__global__ void __launch_bounds__(256) conv1x1_direct_kernel(const ConvParams p, int nunits, float act_slope) {
    DSVC_F16_SAT();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NCH = (KS + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    char* lds_w = smem;                                              // weights [chunk][Np rows] at the 80-B pitch
    float* lds_bias = (float*)(smem + (size_t)NCH * p.Np * ROWB);    // Np floats (+ Np floats of the gate)
    {
        const int nrows = NCH * p.Np, q = tid & 3;
        for (int row = tid >> 2; row < nrows; row += 64)
            *(uint4*)(lds_w + row * ROWB + q * 16) = *(const uint4*)((const char*)p.w + (size_t)row * 64 + q * 16);
        for (int c = tid; c < p.Np; c += 256) {
            lds_bias[c] = (p.bias && c < p.Cout) ? p.bias[c] : 0.f;
            if constexpr (HAS_GATE) lds_bias[p.Np + c] = c < p.Cout ? p.res1_scale[c] : 0.f;
        }
    }
    __syncthreads();
    const rsrc_t rin = make_rsrc(p.seg[0].ptr, p.seg[0].bytes);
    const bool two = p.nseg > 1;
    const rsrc_t rin1 = make_rsrc(p.seg[two ? 1 : 0].ptr, p.seg[two ? 1 : 0].bytes);
    const int ks0 = two ? p.seg[0].C / 16 : KS;                      // k-steps served by segment 0
    const unsigned pitch1 = (unsigned)p.seg[two ? 1 : 0].ld * 2;
    const rsrc_t rout = make_rsrc(p.out, p.out_bytes);
    const rsrc_t rr1 = make_rsrc(HAS_R1 ? p.res1 : p.out, HAS_R1 ? p.res1_bytes : p.out_bytes);
    const unsigned pitch = (unsigned)p.seg[0].ld * 2, opitch = (unsigned)p.out_ld * 2, rpitch = (unsigned)p.res1_ld * 2;
    const unsigned npix = (unsigned)p.Wo;
    const int nwaves = gridDim.x * 4;
    int unit = blockIdx.x * 4 + wave;
    auto load_a = [&](int u, uint4 (&a)[KS]) {
        const unsigned pix = (unsigned)u * 32 + r;
        const unsigned m = (pix < npix && u < nunits) ? 0u : OOB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            a[ks] = ks < ks0 ? buf_load16(rin, (pix * pitch + h * 16 + ks * 32) | m)
                             : buf_load16(rin1, (pix * pitch1 + h * 16 + (ks - ks0) * 32) | m);
    };
    // lane (r, h) ends up with channels n*32 + 16*P + 8*h .. +7 of pixel unit*32 + r
    auto load_r = [&](int u, uint4 (&rv)[NT * 2]) {
        const unsigned pix = (unsigned)u * 32 + r;
        const bool okp = pix < npix && u < nunits;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                const int co = n * 32 + 16 * P + 8 * h;
                rv[n * 2 + P] = buf_load16(rr1, (pix * rpitch + (unsigned)co * 2) | ((okp && co < p.Cout) ? 0u : OOB));
            }
    };
    const int b_lane = r * ROWB + h * 16;
    uint4 a_cur[KS], a_nxt[KS], r_cur[NT * 2], r_nxt[NT * 2];
    // ---- COAL: row-granular global access + LDS transposition ----
    constexpr int PPI = KS * 2, PPO = NT * 4;                       // 16-byte pieces per input / output pixel row
    constexpr int TP = ((PPI > PPO ? PPI : PPO) + 1) * 16;          // tile pitch in bytes
    char* tile = smem + (size_t)NCH * p.Np * ROWB + (size_t)2 * p.Np * sizeof(float) + (size_t)wave * (32 * TP);
    auto load_g = [&](int u, uint4 (&g)[KS]) {                      // input rows of unit u, vector v = 64 i + lane -> (pixel v / PPI, piece v % PPI)
        const bool uok = u < nunits;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const int v = i * 64 + lane, px = v / PPI, pc = v - px * PPI;
            const unsigned pix = (unsigned)u * 32 + px;
            const unsigned m = (uok && pix < npix) ? 0u : OOB;
            g[i] = (pc < ks0 * 2) ? buf_load16(rin, (pix * pitch + (unsigned)pc * 16u) | m)
                                  : buf_load16(rin1, (pix * pitch1 + (unsigned)(pc - ks0 * 2) * 16u) | m);
        }
    };
    auto load_gr = [&](int u, uint4 (&g)[NT * 2]) {                 // residual rows in the same layout as the output rows
        const bool uok = u < nunits;
#pragma unroll
        for (int i = 0; i < NT * 2; ++i) {
            const int v = i * 64 + lane, px = v / PPO, pc = v - px * PPO;
            const unsigned pix = (unsigned)u * 32 + px;
            g[i] = buf_load16(rr1, (pix * rpitch + (unsigned)pc * 16u) | ((uok && pix < npix && pc * 8 < p.Cout) ? 0u : OOB));
        }
    };
    auto tile_sync = [&]() { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); };   // DS ops of a wave execute in order: only the compiler must not reorder
    if constexpr (COAL) {
        load_g(unit, a_nxt);
        if constexpr (HAS_R1) load_gr(unit, r_nxt);
    } else {
        load_a(unit, a_cur);
        if constexpr (HAS_R1) load_r(unit, r_cur);
    }
    for (; unit < nunits; unit += nwaves) {
        if constexpr (COAL) {
            // rows -> tile -> operand layout: lane (r, h) takes piece 2 ks + h of pixel r
#pragma unroll
            for (int i = 0; i < KS; ++i) { const int v = i * 64 + lane, px = v / PPI, pc = v - px * PPI; *(uint4*)(tile + px * TP + pc * 16) = a_nxt[i]; }
            tile_sync();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a_cur[ks] = *(const uint4*)(tile + r * TP + (2 * ks + h) * 16);
            tile_sync();
            if constexpr (HAS_R1) {
#pragma unroll
                for (int i = 0; i < NT * 2; ++i) { const int v = i * 64 + lane, px = v / PPO, pc = v - px * PPO; *(uint4*)(tile + px * TP + pc * 16) = r_nxt[i]; }
                tile_sync();
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int P = 0; P < 2; ++P) r_cur[n * 2 + P] = *(const uint4*)(tile + r * TP + (n * 4 + 2 * P + h) * 16);
                tile_sync();
            }
            load_g(unit + nwaves, a_nxt);                          // prefetch (masked past the end)
            if constexpr (HAS_R1) load_gr(unit + nwaves, r_nxt);
        } else {
            load_a(unit + nwaves, a_nxt);                          // prefetch (masked past the end)
            if constexpr (HAS_R1) load_r(unit + nwaves, r_nxt);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned pix = (unsigned)unit * 32 + r;
        const bool okp = pix < npix;
#define C1D_TILE(N0)                                                                                                 \
    if constexpr (N0 < NT) {                                                                                         \
        f32x16 acc;                                                                                                  \
        _Pragma("unroll") for (int qd = 0; qd < 4; ++qd) {      /* accumulator row (reg i) = channel (i&3) + 8*(i>>2) + 4*h */ \
            const float4 bq = *(const float4*)(lds_bias + N0 * 32 + 8 * qd + 4 * h);                                  \
            acc[4 * qd] = bq.x; acc[4 * qd + 1] = bq.y; acc[4 * qd + 2] = bq.z; acc[4 * qd + 3] = bq.w;              \
        }                                                                                                            \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                          \
            const uint4 bw = *(const uint4*)(lds_w + ((ks >> 1) * p.Np + N0 * 32) * ROWB + b_lane + (ks & 1) * 32);   \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, bw), __builtin_bit_cast(h16x8, a_cur[ks]), acc, 0, 0, 0); \
        }                                                                                                            \
        _Pragma("unroll") for (int P = 0; P < 2; ++P) {                                                              \
            float v[8];                                                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
                const float xq = acc[8 * P + j], yq = acc[8 * P + 4 + j];   /* scalars first: hipcc 7.2 mis-casts vector elements */ \
                const v2u_t sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false); \
                const unsigned lo_ = sw.x, hi_ = sw.y;                                                               \
                v[j] = __builtin_bit_cast(float, lo_); v[4 + j] = __builtin_bit_cast(float, hi_);                    \
            }                                                                                                        \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j], v[j] * act_slope, __builtin_inff()); \
            if constexpr (HAS_R1) {                                                                                  \
                const uint4 a = r_cur[N0 * 2 + P];                                                                   \
                const unsigned ua[4] = {a.x, a.y, a.z, a.w};                                                         \
                if constexpr (HAS_GATE) {                                                                            \
                    const float4 g0 = *(const float4*)(lds_bias + p.Np + N0 * 32 + 16 * P + 8 * h);                  \
                    const float4 g1 = *(const float4*)(lds_bias + p.Np + N0 * 32 + 16 * P + 8 * h + 4);              \
                    const float gt[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};                            \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                  \
                        v[2 * j] += gt[2 * j] * h16_lo(ua[j]);                              \
                        v[2 * j + 1] += gt[2 * j + 1] * h16_hi(ua[j]);              \
                    }                                                                                                \
                } else {                                                                                             \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                  \
                        v[2 * j] += h16_lo(ua[j]);                                          \
                        v[2 * j + 1] += h16_hi(ua[j]);                              \
                    }                                                                                                \
                }                                                                                                    \
            }                                                                                                        \
            const int co = N0 * 32 + 16 * P + 8 * h;                                                                 \
            v4u_t ov;                                                                                                \
            ov.x = pk_h16(v[0], v[1]); ov.y = pk_h16(v[2], v[3]); ov.z = pk_h16(v[4], v[5]); ov.w = pk_h16(v[6], v[7]); \
            if constexpr (COAL) *(v4u_t*)(tile + r * TP + (N0 * 4 + 2 * P + h) * 16) = ov;                           \
            else __builtin_amdgcn_raw_buffer_store_b128(ov, rout, (pix * opitch + (unsigned)co * 2) | ((okp && co < p.Cout) ? 0u : OOB), 0, 0); \
        }                                                                                                            \
    }
        C1D_TILE(0) C1D_TILE(1) C1D_TILE(2) C1D_TILE(3)
#undef C1D_TILE
        if constexpr (COAL) {                                      // tile -> whole output rows
            tile_sync();
#pragma unroll
            for (int i = 0; i < NT * 2; ++i) {
                const int v = i * 64 + lane, px = v / PPO, pc = v - px * PPO;
                const uint4 o = *(const uint4*)(tile + px * TP + pc * 16);
                const unsigned opx = (unsigned)unit * 32 + px;
                v4u_t ov; ov.x = o.x; ov.y = o.y; ov.z = o.z; ov.w = o.w;
                __builtin_amdgcn_raw_buffer_store_b128(ov, rout, (opx * opitch + (unsigned)pc * 16u) | ((opx < npix && pc * 8 < p.Cout) ? 0u : OOB), 0, 0);
            }
            tile_sync();
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a_cur[ks] = a_nxt[ks];
            if constexpr (HAS_R1) {
#pragma unroll
                for (int i = 0; i < NT * 2; ++i) r_cur[i] = r_nxt[i];
            }
        }
    }
}

template <int KS, int NT>
__global__ void __launch_bounds__(256) conv1x1_kernel(const ConvParams p, int nunits) {
    DSVC_F16_SAT();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPN = NT >= 2 ? 2 : 1;
    constexpr int EPLD = EPN * 32 + 4, GPP = EPN * 8;
    constexpr int NCH = (KS + 1) / 2;                          // 64-B weight chunks (32 input channels each)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // LDS: weights [chunk][Np rows] at the 80-B pitch (conflict-free ds_read_b128) | one epilogue slab per wave
    char* lds_w = smem;
    float* slab = (float*)(smem + (size_t)NCH * p.Np * ROWB) + wave * (32 * EPLD);
    {
        const int nrows = NCH * p.Np, q = tid & 3;
        for (int row = tid >> 2; row < nrows; row += 64)
            *(uint4*)(lds_w + row * ROWB + q * 16) = *(const uint4*)((const char*)p.w + (size_t)row * 64 + q * 16);
    }
    __syncthreads();
    const EpiRsrc R = make_epi_rsrc(p);
    const rsrc_t rin = make_rsrc(p.seg[0].ptr, p.seg[0].bytes);
    const unsigned pitch = (unsigned)p.seg[0].ld * 2;
    const unsigned npix = (unsigned)p.Wo;                      // host presents the tensor as one row of B*H*W pixels

    const int nwaves = gridDim.x * 4;
    int unit = blockIdx.x * 4 + wave;
    auto load_a = [&](int u, uint4 (&a)[KS]) {
        const unsigned pix = (unsigned)u * 32 + r;
        const bool ok = pix < npix && u < nunits;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = buf_load16(rin, ok ? pix * pitch + h * 16 + ks * 32 : OOB);
    };
    const int b_lane = r * ROWB + h * 16;
    uint4 a_cur[KS], a_nxt[KS];
    load_a(unit, a_cur);
    for (; unit < nunits; unit += nwaves) {
        load_a(unit + nwaves, a_nxt);                          // prefetch (reads zeros past the end)
        __builtin_amdgcn_sched_barrier(0);
        // output channels in pairs of N-tiles: accumulate, then run the fused epilogue pass for that pair
#define C1_PAIR(N0)                                                                                                  \
    if constexpr (N0 < NT) {                                                                                         \
        constexpr int NP = NT - N0 < EPN ? NT - N0 : EPN;                                                            \
        f32x16 acc[EPN];                                                                                             \
        _Pragma("unroll") for (int e = 0; e < EPN; ++e)                                                              \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[e][i] = 0.f;                                          \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                            \
            _Pragma("unroll") for (int e = 0; e < NP; ++e) {                                                         \
                const uint4 bw = *(const uint4*)(lds_w + ((ks >> 1) * p.Np + (N0 + e) * 32) * ROWB + b_lane + (ks & 1) * 32); \
                acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a_cur[ks]),              \
                                                                 __builtin_bit_cast(h16x8, bw), acc[e], 0, 0, 0);   \
            }                                                                                                        \
        _Pragma("unroll") for (int e = 0; e < NP; ++e)                                                               \
            _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                           \
                slab[((i & 3) + 8 * (i >> 2) + 4 * h) * EPLD + e * 32 + r] = acc[e][i];                              \
        __builtin_amdgcn_wave_barrier();                                                                             \
        run_epilogue<GPP, EPLD, 32>(p, R, slab, lane, 0, 0, unit * 32, N0 * 32, NP * 32);                            \
        __builtin_amdgcn_wave_barrier();                                                                             \
    }
        C1_PAIR(0) C1_PAIR(2)
#undef C1_PAIR
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a_cur[ks] = a_nxt[ks];
    }
}

template <int KS, int NT>
int launch_1x1(hipStream_t st, const ConvParams& p, int nunits) {
    constexpr int EPN = NT >= 2 ? 2 : 1;
    const size_t lds = (size_t)((KS + 1) / 2) * p.Np * ROWB + (size_t)4 * 32 * (EPN * 32 + 4) * sizeof(float);
    int grid = (nunits + 3) / 4;
    if (grid > 256 * 4) grid = 256 * 4;                        // 4 workgroups (16 waves) per CU, grid-stride beyond
    hipLaunchKernelGGL((conv1x1_kernel<KS, NT>), dim3(grid), dim3(256), lds, st, p, nunits);
    DSVC_CHECK_LAUNCH("conv1x1_kernel");
    return DSVC_OK;
}

template <int KS, int NT>
int launch_1x1_direct(hipStream_t st, const ConvParams& p, int nunits, float act_slope) {
    const bool coal = g_dsvc_opt.conv1x1_rows != 0;
    constexpr int TP = ((KS * 2 > NT * 4 ? KS * 2 : NT * 4) + 1) * 16;
    const size_t lds = (size_t)((KS + 1) / 2) * p.Np * ROWB + (size_t)2 * p.Np * sizeof(float) + (coal ? (size_t)4 * 32 * TP : 0);
    int grid = (nunits + 3) / 4;
    if (grid > 256 * 4) grid = 256 * 4;
#define C1D_LAUNCH(R1, GATE)                                                                                                          \
    do {                                                                                                                              \
        if (coal) {                                                                                                                   \
            auto kern = conv1x1_direct_kernel<KS, NT, R1, GATE, true>;                                                                \
            static bool attr_set = false;                    /* 128 -> 128: 42 KB of weights + 35 KB of transposition tiles */        \
            if (!attr_set && lds > 64 * 1024) {                                                                                       \
                hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
                if (e != hipSuccess) { dsvc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return DSVC_E_LAUNCH; }       \
                attr_set = true;                                                                                                      \
            }                                                                                                                         \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p, nunits, act_slope);                                           \
        } else hipLaunchKernelGGL((conv1x1_direct_kernel<KS, NT, R1, GATE, false>), dim3(grid), dim3(256), lds, st, p, nunits, act_slope);   \
    } while (0)
    if (p.res1 && p.res1_scale) C1D_LAUNCH(true, true);
    else if (p.res1) C1D_LAUNCH(true, false);
    else C1D_LAUNCH(false, false);
#undef C1D_LAUNCH
    DSVC_CHECK_LAUNCH("conv1x1_direct_kernel");
    return DSVC_OK;
}

template <int KS>
int launch_1x1_direct_nt(hipStream_t st, const ConvParams& p, int nunits, int NT, float act_slope) {
    switch (NT) {
        case 1: return launch_1x1_direct<KS, 1>(st, p, nunits, act_slope);
        case 2: return launch_1x1_direct<KS, 2>(st, p, nunits, act_slope);
        case 3: return launch_1x1_direct<KS, 3>(st, p, nunits, act_slope);
        default: return launch_1x1_direct<KS, 4>(st, p, nunits, act_slope);
    }
}

template <int KS>
int launch_1x1_nt(hipStream_t st, const ConvParams& p, int nunits, int NT) {
    switch (NT) {
        case 1: return launch_1x1<KS, 1>(st, p, nunits);
        case 2: return launch_1x1<KS, 2>(st, p, nunits);
        case 3: return launch_1x1<KS, 3>(st, p, nunits);
        default: return launch_1x1<KS, 4>(st, p, nunits);
    }
}

}  // namespace

// returns 1 when the shape is not covered (caller falls back to the generic kernel), else DSVC_OK / error
int dsvc_conv1x1_try(hipStream_t st, const ConvParams& p_in, const dsvc_conv_desc* d) {
    if (d->compute != DSVC_F16 || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->shuffle != 1) return 1;
    if (d->nseg > 2 || d->pre_act != DSVC_ACT_NONE || !p_in.epi_vec || (d->res1_scale && d->B > 1)) return 1;
    int ctot = 0;
    for (int s = 0; s < d->nseg; ++s) {                        // whole 32-channel chunks of aligned f16 rows
        const ConvSeg& g = p_in.seg[s];
        if (g.dtype != DSVC_F16 || !g.vec_ok || (g.C % 32) != 0) return 1;
        ctot += g.C;
    }
    if (ctot > 128 || d->Cout > 128) return 1;
    ConvSeg sg = p_in.seg[0];
    sg.C = ctot;                                               // total input channels select the k-step count
    const int64_t npix = (int64_t)d->B * d->H * d->W;
    // 16 K pixels and up (136x240 -- the 1/8-resolution bottlenecks -- included: 96 -> 96 + residual 8.1 us streamed, 25.2 us in the general
    // kernel; 64 -> 64 8.4 vs 11.4); the entropy-parameter nets' images (68x120) stay on the general kernel
    static const int64_t min_px = getenv("DSVC_1X1_MIN_PX") ? atoll(getenv("DSVC_1X1_MIN_PX")) : 16 * 1024;
    if (npix < min_px) return 1;
    ConvParams p = p_in;
    p.B = 1; p.Ho = 1; p.Wo = (int)npix;                        // flat pixel list: offsets are pixel-linear
    const int nunits = (int)((npix + 31) / 32);
    const int NT = (d->Cout + 31) / 32;
    // register epilogue: f16 out, activation none / ReLU / LeakyReLU, at most one plain f16 residual, 16-byte rows
    float act_slope = 1.f;
    const bool act_ok = d->act == DSVC_ACT_NONE || (d->act == DSVC_ACT_RELU ? (act_slope = 0.f, true)
                        : (d->act == DSVC_ACT_LRELU && d->slope >= 0.f && d->slope <= 1.f ? (act_slope = d->slope, true) : false));
    auto v16 = [](const void* ptr, int ld) { return ((uintptr_t)ptr & 15) == 0 && (ld % 8) == 0; };
    static const bool direct_on = getenv("DSVC_1X1_SLAB") == nullptr;
    if (direct_on && act_ok && d->out_dtype == DSVC_F16 && v16(d->out, d->out_ld) && (d->Cout % 8) == 0 && !d->res2 &&
        (!d->res1_scale || d->res1) && (!d->res1 || (d->res1_dtype == DSVC_F16 && v16(d->res1, d->res1_ld))) && npix < (1 << 24)) {
        switch (sg.C / 16) {
            case 2: return launch_1x1_direct_nt<2>(st, p, nunits, NT, act_slope);
            case 4: return launch_1x1_direct_nt<4>(st, p, nunits, NT, act_slope);
            case 6: return launch_1x1_direct_nt<6>(st, p, nunits, NT, act_slope);
            default: return launch_1x1_direct_nt<8>(st, p, nunits, NT, act_slope);
        }
    }
    if (d->nseg != 1) return 1;                                // the LDS-slab variant reads one segment
    switch (sg.C / 16) {
        case 2: return launch_1x1_nt<2>(st, p, nunits, NT);
        case 4: return launch_1x1_nt<4>(st, p, nunits, NT);
        case 6: return launch_1x1_nt<6>(st, p, nunits, NT);
        default: return launch_1x1_nt<8>(st, p, nunits, NT);
    }
}

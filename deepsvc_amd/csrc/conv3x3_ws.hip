// conv3x3_ws.hip -- f16 3x3 stride-1 convolution with <= 64 input channels at large resolution: the shape family
// that carries ~2/3 of a P-frame's MACs (ResBlock1 / ResBottleneck conv2 /
// in_conv / weight heads of modules.py:177,200-204,307-311,322-340,382-418).
//
// Structure -- chosen for compiler-scheduled HIP on CDNA4, where latency is hidden by having
// two waves per SIMD rather than by hand-placed instruction streams:
//   * PERSISTENT workgroups, one per CU, 8 waves (2 per SIMD), walking 16x32-pixel tiles: no block launch tail;
//   * WEIGHT-STATIONARY: all taps x input chunks x output channels of the conv (<= 92 KiB with the 80-B row pitch)
//     are staged into LDS ONCE per workgroup and stay there; per tile only the input halo patch moves
//     (the generic kernel re-stages the weights for every 8x32 tile: 63 % of its L2->LDS traffic);
//   * the patch of the NEXT (tile, channel chunk) is prefetched into registers with unconditional, clamped,
//     straight-line loads before the current chunk's MFMA phase and written to LDS after it (async-stage split):
//     HBM/L2 latency rides under 36 MFMAs per wave; zero padding and the fused ResBlock pre-activation are
//     applied on the way into LDS;
//   * wave w owns tile rows 2w, 2w+1 (two MFMA M-tiles) and all NT*32 output channels; LDS rows are 64 B + 16 B pad
//     (conflict-free ds_read_b128); the A/B byte->k assignment is the generic kernel's;
//   * epilogue = the shared fused pass through wave-private LDS slabs that alias the (then idle) patch buffer.
#include <cstdlib>
#include "conv_common.h"

using namespace dsvc_conv;

namespace {

constexpr int WS_TH = 16, WS_PH = 18, WS_PW = 34, WS_NPATCH = WS_PH * WS_PW;   // 612 patch pixels

// WAVES = 8: two waves per SIMD, each 2x NT accumulator tiles; WAVES = 4: ONE wave per SIMD with a 4 x NT register tile
// (every fragment read feeds twice the MFMAs, no MFMA-pipe contention between lock-stepped partner waves).
template <int NT, bool FAST, bool DIRECT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) conv3x3_ws_kernel(const ConvParams p, int ntiles) {
    DSVC_F16_SAT();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTN = NT * 32, KCE = 32, EPV = 8;
    constexpr int NTHR = WAVES * 64, MTW = 16 / WAVES;          // tile rows per wave
    constexpr int WS_PFV = (WS_NPATCH * 4 + NTHR - 1) / NTHR;    // prefetch vectors per thread
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nchunks = p.nchunks;
    char* lds_w = smem;                                              // [chunk][tap][NTN] rows of 80 B
    char* lds_a = smem + (size_t)nchunks * 9 * NTN * ROWB;           // 612 rows of 80 B; epilogue slabs alias it
    float* slab = (float*)lds_a + wave * (32 * 36);
    const EpiRsrc R = make_epi_rsrc(p);
    const rsrc_t rbias = make_rsrc(p.bias, p.bias ? (unsigned)p.Cout * 4u : 0u);   // absent bias reads zeros

    // ---- stage every weight row once (batched 16-B copies) ----
    {
        const int nrows = nchunks * 9 * NTN;                         // global rows are [chunk][tap][Np] x 64 B, Np == NTN here
        const int q = tid & 3;
        for (int base = tid >> 2; base < nrows; base += (NTHR / 4) * 6) {
            uint4 v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int row = base + (NTHR / 4) * j;
                v[j] = make_uint4(0, 0, 0, 0);
                if (row < nrows) v[j] = *(const uint4*)((const char*)p.w + (size_t)row * 64 + q * 16);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (base + (NTHR / 4) * j < nrows) *(uint4*)(lds_w + (base + (NTHR / 4) * j) * ROWB + q * 16) = v[j];
        }
    }

    const int H = p.H, W = p.W, tiles_x = p.tiles_x, tiles_y = p.tiles_y, pre_act = p.pre_act;
    const float pre_sl = pre_act == DSVC_ACT_RELU ? 0.f : p.slope;
    auto tile_origin = [&](int tile, int& b, int& oy0, int& ox0) {
        const int tx = tile % tiles_x;
        const int rest = tile / tiles_x;
        const int ty = rest % tiles_y;
        b = rest / tiles_y;
        oy0 = ty * WS_TH; ox0 = tx * 32;
    };
    // patch vector v = tid + 512*j  ->  pixel v>>2 (row-major in the 18x34 patch), 16-B quarter v&3
    const int q = tid & 3;
    int ppy[WS_PFV], ppx[WS_PFV];
#pragma unroll
    for (int j = 0; j < WS_PFV; ++j) {
        const int pix = (tid >> 2) + (NTHR / 4) * j;
        ppy[j] = pix / WS_PW; ppx[j] = pix - ppy[j] * WS_PW;
    }
    auto seg_of = [&](int ch) {
        int s = 0;
        if (p.nseg > 1 && ch >= p.seg[1].chunk0) s = 1;
        if (p.nseg > 2 && ch >= p.seg[2].chunk0) s = 2;
        return s;
    };
    auto load_chunk = [&](int tile, int ch, uint4 (&pf)[WS_PFV]) {
        int b, oy0, ox0;
        tile_origin(tile, b, oy0, ox0);
        const ConvSeg& sg = p.seg[seg_of(ch)];
        const int c0 = (ch - sg.chunk0) * KCE;
#pragma unroll
        for (int j = 0; j < WS_PFV; ++j) {
            const int iy = oy0 - 1 + ppy[j], ix = ox0 - 1 + ppx[j];
            if constexpr (FAST) {   // bounds-checked buffer load: out-of-image pixels read as zeros, code stays straight-line
                const bool inb = c0 + q * EPV < sg.C && iy >= 0 && iy < H && ix >= 0 && ix < W;   // short chunk: zeros
                pf[j] = buf_load16(make_rsrc(sg.ptr, sg.bytes), inb ? ((((unsigned)b * H + iy) * W + ix) * sg.ld + c0) * 2 + q * 16 : OOB);
            } else {
                pf[j] = make_uint4(0, 0, 0, 0);
                if ((tid >> 2) + (NTHR / 4) * j < WS_NPATCH && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    float vals[EPV];
                    load_vec<EPV>(sg, (((int64_t)b * H + iy) * W + ix) * sg.ld, c0 + q * EPV, vals);
                    pf[j] = pack_vec<h16, EPV>(vals);
                }
            }
        }
    };
    auto store_chunk = [&](uint4 (&pf)[WS_PFV]) {
#pragma unroll
        for (int j = 0; j < WS_PFV; ++j) {
            const int pix = (tid >> 2) + (NTHR / 4) * j;
            if (pix >= WS_NPATCH) continue;
            uint4 val = pf[j];
            if (pre_act != DSVC_ACT_NONE) {   // ResBlock1's leading ReLU / LeakyReLU on the packed f16 pairs
                unsigned u[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = h16_lo(u[k]), hi = h16_hi(u[k]);
                    u[k] = pk_h16(fmaxf(lo, lo * pre_sl), fmaxf(hi, hi * pre_sl));
                }
                val = make_uint4(u[0], u[1], u[2], u[3]);
            }
            *(uint4*)(lds_a + pix * ROWB + q * 16) = val;
        }
    };

    int tile = blockIdx.x;
    uint4 pf[WS_PFV];
    load_chunk(min(tile, ntiles - 1), 0, pf);
    store_chunk(pf);
    __syncthreads();

    int a_off[MTW], b_off[NT];
#pragma unroll
    for (int m = 0; m < MTW; ++m) a_off[m] = ((wave * MTW + m) * WS_PW + r) * ROWB + h * 32;
#pragma unroll
    for (int n = 0; n < NT; ++n) b_off[n] = (n * 32 + r) * ROWB + h * 32;

    for (; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[MTW][NT];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

        for (int ch = 0; ch < nchunks; ++ch) {
            // what the patch buffer must hold next: the following chunk of this tile, or chunk 0 of the next tile
            const bool last = ch + 1 == nchunks;
            const int ntile = last ? min(tile + (int)gridDim.x, ntiles - 1) : tile;
            const int nch = last ? 0 : ch + 1;
            load_chunk(ntile, nch, pf);                          // in flight during this chunk's 36 MFMAs / wave
            __builtin_amdgcn_sched_barrier(0);                   // hipcc otherwise sinks these loads below the MFMA phase

            const char* wt = lds_w + (size_t)ch * 9 * NTN * ROWB;
            // 18 k-steps (9 taps x 2 halves of the 64-B chunk), fragments of step s+1 are read before the MFMAs of step s
            auto read_frags = [&](int s, uint4 (&af)[MTW], uint4 (&bf)[NT]) {
                const int t = s >> 1, ks = s & 1;
                const int tap_off = ((t / 3) * WS_PW + (t % 3)) * ROWB;
#pragma unroll
                for (int n = 0; n < NT; ++n) bf[n] = *(const uint4*)(wt + t * NTN * ROWB + b_off[n] + ks * 16);
#pragma unroll
                for (int m = 0; m < MTW; ++m) af[m] = *(const uint4*)(lds_a + a_off[m] + tap_off + ks * 16);
            };
            uint4 af[2][MTW], bf[2][NT];
            read_frags(0, af[0], bf[0]);
#pragma unroll
            for (int s2 = 0; s2 < 18; ++s2) {
                if (s2 + 1 < 18) read_frags(s2 + 1, af[(s2 + 1) & 1], bf[(s2 + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);   // pin: next step's LDS reads are issued BEFORE this step's MFMAs
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        if constexpr (DIRECT)   // swapped roles: D[channel][pixel], the pixel lands on the lane
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, bf[s2 & 1][n]),
                                                                                 __builtin_bit_cast(h16x8, af[s2 & 1][m]), acc[m][n], 0, 0, 0);
                        else
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, af[s2 & 1][m]),
                                                                                 __builtin_bit_cast(h16x8, bf[s2 & 1][n]), acc[m][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }

            if (last && DIRECT) {   // accumulators -> HBM straight from registers (pixel on the lane, 4 channels per vector)
                int b, oy0, ox0;
                tile_origin(tile, b, oy0, ox0);
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const int oy = oy0 + wave * MTW + m, ox = ox0 + r;
                    const bool pix_ok = oy < p.Ho && ox < p.Wo;
                    const unsigned opix = ((unsigned)b * p.Ho + oy) * p.Wo + ox;
#pragma unroll
                    for (int n = 0; n < NT; ++n) run_epilogue_direct(p, R, rbias, acc[m][n], opix, pix_ok, n * 32, h);
                }
            }
            lds_barrier();                                       // every wave is done reading this chunk's patch
            if (last && !DIRECT) {                               // epilogue through slabs that alias the patch buffer
                int b, oy0, ox0;
                tile_origin(tile, b, oy0, ox0);
                // passes written out with static accumulator indices (a rolled (m,n) loop would push acc[][] to scratch)
#define WS_PASS(M, N)                                                                                              \
    if constexpr (N < NT && M < MTW) {                                                                             \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) slab[((i & 3) + 8 * (i >> 2) + 4 * h) * 36 + r] = acc[M][N][i]; \
        __builtin_amdgcn_wave_barrier();                                                                           \
        run_epilogue<8, 36, 32>(p, R, slab, lane, b, oy0 + wave * MTW + M, ox0, N * 32, 32);                       \
        __builtin_amdgcn_wave_barrier();                                                                           \
    }
                WS_PASS(0, 0) WS_PASS(0, 1) WS_PASS(1, 0) WS_PASS(1, 1) WS_PASS(2, 0) WS_PASS(2, 1) WS_PASS(3, 0) WS_PASS(3, 1)
#undef WS_PASS
                lds_barrier();                                   // slabs are dead before the patch buffer is refilled
            }
            store_chunk(pf);
            lds_barrier();
        }
    }
}

template <int NT, bool FAST, bool DIRECT, int WAVES>
int launch_ws(hipStream_t st, const ConvParams& p, int ntiles) {
    const size_t lds = (size_t)p.nchunks * 9 * NT * 32 * ROWB + (size_t)WS_NPATCH * ROWB;
    static bool attr_set = false;
    auto kern = conv3x3_ws_kernel<NT, FAST, DIRECT, WAVES>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { dsvc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return DSVC_E_LAUNCH; }
        attr_set = true;
    }
    const int grid = ntiles < 256 ? ntiles : 256;   // one persistent workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, st, p, ntiles);
    DSVC_CHECK_LAUNCH("conv3x3_ws_kernel");
    return DSVC_OK;
}

}  // namespace

// returns 1 when the shape is not covered (caller falls back to the generic kernel), else DSVC_OK / error
int dsvc_conv3x3_ws_try(hipStream_t st, const ConvParams& p_in, const dsvc_conv_desc* d) {
    if (d->compute != DSVC_F16 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->shuffle != 1) return 1;
    if (p_in.nchunks > 2 || d->Cout > 64) return 1;
    ConvParams p = p_in;
    p.tiles_x = (p.Wo + 31) / 32;
    p.tiles_y = (p.Ho + WS_TH - 1) / WS_TH;
    const int ntiles = p.B * p.tiles_x * p.tiles_y;
    if (ntiles < 512) return 1;            // small images: the generic kernel's finer tiling fills the chip better
    bool fast = true;
    for (int s = 0; s < p.nseg; ++s) fast = fast && p.seg[s].vec_ok && p.seg[s].dtype == DSVC_F16 && (p.seg[s].C % 8) == 0;
    const int NT = d->Cout > 32 ? 2 : 1;
    // Direct stores from swapped-operand accumulators are implemented but measured SLOWER than the LDS-transposed epilogue
    // (285 vs 257 us on 64->64 @1088x1920: each store instruction touches 32 cache lines); opt-in for experiments only.
    const bool direct_enabled = g_dsvc_opt.ws_direct_store != 0;
    const bool direct = direct_enabled && p.epi_vec && !p.res1_scale;
    // one wave per SIMD with a 4xNT register tile measured slower (371 vs 251 us): opt-in only
    const bool one_wave_per_simd = g_dsvc_opt.ws_waves4 != 0;
    if (direct) {
        if (fast) return NT == 2 ? launch_ws<2, true, true, 8>(st, p, ntiles) : launch_ws<1, true, true, 8>(st, p, ntiles);
        return NT == 2 ? launch_ws<2, false, true, 8>(st, p, ntiles) : launch_ws<1, false, true, 8>(st, p, ntiles);
    }
    if (one_wave_per_simd) {
        if (fast) return NT == 2 ? launch_ws<2, true, false, 4>(st, p, ntiles) : launch_ws<1, true, false, 4>(st, p, ntiles);
        return NT == 2 ? launch_ws<2, false, false, 4>(st, p, ntiles) : launch_ws<1, false, false, 4>(st, p, ntiles);
    }
    if (fast) return NT == 2 ? launch_ws<2, true, false, 8>(st, p, ntiles) : launch_ws<1, true, false, 8>(st, p, ntiles);
    return NT == 2 ? launch_ws<2, false, false, 8>(st, p, ntiles) : launch_ws<1, false, false, 8>(st, p, ntiles);
}

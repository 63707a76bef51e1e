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

#ifdef DSVC_PPS_PROBE
// python -m deepsvc_amd.build --variant ppsprobe -DDSVC_PPS_PROBE; tools/pps_phase_probe.py: for wave 0 of each group of every workgroup, shader clocks
// [g*8 + 0] M phases, [1] waiting at the barrier that ends them, [2] S phases (patch / weight commit + requests), [3] the barrier after them,
// [4] epilogues (part of the S slot), [5] s_memrealtime ticks of the whole kernel, [6] s_memtime clocks of the whole kernel; [16] = stream steps
__device__ unsigned long long g_pps_phase[17];
extern "C" int dsvc_dbg_pps_phase_read(unsigned long long* out17, int reset) {
    if (hipMemcpyFromSymbol(out17, HIP_SYMBOL(g_pps_phase), sizeof(g_pps_phase)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[17] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pps_phase), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define PPS_STAMP(v) { v = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#define PPS_ADD(i, a, b) { psp[i] += (b) - (a); }
#else
#define PPS_STAMP(v)
#define PPS_ADD(i, a, b)
#endif

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
// 8 packed f16 -> max(x, x * slope) -> 8 packed f16, bit-identical to cvt -> fmaxf(x, x * slope) -> cvt (see conv3x3_pp.hip: act_h16x8)
__device__ __forceinline__ uint4 ps_act_h16x8(uint4 val, float slope) {
    unsigned u[4] = {val.x, val.y, val.z, val.w};
    if (slope == 0.f) {
#pragma unroll
        for (int k = 0; k < 4; ++k) asm("v_pk_max_i16 %0, %1, 0" : "=v"(u[k]) : "v"(u[k]));
        return make_uint4(u[0], u[1], u[2], u[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned t = 0;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "+v"(t) : "v"(u[k]), "v"(slope));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(t) : "v"(u[k]), "v"(slope));
        asm("v_pk_max_f16 %0, %1, %2" : "=v"(u[k]) : "v"(u[k]), "v"(t));
    }
    return make_uint4(u[0], u[1], u[2], u[3]);
}

// CT = h16: f16 storage and MFMA (v_mfma_f32_32x32x16_f16), activation none / ReLU / LeakyReLU (ACT), no residuals (the shapes above).
// CT = float [round 4]: fp32 storage and v_mfma_f32_32x32x2_f32 -- the PARITY mode's arithmetic -- over 16-channel chunks (64 bytes, so
// every byte count above is unchanged): 64 -> 64 is four chunks, 64+64 -> 64 eight.  An M phase is then 288 MFMAs of 64 cycles (18 432
// cycles against ~3 000 for everything else a group does per phase), so the schedule keeps one wave per SIMD on the matrix pipe all the
// time and the epilogue has slack: pre-activation, activation and up to two fp32 residuals are RUN-TIME parameters and the residuals are
// loaded where they are used.  Operation order per output element is the general kernel's (conv_mfma.hip + epilogue_vec): MFMAs by
// (chunk, tap, 16-byte half, element), + bias, activation, + (res1 + res2) -- fp32 results are bit-identical to it, which is what keeps
// the byte-identity tests against the reference green.
// EXT (f16 only) [r4]: output-channel split (96 / 128 output channels as two passes of 64), pre-activation on the staged patch, one f16 residual.
// Kept out of the plain instantiations at compile time: they sit at 238-244 VGPRs and the extra address arithmetic spilled them.
// ONESEG [r5]: one input segment (the fp32 64 -> 64 family, the f16 128 -> 128 split): the patch requests use tile-invariant addressing as in
// conv3x3_pp.hip -- descriptor (one image row + one pixel of guard in front), per-thread vector offsets and static masks computed once, the tile
// and the chunk in the SCALAR offset operand, border masks only on border tiles.  The fp32 instantiation issues its requests from inside the
// M phase, one per k-step, and each one used to re-read the segment descriptor from the kernel arguments (two scalar loads + a wait) and
// spend ~9 vector instructions on address and mask: the wave's own MFMA stream paid 110-150 clocks per request (tools/pps_phase_probe.py).
template <typename CT, bool ACT, bool EXT = false, bool R1P = false, bool ONESEG = false>
__global__ void __launch_bounds__(512, 2) conv3x3_pps_kernel(const ConvParams p, int ntiles, int iters, float act_slope, int nsplit_arg) {
    DSVC_F16_SAT();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MTW = 2, NT = 2;
    constexpr bool F32 = sizeof(CT) == 4;
    constexpr int KCE = CTraits<CT>::KCE, EPV = CTraits<CT>::EPV;     // channels per 64-byte chunk / per 16-byte vector
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wg = wave & 3, gtid = tid & 255;      // group, wave in group, thread in group
    const int r = lane & 31, h = lane >> 5;
    char* lds_w = smem;                                                                   // [2][PS_WROWS] rows of 80 B
    char* lds_a = smem + 2 * PS_WBYTES + g * (PS_NPATCH * ROWB);                           // this group's patch
    float* lds_bias = (float*)(smem + 2 * PS_WBYTES + 2 * PS_NPATCH * ROWB);               // 64 floats
    const rsrc_t rout = make_rsrc(p.out, p.out_bytes);
    const int NCH = p.nchunks;
    // OUTPUT-CHANNEL SPLIT [r4]: a conv with 96 / 128 output channels runs as nsplit = 2 passes of 64 channels inside ONE launch -- workgroup b
    // computes channels [64 (b % nsplit), +64) (the weight buffers are shared by the two groups of a workgroup, so a workgroup owns one pass) and
    // walks the tiles among the gridDim / nsplit workgroups of its pass; the input patch is staged once per pass (L2 / MALL hits at 1/2 - 1/4
    // resolution).  96 channels: the second pass computes 32 real + 32 zero-weight columns (the packed weights are padded to Np rows, rows beyond
    // that are masked) and stores the real ones.
    static_assert(!(EXT && sizeof(CT) == 4), "the extended variant is f16 only");
    const int nsplit = EXT ? nsplit_arg : 1;
    const int csel = EXT ? (int)blockIdx.x % nsplit : 0, wgi = EXT ? (int)blockIdx.x / nsplit : (int)blockIdx.x, nwg = EXT ? (int)gridDim.x / nsplit : (int)gridDim.x;
    const int n0 = EXT ? csel * PS_NTN : 0;
    const rsrc_t rw = make_rsrc(p.w, (unsigned)NCH * 9u * (unsigned)p.Np * 64u);

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
    const bool last_ok = (gtid >> 2) + 64 * (PS_PFV - 1) < PS_NPATCH;
    // pixel offset of this thread's patch vector j inside the patch, in pixels of the image (py * W + px)
    auto patch_pix = [&](int j) { const int pix = (gtid >> 2) + 64 * j, py = pix / PS_PW; return py * W + (pix - py * PS_PW); };
    // border mask of patch vector j of tile t (only evaluated on border tiles: the arithmetic is pinned inside the wave-uniform branch)
    auto border_mask = [&](const TileOrg& t, int gt, int j) -> unsigned {
        const int pix = (gt >> 2) + 64 * j, py = pix / PS_PW, px = pix - py * PS_PW;
        const unsigned iy = (unsigned)(t.oy0 - 1 + py), ix = (unsigned)(t.ox0 - 1 + px);
        return (iy < (unsigned)H && ix < (unsigned)W) ? 0u : OOB;
    };
    auto tile_interior = [&](const TileOrg& t) { return t.oy0 >= 1 && t.oy0 + PS_TH + 1 <= H && t.ox0 >= 1 && t.ox0 + PS_PW - 1 <= W; };
    // ONESEG: everything tile-invariant once
    const ConvSeg& sg0 = p.seg[0];
    const unsigned ldb0 = (unsigned)sg0.ld * (unsigned)sizeof(CT), guard0 = (unsigned)(W + 1) * ldb0;
    const rsrc_t rin0 = make_rsrc((const char*)sg0.ptr - guard0, sg0.bytes + guard0);
    unsigned pvo[ONESEG ? PS_PFV : 1];
    int pp_pix[ONESEG ? 1 : PS_PFV];
    if constexpr (ONESEG) {
#pragma unroll
        for (int j = 0; j < PS_PFV; ++j) pvo[j] = ((unsigned)patch_pix(j) * ldb0 + q * 16) | ((j < PS_PFV - 1 || last_ok) ? 0u : OOB);
    } else {
#pragma unroll
        for (int j = 0; j < PS_PFV; ++j) pp_pix[j] = patch_pix(j);
    }
    // one patch vector: descriptor / pitch / chunk of segment `si` are looked up per call when there are several segments
    auto load_vec = [&](const TileOrg& t, int ch, int j, unsigned dm) -> uint4 {
        const int base_pix = (t.b * H + t.oy0 - 1) * W + t.ox0 - 1;        // scalar; >= -(W + 1): the guard makes the byte offset non-negative
        if constexpr (ONESEG) {
            const unsigned cokm = (ch * KCE + q * EPV < sg0.C) ? 0u : OOB;   // short chunk (C % EPV == 0): the tail reads zeros
            const unsigned so = (unsigned)(base_pix * (int)ldb0) + guard0 + (unsigned)ch * 64u;
            const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rin0, pvo[j] | cokm | dm, so, 0);
            return make_uint4(v.x, v.y, v.z, v.w);
        } else {
            int si = 0;
            if (p.nseg > 1 && ch >= p.seg[1].chunk0) si = 1;
            if (p.nseg > 2 && ch >= p.seg[2].chunk0) si = 2;
            const ConvSeg& sg = p.seg[si];
            const unsigned ldb = (unsigned)sg.ld * (unsigned)sizeof(CT), guard = (unsigned)(W + 1) * ldb;
            const rsrc_t rin = make_rsrc((const char*)sg.ptr - guard, sg.bytes + guard);
            const unsigned cokm = ((ch - sg.chunk0) * KCE + q * EPV < sg.C && (j < PS_PFV - 1 || last_ok)) ? 0u : OOB;
            const unsigned so = (unsigned)(base_pix * (int)ldb) + guard + (unsigned)(ch - sg.chunk0) * 64u;
            const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rin, (__umul24((unsigned)pp_pix[j], ldb) + q * 16) | cokm | dm, so, 0);
            return make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    auto load_chunk = [&](const TileOrg& t, int ch, uint4 (&pf)[PS_PFV]) {
        unsigned dm[PS_PFV];
#pragma unroll
        for (int j = 0; j < PS_PFV; ++j) dm[j] = 0u;
        if (!tile_interior(t)) {                                             // wave-uniform
            int gt = gtid;
            asm volatile("" : "+v"(gt));
#pragma unroll
            for (int j = 0; j < PS_PFV; ++j) dm[j] = border_mask(t, gt, j);
        }
#pragma unroll
        for (int j = 0; j < PS_PFV; ++j) pf[j] = load_vec(t, ch, j, dm[j]);
    };
    // one vector of load_chunk (fp32: the requests are issued from inside the M phase, one per k-step -- see mfma_phase)
    auto load_chunk_one = [&](const TileOrg& t, int ch, int j) -> uint4 {
        unsigned dm = 0u;
        if (!tile_interior(t)) {
            int gt = gtid;
            asm volatile("" : "+v"(gt));
            dm = border_mask(t, gt, j);
        }
        return load_vec(t, ch, j, dm);
    };
    const float pre_sl = p.pre_act == DSVC_ACT_RELU ? 0.f : p.slope;
    auto store_chunk = [&](uint4 (&pf)[PS_PFV]) {
#pragma unroll
        for (int j = 0; j < PS_PFV; ++j) {
            const int pix = (gtid >> 2) + 64 * j;
            if (pix >= PS_NPATCH) continue;
            uint4 v = pf[j];
            if constexpr (EXT) {
                if (p.pre_act != DSVC_ACT_NONE) v = ps_act_h16x8(v, pre_sl);     // ResBlock1's leading ReLU / LeakyReLU on the packed patch (as conv3x3_pp.hip)
            }
            if constexpr (F32) {
                if (p.pre_act != DSVC_ACT_NONE) {        // ResBlock1's leading activation, exactly as the general kernel applies it while staging
                    unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float f = __builtin_bit_cast(float, u[k]);
                        u[k] = __builtin_bit_cast(unsigned, fmaxf(f, f * pre_sl));
                    }
                    v = make_uint4(u[0], u[1], u[2], u[3]);
                }
            }
            *(uint4*)(lds_a + pix * ROWB + q * 16) = v;
        }
    };
    // ---- weight staging: half `g` of a chunk = rows [288 g, 288 g + 288), vector v = gtid + 256 j -> row v >> 2, quarter v & 3 ----
    // (LDS row r = tap * 64 + n of this pass <-> packed row (chunk * 9 + tap) * Np + n0 + n; rows beyond Np read zeros)
    auto load_whalf = [&](int ch, uint4 (&wv)[PS_WV]) {
        if constexpr (!EXT) {
            const unsigned base = ((unsigned)ch * PS_WROWS + (unsigned)g * PS_HALF) * 64u + q * 16u;
#pragma unroll
            for (int j = 0; j < PS_WV; ++j) {
                const unsigned row = (unsigned)(gtid >> 2) + 64u * j;
                wv[j] = buf_load16(rw, (base + row * 64u) | (row < (unsigned)PS_HALF ? 0u : OOB));
            }
        } else {
            // row r_j = 288 g + (gtid >> 2) + 64 j: 64 more rows are exactly one more tap of the same channel, i.e. + Np rows in the packed layout
            const unsigned r0 = (unsigned)g * PS_HALF + (unsigned)(gtid >> 2);
            const unsigned nn = (unsigned)n0 + (r0 & 63u), tstep = (unsigned)p.Np * 64u;
            const unsigned base = (((unsigned)ch * 9u + (r0 >> 6)) * (unsigned)p.Np + nn) * 64u + q * 16u;
            const unsigned nok = nn < (unsigned)p.Np ? 0u : OOB;
#pragma unroll
            for (int j = 0; j < PS_WV; ++j) {
                const unsigned row = (unsigned)(gtid >> 2) + 64u * j;
                wv[j] = buf_load16(rw, (base + (unsigned)j * tstep) | nok | (row < (unsigned)PS_HALF ? 0u : OOB));
            }
        }
    };
    auto load_whalf_one = [&](int ch, int j) -> uint4 {          // (!EXT instances only)
        const unsigned base = ((unsigned)ch * PS_WROWS + (unsigned)g * PS_HALF) * 64u + q * 16u;
        const unsigned row = (unsigned)(gtid >> 2) + 64u * j;
        return buf_load16(rw, (base + row * 64u) | (row < (unsigned)PS_HALF ? 0u : OOB));
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
    // F32_MLOAD (fp32): a wave issuing v_mfma_f32_32x32x2_f32 back to back starves the other wave of its SIMD (tools/pps_phase_probe.py: the partner
    // group's S + E slot stretched to the M phase's length and its natural length -- 4.1 K clocks per step, mostly the issue of 11 vector-memory
    // requests per wave -- came on top), but between two of ITS OWN MFMAs a request issues for free: the matrix pipe is busy for 64 clocks either
    // way.  So the patch and weight requests of this group ride inside its M phase, one per k-step, and the S slot only commits registers to LDS.
    // Prefetch distance: patch(t + 2) is requested in M(t) into the buffer that S(t - 1) emptied and committed in S(t + 1).
    constexpr bool F32_MLOAD = F32 && !EXT;
    auto mfma_phase = [&](int wsel, uint4 (&pf_load)[PS_PFV], uint4 (&wv_load)[PS_WV], const TileOrg& lt, int lch, int wch, bool wload) {          // 18 k-steps (9 taps x 2 halves of the 64-byte chunk), fragments read two k-steps ahead
        const char* wt = lds_w + wsel * PS_WBYTES;
        auto read_frags = [&](int s, uint4 (&af)[MTW], uint4 (&bf)[NT]) {
            const int t = s >> 1, ks = s & 1;
            const int tap_off = ((t / 3) * PS_PW + (t % 3)) * ROWB;
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = *(const uint4*)(wt + t * PS_NTN * ROWB + b_off[n] + ks * 16);
#pragma unroll
            for (int m = 0; m < MTW; ++m) af[m] = *(const uint4*)(lds_a + a_off[m] + tap_off + ks * 16);
        };
        // fragments are read FST - 1 k-steps ahead: two for f16 (one step = 4 MFMAs = 128 cycles does not cover the ds_read latency beside the other
        // group's LDS traffic), one for fp32 (a step is 16 MFMAs = 1024 cycles; the third register set would spill)
        constexpr int FST = 3;
#ifdef DSVC_PPS_NO_INTERLEAVE
        constexpr bool F32_INTERLEAVE = false;
#else
        constexpr bool F32_INTERLEAVE = true;
#endif
        uint4 af[FST][MTW], bf[FST][NT];
        read_frags(0, af[0], bf[0]);
        if constexpr (FST == 3) read_frags(1, af[1], bf[1]);
#ifndef DSVC_PPS_F32_PRIO
#define DSVC_PPS_F32_PRIO 0
#endif
        if constexpr (F32) { if (DSVC_PPS_F32_PRIO) __builtin_amdgcn_s_setprio(DSVC_PPS_F32_PRIO); } else __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int s2 = 0; s2 < 18; ++s2) {
            if (s2 + FST - 1 < 18) read_frags(s2 + FST - 1, af[(s2 + FST - 1) % FST], bf[(s2 + FST - 1) % FST]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (F32 && F32_INTERLEAVE) {
                // consecutive MFMAs on DIFFERENT accumulators (each accumulator still sees j = 0..3 in order: same bits)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(f32x4, bf[s2 % FST][n])[j], __builtin_bit_cast(f32x4, af[s2 % FST][m])[j], acc[m][n], 0, 0, 0);
                        }
                if constexpr (F32_MLOAD) {
                    if (s2 < PS_PFV) pf_load[s2 < PS_PFV ? s2 : 0] = load_chunk_one(lt, lch, s2 < PS_PFV ? s2 : 0);
                    else if (s2 - PS_PFV < PS_WV) { if (wload) wv_load[s2 - PS_PFV < PS_WV ? s2 - PS_PFV : 0] = load_whalf_one(wch, s2 - PS_PFV < PS_WV ? s2 - PS_PFV : 0); }
                }
            } else
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if constexpr (F32) {
                        const f32x4 a4 = __builtin_bit_cast(f32x4, af[s2 % FST][m]), b4 = __builtin_bit_cast(f32x4, bf[s2 % FST][n]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[j], a4[j], acc[m][n], 0, 0, 0);   // D[channel][pixel]
                    } else {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, bf[s2 % FST][n]),   // D[channel][pixel]
                                                                             __builtin_bit_cast(h16x8, af[s2 % FST][m]), acc[m][n], 0, 0, 0);
                    }
                }
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
    auto row_off = [&](const TileOrg& t, int m, bool valid, int half, int ld) -> unsigned {
        const int oy = t.oy0 + wg * MTW + m;
        const int px = (r & 15) + 16 * half;
        const unsigned pix = (unsigned)((t.b * p.Ho + oy) * p.Wo + t.ox0) + px;
        return (__umul24(pix & 0xFFFFFFu, (unsigned)ld * 2) + 2 * n0 + 32 * (r >> 4) + 16 * h) | ((valid && oy < p.Ho && t.ox0 + px < p.Wo) ? 0u : OOB);
    };
    const rsrc_t rres = EXT ? make_rsrc(p.res1 ? p.res1 : p.out, p.res1 ? p.res1_bytes : p.out_bytes) : rout;
    const bool has_r1 = EXT && p.res1 != nullptr;
    const unsigned nmask[NT] = {(!EXT || n0 < p.Cout) ? 0u : OOB, (!EXT || n0 + 32 < p.Cout) ? 0u : OOB};       // an N-tile of zero-weight padding columns is not stored
#define PS_SWAP32(X, Y, LO, HI)                                                                                     \
    { const float xq = (X), yq = (Y);   /* scalars first: hipcc 7.2 mis-casts vector elements */                    \
      const v2u_t sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false); \
      const unsigned lo_ = sw.x, hi_ = sw.y; (LO) = __builtin_bit_cast(float, lo_); (HI) = __builtin_bit_cast(float, hi_); }
#define PS_SWAP16(X, Y, A, B)                                                                                       \
    { const v2u_t sw = __builtin_amdgcn_permlane16_swap((X), (Y), false, false); (A) = sw.x; (B) = sw.y; }
    // ---- fp32 epilogue: D[channel][pixel] as the MFMA leaves it -- lane (r, h) holds, of pixel r of tile row M, the four channels
    // 32 N + 8 G + 4 h .. +3 in accumulator registers 4 G .. 4 G + 3 (G = 0..3): 16-byte vectors already, so (unlike f16, whose 16-byte vector is
    // eight channels) no lane swap is needed; lanes (r, 0) and (r, 1) cover 32 contiguous bytes of the pixel per instruction.
    // (The S / E work of one group does NOT hide under the other group's fp32 MFMAs: a wave issuing v_mfma_f32_32x32x2_f32 back to back
    // starves its SIMD partner -- tools/pps_phase_probe.py: the S + E slot stretches to the M phase's length and its natural length, 4.1 K
    // clocks per step, is added on top -- so every instruction removed here is MFMA time won: 0.807 -> 0.84 of the fp32 peak.)
    const rsrc_t rr1 = make_rsrc(p.res1 ? p.res1 : p.out, p.res1 ? p.res1_bytes : p.out_bytes);
    const rsrc_t rr2 = make_rsrc(p.res2 ? p.res2 : p.out, p.res2 ? p.res2_bytes : p.out_bytes);
    const bool has_res = p.res1 != nullptr || p.res2 != nullptr;
    // residual operands of one 32x32 accumulator tile: 4 16-byte loads per operand, all requested before the first is used
    struct ResRegs { uint4 a[4], b[4]; };
    auto tile_pos = [&](const TileOrg& t, int m, int n, bool valid, unsigned* pm, unsigned* opix, unsigned* c0) {
        const int oy = t.oy0 + wg * MTW + m, ox = t.ox0 + r;
        *pm = (valid && oy < p.Ho && ox < p.Wo) ? 0u : OOB;
        *opix = (unsigned)((t.b * p.Ho + oy) * p.Wo + ox);
        *c0 = (unsigned)(n * 32 + 4 * h);
    };
    auto res_issue = [&](const TileOrg& t, int m, int n, bool valid, ResRegs& R) {
        unsigned pm, opix, c0;
        tile_pos(t, m, n, valid, &pm, &opix, &c0);
        const unsigned e1 = ((opix * (unsigned)p.res1_ld + c0) * 4u) | pm | (p.res1 ? 0u : OOB);
        const unsigned e2 = ((opix * (unsigned)p.res2_ld + c0) * 4u) | pm | (p.res2 ? 0u : OOB);
#pragma unroll
        for (int G = 0; G < 4; ++G) { R.a[G] = buf_load16(rr1, e1 + 32u * G); R.b[G] = buf_load16(rr2, e2 + 32u * G); }
    };
    // R1P (fp32, ONE residual operand -- ResBlock1.conv2 + x, 22 launches per frame): half the operand registers per tile, so the NEXT 32x32
    // tile's loads are in flight behind the current tile's arithmetic and stores (two sets, same register count as one two-operand set)
    auto res1_issue = [&](const TileOrg& t, int m, int n, bool valid, uint4 (&ra)[4]) {
        unsigned pm, opix, c0;
        tile_pos(t, m, n, valid, &pm, &opix, &c0);
        const unsigned e1 = ((opix * (unsigned)p.res1_ld + c0) * 4u) | pm;
#pragma unroll
        for (int G = 0; G < 4; ++G) ra[G] = buf_load16(rr1, e1 + 32u * G);
    };
    auto tile_f32 = [&](const TileOrg& t, const f32x16& a, int m, int n, bool valid, const ResRegs& R) {
        unsigned pm, opix, c0;
        tile_pos(t, m, n, valid, &pm, &opix, &c0);
        const unsigned eo = ((opix * (unsigned)p.out_ld + c0) * 4u) | pm;
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            const float4 bq = *(const float4*)(lds_bias + c0 + 8 * G);
            float v[4] = {a[4 * G] + bq.x, a[4 * G + 1] + bq.y, a[4 * G + 2] + bq.z, a[4 * G + 3] + bq.w};
            // one wave-uniform branch per vector, not per element (per-element selects compiled to 128 scalar compare + branch pairs per tile).
            // max(x, +0) and max(x, x * slope), 0 <= slope <= 1, are the general kernel's x > 0 ? x : 0 / x > 0 ? x : x * slope bit for bit,
            // zero signs included (-0 -> +0 under ReLU, -0 * slope = -0 under LeakyReLU)
            if (p.act == DSVC_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaxf(v[j], 0.f);
            } else if (p.act == DSVC_ACT_LRELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaxf(v[j], v[j] * p.slope);
            }
            if (R1P || has_res) {       // + (res1 + res2), the general kernel's order (conv_common.h: epilogue_vec); an absent operand reads zeros
                const uint4 f1 = R.a[G], f2 = R.b[G];
                const unsigned u1[4] = {f1.x, f1.y, f1.z, f1.w}, u2[4] = {f2.x, f2.y, f2.z, f2.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float r1 = __builtin_bit_cast(float, u1[j]) + 0.f, r2 = __builtin_bit_cast(float, u2[j]) + 0.f;
                    v[j] += r1 + r2;
                }
            }
            ps_store16(rout, eo + 32u * G, make_uint4(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])));
        }
    };
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
        if constexpr (EXT) if (has_r1) {      /* the residual arrives in the MEMORY layout; the (involutive) 16-lane swap brings it to the compute layout (conv3x3_pp.hip) */ \
            const uint4 ra = rres_v[(M * NT + N) * 2], rb = rres_v[(M * NT + N) * 2 + 1];                           \
            const unsigned ma[4] = {ra.x, ra.y, ra.z, ra.w}, mb[4] = {rb.x, rb.y, rb.z, rb.w};                      \
            unsigned xa[4], xb[4];                                                                                  \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) PS_SWAP16(ma[d], mb[d], xa[d], xb[d])                     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
                v0[2 * j] += h16_lo(xa[j]); v0[2 * j + 1] += h16_hi(xa[j]);                                         \
                v1[2 * j] += h16_lo(xb[j]); v1[2 * j + 1] += h16_hi(xb[j]);                                         \
            }                                                                                                       \
        }                                                                                                           \
        unsigned s0[4], s1[4];                                                                                      \
        _Pragma("unroll") for (int d = 0; d < 4; ++d)                                                               \
            PS_SWAP16(pk_h16(v0[2 * d], v0[2 * d + 1]), pk_h16(v1[2 * d], v1[2 * d + 1]), s0[d], s1[d])             \
        ps_store16(rout, (o_out[M][0] + 64 * N) | nmask[N], make_uint4(s0[0], s0[1], s0[2], s0[3]));                \
        ps_store16(rout, (o_out[M][1] + 64 * N) | nmask[N], make_uint4(s1[0], s1[1], s1[2], s1[3]));                \
    }

    // ---- XCD-aware tile walk (conv3x3_pp.hip) ----
    // (with nsplit = 2 the workgroups of a band still share an XCD: blockIdx = 2 wgi + csel, and wgi = band (mod 8) fixes blockIdx mod 8)
    const int nband = (nwg % 8 == 0) ? 8 : 1, per = nwg / nband;
    const int band = wgi % nband, band_tiles = (ntiles + nband - 1) / nband;
    const int lt0 = (wgi / nband) * 2 + g;
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
            if constexpr (!EXT) w0[j] = buf_load16(rw, (row * 64u + q4 * 16u) | (row < (unsigned)PS_WROWS ? 0u : OOB));
            else {      // rows (tid >> 2) + 128 j: two more taps per step
                const unsigned r0 = (unsigned)(tid >> 2), nn = (unsigned)n0 + (r0 & 63u);
                const unsigned off = ((r0 >> 6) + 2u * j) * (unsigned)p.Np * 64u + nn * 64u + q4 * 16u;
                w0[j] = buf_load16(rw, off | ((row < (unsigned)PS_WROWS && nn < (unsigned)p.Np) ? 0u : OOB));
            }
        }
        load_whalf(1 % NCH, wv);
        if (tid < PS_NTN) lds_bias[tid] = (p.bias && n0 + tid < p.Cout) ? p.bias[n0 + tid] : 0.f;
        store_chunk(pfA);                                  // the patch loads were issued first and return first
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const int row = (tid >> 2) + 128 * j;
            if (row < PS_WROWS) *(uint4*)(lds_w + row * ROWB + q4 * 16) = w0[j];
        }
        if (g == 1) store_whalf(1, wv);                    // (wave-uniform branch; group 0 keeps its half of weights(1) in wv)
    }
    if constexpr (!F32_MLOAD) { load_chunk(lorg, lc, pfA); advance_load(); }            // patch(2) (fp32: requested inside M(0))
    // weight stream position of this group: after M(t) it commits half g of weights(t + 1 + g) and requests weights(t + 2 + g)
    int wc = (1 + g) % NCH;                                // chunk of the half held in wv (group 0) / to hold next
    if (g == 1) { wc = 2 % NCH; load_whalf(wc, wv); }      // group 1: wv <- its half of weights(2)
    lds_barrier();                       // weights(0), bias, group 1's half of weights(1) and both groups' first patches are in place
    if (g) lds_barrier();                // group 1 runs one phase behind group 0

    int c = 0, k = 0;
    const int steps = iters * NCH;
    // one stream step; PFX = the register buffer that holds patch(t + 1) (and is refilled with patch(t + 3))
#ifdef DSVC_PPS_PROBE
    unsigned long long psp[5] = {0, 0, 0, 0, 0};
    const unsigned long long pk_t0 = __builtin_amdgcn_s_memtime(), pk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto step = [&](int t, uint4 (&pf_next)[PS_PFV], uint4 (&pf_other)[PS_PFV]) {
#ifdef DSVC_PPS_PROBE
        unsigned long long q0, q1, q2, q3, q4, q5;
#endif
        PPS_STAMP(q0)
        if (c == 0) init_acc();
        if constexpr (F32_MLOAD) {
            int wnext = wc;
            if (t > 0 && ++wnext == NCH) wnext = 0;
            mfma_phase(t & 1, pf_other, wv, lorg, lc, wnext, t > 0);      // ... and requests patch(t + 2) and (from the second step on) the next weight half
            advance_load();
            wc = wnext;
        } else {
            mfma_phase(t & 1, pf_other, wv, lorg, lc, 0, false);
        }
        PPS_STAMP(q1)
        lds_barrier();
        PPS_STAMP(q2)
        PPS_ADD(0, q0, q1) PPS_ADD(1, q1, q2)
        if (c == NCH - 1) {              // ---- epilogue of the finished tile (wave-uniform branch) ----
            const bool valid = tile_of(k) < ntiles;
            if constexpr (F32) {
                ResRegs RA;              // (two sets -- the next tile's operands in flight behind the current one -- spilled: 256 VGPRs + 184 B of scratch)
                if constexpr (R1P) {
                    ResRegs RB;
#pragma unroll
                    for (int i = 0; i < 4; ++i) RA.b[i] = RB.b[i] = make_uint4(0u, 0u, 0u, 0u);      // the absent second operand reads zeros (constants, no registers)
                    res1_issue(cur, 0, 0, valid, RA.a);
                    res1_issue(cur, 0, 1, valid, RB.a);
                    tile_f32(cur, acc[0][0], 0, 0, valid, RA);
                    res1_issue(cur, 1, 0, valid, RA.a);
                    tile_f32(cur, acc[0][1], 0, 1, valid, RB);
                    res1_issue(cur, 1, 1, valid, RB.a);
                    tile_f32(cur, acc[1][0], 1, 0, valid, RA);
                    tile_f32(cur, acc[1][1], 1, 1, valid, RB);
                } else {
                if (has_res) res_issue(cur, 0, 0, valid, RA);
                tile_f32(cur, acc[0][0], 0, 0, valid, RA);
                if (has_res) res_issue(cur, 0, 1, valid, RA);
                tile_f32(cur, acc[0][1], 0, 1, valid, RA);
                if (has_res) res_issue(cur, 1, 0, valid, RA);
                tile_f32(cur, acc[1][0], 1, 0, valid, RA);
                if (has_res) res_issue(cur, 1, 1, valid, RA);
                tile_f32(cur, acc[1][1], 1, 1, valid, RA);
                }
            } else {
                unsigned o_out[MTW][2];
                uint4 rres_v[EXT ? MTW * NT * 2 : 1];
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        o_out[m][half] = row_off(cur, m, valid, half, p.out_ld);
                        if constexpr (EXT) {
                            // one f16 residual (ResBlock1.conv2 of the 128-channel bottleneck blocks): all eight vectors requested before the first is used
                            const unsigned rb = has_r1 ? row_off(cur, m, valid, half, p.res1_ld) : OOB;
#pragma unroll
                            for (int n = 0; n < NT; ++n) rres_v[(m * NT + n) * 2 + half] = buf_load16(rres, (rb + 64 * n) | nmask[n]);
                        }
                    }
                PS_TILE(0, 0) PS_TILE(0, 1) PS_TILE(1, 0) PS_TILE(1, 1)
            }
            c = 0; ++k;
            cur = org_of(k);
            PPS_STAMP(q3)
            PPS_ADD(4, q2, q3)
        } else {
            ++c;
        }
        // ---- S: patch(t + 1) into the group's LDS patch, request patch(t + 3); commit / request this group's weight halves ----
        store_chunk(pf_next);
        if constexpr (!F32_MLOAD) { load_chunk(lorg, lc, pf_next); advance_load(); }
        store_whalf((t + 1 + g) & 1, wv);
        if constexpr (!F32_MLOAD) {
            if (++wc == NCH) wc = 0;
            load_whalf(wc, wv);
        }
        PPS_STAMP(q4)
        lds_barrier();
        PPS_STAMP(q5)
        PPS_ADD(2, q2, q4) PPS_ADD(3, q4, q5)
    };
    for (int t = 0; t < steps; t += 2) {
        step(t, pfB, pfA);
        if (t + 1 < steps) step(t + 1, pfA, pfB);
    }
    if (!g) lds_barrier();               // matches group 1's leading barrier
#ifdef DSVC_PPS_PROBE
    if (wg == 0 && lane == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd(&g_pps_phase[g * 8 + i], psp[i]);
        atomicAdd(&g_pps_phase[g * 8 + 5], __builtin_amdgcn_s_memrealtime() - pk_r0);
        atomicAdd(&g_pps_phase[g * 8 + 6], __builtin_amdgcn_s_memtime() - pk_t0);
        if (g == 0) atomicAdd(&g_pps_phase[16], (unsigned long long)steps);
    }
#endif
#undef PS_TILE
#undef PS_SWAP16
#undef PS_SWAP32
}

template <typename CT, bool ACT, bool EXT = false, bool R1P = false, bool ONESEG = false>
int launch_pps(hipStream_t st, const ConvParams& p, int ntiles, float act_slope, int nsplit = 1) {
    if constexpr (!ONESEG) {
        if (p.nseg == 1 && g_dsvc_opt.pps_oneseg) return launch_pps<CT, ACT, EXT, R1P, true>(st, p, ntiles, act_slope, nsplit);
    }
    const size_t lds = (size_t)2 * PS_WBYTES + (size_t)2 * PS_NPATCH * ROWB + (size_t)PS_NTN * sizeof(float);
    static bool attr_set = false;
    auto kern = conv3x3_pps_kernel<CT, ACT, EXT, R1P, ONESEG>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { dsvc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return DSVC_E_LAUNCH; }
        attr_set = true;
    }
    int nwg = (ntiles + 1) / 2;                                   // workgroups per output-channel pass
    if (nwg > 256 / nsplit) nwg = 256 / nsplit;                   // one persistent workgroup per CU, two tiles in flight each
    const int nband = nwg % 8 == 0 ? 8 : 1, band_tiles = (ntiles + nband - 1) / nband;   // as in the kernel
    const int iters = (band_tiles + 2 * (nwg / nband) - 1) / (2 * (nwg / nband));
    hipLaunchKernelGGL(kern, dim3(nwg * nsplit), dim3(512), lds, st, p, ntiles, iters, act_slope, nsplit);
    DSVC_CHECK_LAUNCH("conv3x3_pps_kernel");
    return DSVC_OK;
}

bool ps_vec16_ok(const void* ptr, int ld, int esz) { return ((uintptr_t)ptr & 15) == 0 && ((int64_t)ld * esz) % 16 == 0; }

}  // namespace

// returns 1 when the shape is not covered (the caller falls through to the general kernel), else DSVC_OK / error
int dsvc_conv3x3_pps_try(hipStream_t st, const ConvParams& p_in, const dsvc_conv_desc* d) {
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->shuffle != 1) return 1;
    if (p_in.nchunks > 32 || d->res1_scale || d->post_weight) return 1;
    const bool f32 = d->compute == DSVC_F32;
    // 64 output channels: 3+ chunks (two chunks are conv3x3_pp.hip's, weights resident); f16 also 96 / 128 output channels as two passes of 64
    // (measured, tools/bench_pps.py, us streamed / general: 128->128 @272x480 45 / 50; 96->96 @544x960 114 / 116 and @272x480 36 / 32 -- the second
    //  pass computes 32 padding columns; 64->128 @272x480 29 / 30: only the first is on by default, pps = 2 forces the others for the tests)
    const bool split_ok = !f32 && ((d->Cout == 128 && p_in.nchunks >= 4) || (g_dsvc_opt.pps >= 2 && (d->Cout == 96 || d->Cout == 128) && p_in.nchunks >= 2));
    const int nsplit = d->Cout == 64 ? 1 : (split_ok ? 2 : 0);
    if (nsplit == 0 || (nsplit == 1 && p_in.nchunks < (f32 ? 2 : 3))) return 1;     // (fp32 has no resident-weights sibling: two chunks -- 32 -> 64 -- stream as well)
    const int esz = f32 ? 4 : 2, epv = f32 ? 4 : 8, dt = f32 ? DSVC_F32 : DSVC_F16;
    for (int s = 0; s < p_in.nseg; ++s) {                          // every pixel's chunk row is an aligned run of 16-byte vectors
        const ConvSeg& sg = p_in.seg[s];
        if (sg.dtype != dt || (sg.C % epv) != 0 || !ps_vec16_ok(sg.ptr, sg.ld, esz)) return 1;
    }
    if (d->out_dtype != dt || !ps_vec16_ok(d->out, d->out_ld, esz)) return 1;
    auto act_ok = [&](int a) { return a == DSVC_ACT_NONE || a == DSVC_ACT_RELU || (a == DSVC_ACT_LRELU && d->slope >= 0.f && d->slope <= 1.f); };
    if (!act_ok(d->act) || !act_ok(d->pre_act)) return 1;
    if (f32) {                                                     // fp32 (parity) mode: pre-activation and fp32 residuals are run-time parameters
        if (!g_dsvc_opt.pps_f32) return 1;
        if (d->res1 && (d->res1_dtype != DSVC_F32 || !ps_vec16_ok(d->res1, d->res1_ld, 4))) return 1;
        if (d->res2 && (d->res2_dtype != DSVC_F32 || !ps_vec16_ok(d->res2, d->res2_ld, 4))) return 1;
        // (two residual operands: 1315 vs 1328 us in the general kernel since the requests moved into the M phase -- taken too)
    } else {                                                       // f16: pre-activation and ONE f16 residual are run-time parameters
        if (d->res2) return 1;
        if (d->res1 && (d->res1_dtype != DSVC_F16 || !ps_vec16_ok(d->res1, d->res1_ld, 2))) return 1;
    }
    ConvParams p = p_in;
    p.tiles_x = (p.Wo + 31) / 32;
    p.tiles_y = (p.Ho + PS_TH - 1) / PS_TH;
    const int ntiles = p.B * p.tiles_x * p.tiles_y;
    if ((int64_t)p.B * p.H * p.W >= (1 << 24)) return 1;          // pixel indices go through the 24-bit multiplier
    if (ntiles * nsplit < 1000) return 1;                          // small images: the general kernel's finer tiling fills the chip better
    if (f32) return (d->res1 && !d->res2 && g_dsvc_opt.pps_f32 != 3) ? launch_pps<float, false, false, true>(st, p, ntiles, 1.f) : launch_pps<float, false>(st, p, ntiles, 1.f);
    const float act_slope = d->act == DSVC_ACT_NONE ? 1.f : (d->act == DSVC_ACT_RELU ? 0.f : d->slope);
    if (nsplit > 1 || d->res1 || d->pre_act != DSVC_ACT_NONE)          // output-channel split / pre-activation / residual: the extended instantiation
        return d->act == DSVC_ACT_NONE ? launch_pps<h16, false, true>(st, p, ntiles, 1.f, nsplit) : launch_pps<h16, true, true>(st, p, ntiles, act_slope, nsplit);
    return d->act == DSVC_ACT_NONE ? launch_pps<h16, false>(st, p, ntiles, 1.f) : launch_pps<h16, true>(st, p, ntiles, act_slope);
}

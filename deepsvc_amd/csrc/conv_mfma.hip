// conv_mfma.hip -- the convolution family of the DeepSVC hot path as ONE implicit-GEMM MFMA kernel.
//
// Replaces every nn.Conv2d / nn.ConvTranspose2d + following pointwise op of the reference
// (modules.py:127-131,176-178,200-204,237-242,253,307-311,322-340,364,382-418;
//  image_model.py:25-27,53-146).  See DESIGN.md section 4.1.
//
// Mapping (gfx950, wave64):
//   GEMM M = output pixels, N = output channels, K = taps x input channels.
//   A workgroup (4 waves) owns a tile of (4*MT) x 32 output pixels and NT*32 output channels.
//   Wave w owns tile rows [w*MT, w*MT+MT): each row of 32 pixels is one MFMA M-tile.
//   K is walked in 64-byte "chunks" of input channels (16 fp32 / 32 f16).  For one chunk the
//   input HALO PATCH ((TH-1)*s+kh) x ((TW-1)*s+kw) pixels is staged ONCE into LDS and every tap
//   reads it at a shifted pixel offset -- HBM/L2 sees each input element once per tile, not kh*kw times.
//   LDS rows (one pixel of the patch, or one output channel of a tap's weights) are 64 B of data
//   + 16 B pad = 80 B: consecutive rows advance 20 banks, which makes the 16-lane groups of
//   ds_read_b128 conflict-free (MI355X_MICROARCH.md, LDS).  A lane reads 16 B: lane (r=l&31,h=l>>5)
//   takes bytes [32h + 16ks, +16) of row r.  The same byte->k assignment is used for A and B, so
//   the K order inside a chunk is a free permutation.
//   fp32 compute: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain), 4 MFMAs per 16-B fragment pair.
//   f16 compute: v_mfma_f32_32x32x16_f16, 1 MFMA per 16-B fragment pair, fp32 accumulate.
//   Epilogue (fused): + bias, activation, + residual (optionally gated per channel: SELayer),
//   + second residual, optional PixelShuffle(2) scatter, cast to the output storage type.
#include <cstdlib>
#include <type_traits>
#include "conv_common.h"

int dsvc_conv3x3_ws_try(hipStream_t st, const dsvc_conv::ConvParams& p, const dsvc_conv_desc* d);
int dsvc_conv1x1_try(hipStream_t st, const dsvc_conv::ConvParams& p, const dsvc_conv_desc* d);
int dsvc_conv_head_f32_try(hipStream_t st, const dsvc_conv::ConvParams& p, const dsvc_conv_desc* d);
int dsvc_conv1x1_f32_try(hipStream_t st, const dsvc_conv::ConvParams& p, const dsvc_conv_desc* d);
int dsvc_conv3x3_pp_try(hipStream_t st, const dsvc_conv::ConvParams& p, const dsvc_conv_desc* d);
int dsvc_conv_head_try(hipStream_t st, const dsvc_conv::ConvParams& p, const dsvc_conv_desc* d);
int dsvc_conv3x3_pps_try(hipStream_t st, const dsvc_conv::ConvParams& p, const dsvc_conv_desc* d);
int dsvc_conv_headtap_try(hipStream_t st, const dsvc_conv::ConvParams& p, const dsvc_conv_desc* d);
int dsvc_conv_cin8_try(hipStream_t st, const dsvc_conv::ConvParams& p, const dsvc_conv_desc* d);
int dsvc_conv7x7_rr_try(hipStream_t st, const dsvc_conv::ConvParams& p, const dsvc_conv_desc* d);

#ifdef DSVC_PHASE_PROBE
// Build with DSVC_BUILD_DEFINES=-DDSVC_PHASE_PROBE (deepsvc_amd/build.py) to get per-phase shader-clock totals of wave 0 of
// every workgroup of conv_mfma_kernel (tools/phase_probe.py reads them): [0] barrier before staging, [1] patch staging,
// [2] weight staging, [3] barrier before the MFMAs, [4] MFMA phase, [5] epilogue, [6] whole kernel, [7] workgroups.
__device__ unsigned long long g_phase[8];
#define PROBE_STAMP(var) unsigned long long var = 0; if (probe) { var = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   /* no vmcnt wait: prefetches stay in flight */
#define PROBE_ADD(i, a, b) if (probe) ph[i] += (b) - (a);
extern "C" int dsvc_dbg_phase_read(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_phase), sizeof(g_phase)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define PROBE_FINISH                                                                                   \
    { PROBE_STAMP(t_end)                                                                               \
      if (probe) { ph[5] += t_end - t_epi;                                                             \
                   for (int i = 0; i < 6; ++i) atomicAdd(&g_phase[i], ph[i]);                          \
                   atomicAdd(&g_phase[6], t_end - t_begin); atomicAdd(&g_phase[7], 1ull); } }
#else
#define PROBE_STAMP(var)
#define PROBE_ADD(i, a, b)
#define PROBE_FINISH
#endif

#ifdef DSVC_PP_CLOCK
// Diagnostic build only (tools/pp_clock.py): in-kernel shader clock of the general kernel, as in conv3x3_pp.hip -- wave 0 of every
// workgroup stamps s_memtime / s_memrealtime around its channel-chunk loop; [0] += d(memtime), [1] += d(memrealtime), [2] += 1.
__device__ unsigned long long g_mfma_clock[3];
extern "C" int dsvc_dbg_mfma_clock_read(unsigned long long* out3, int reset) {
    if (hipMemcpyFromSymbol(out3, HIP_SYMBOL(g_mfma_clock), sizeof(g_mfma_clock)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[3] = {0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_clock), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

namespace {
using namespace dsvc_conv;

// REGEPI: the MFMA operands are swapped (A = weights, B = pixels), so a lane ends up with 16 output channels of ONE pixel
// and the epilogue needs no LDS: one 32-lane swap turns them into two runs of 8 consecutive channels = two 16-byte f16
// stores (the slab epilogue -- accumulators through LDS, two extra barriers per pass -- is ~20 % of a workgroup's time on
// the large convs, tools/phase_probe.py).  Taken for f16 output, whole 32-channel tiles, act in {none, relu, lrelu},
// at most one f16 residual; everything else keeps the slab epilogue.
// PF > 0 (with REGEPI, f16 storage, one staging batch per chunk): the input patch of the NEXT channel chunk is prefetched
// into PF registers per thread during the last MFMA phase of the current chunk, like the weights.
template <typename CT, int MT, int NT, bool FAST, bool REGEPI = false, int PF = 0>
__global__ void __launch_bounds__(256, 2) conv_mfma_kernel(const ConvParams p) {
    DSVC_F16_SAT();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KCE = CTraits<CT>::KCE, EPV = CTraits<CT>::EPV;
    constexpr int TH = 4 * MT, NTN = NT * 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int ng = blockIdx.y;
    const unsigned b_lo = (unsigned)(b % p.b_inner), b_hi = (unsigned)(b / p.b_inner);   // dense batch: (b, 0)

    const int oy0 = ty * TH, ox0 = tx * TW;
    const int PH = (TH - 1) * p.stride + p.kh, PW = (TW - 1) * p.stride + p.kw;
    const int iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
    const int npatch = PH * PW;
    // LDS image of the patch: row-major pixels at the 80-B pitch.  For stride 2 the even and the odd columns of a patch
    // row are stored apart (pitch PWL = 2 * ceil(PW/2)), so that the fragment of tap kx -- input columns 2r + kx,
    // r = lane & 31 -- is a UNIT-stride run again: with interleaved columns the 160-B lane stride made every
    // ds_read_b128 a two-way bank conflict (5x5 s2 analysis convs ran at ~190 TFLOP/s).
    const bool s2 = p.stride == 2;
    const int PWh = (PW + 1) >> 1, PWL = s2 ? 2 * PWh : PW;
    auto lds_row = [&](int py, int px) { return py * PWL + (s2 ? (px & 1) * PWh + (px >> 1) : px); };
    char* lds_a = smem;
    char* lds_w = smem + (size_t)PH * PWL * ROWB;
    const int ntaps = p.kh * p.kw;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    // per-thread staging walk: vector q of patch pixel (py,px), advancing 64 pixels per step without divisions
    const int q = tid & 3;
    const int pix0 = tid >> 2;
    const int py0 = pix0 / PW, px0 = pix0 - py0 * PW;
    const int dpy = 64 / PW, dpx = 64 - dpy * PW;
    // fragment read offsets: lane (r,h) takes bytes [32h + 16ks, +16) of row r
    int a_off[MT], b_off[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_off[m] = (((wave * MT + m) * p.stride) * PWL + r) * ROWB + h * 32;
#pragma unroll
    for (int n = 0; n < NT; ++n) b_off[n] = (n * 32 + r) * ROWB + h * 32;

    // ---- weight prefetch: the packed weights of the NEXT staging step travel L2 -> registers while the MFMAs of the
    // current one run (a staging step is one L2 round trip of ~2.5 k clocks with nothing else to do in this workgroup:
    // 30 % of its time, tools/phase_probe.py).  WP = vectors per thread that fit the register budget of the variant.
    // (fp32 mode too since round 3: its in-kernel clock is 2.38 GHz -- not power-limited -- so exposed staging latency is lost MFMA time)
    // (not for fp32 with 128 output channels per group: 128 accumulator registers + the prefetch vectors spill)
    constexpr int WP = (MT == 4 || (sizeof(CT) == 4 && NT == 4)) ? 0 : (NT == 1 ? 5 : NT == 2 ? 9 : NT == 3 ? 5 : 6);
    uint4 wv[WP > 0 ? WP : 1];
    const rsrc_t rw = make_rsrc((const char*)p.w + (size_t)b * p.w_bs, p.w_bytes);
    const bool wfull = (ng + 1) * NTN <= p.Np;
    auto w_issue = [&](int ch2, int tg2, bool any) {       // any = false: past the last step, every lane masked (no branch: a
        if constexpr (WP > 0) {                            // conditional call would be a merge point, where hipcc drains vmcnt)
            const unsigned wbase = (((unsigned)ch2 * ntaps + tg2) * p.Np + (unsigned)ng * NTN) * 64u + q * 16u;
            const int nrows = min(p.tapg, ntaps - tg2) * NTN;
#pragma unroll
            for (int j = 0; j < WP; ++j) {
                const int row = (tid >> 2) + 64 * j;
                const int t = row / NTN, n = row - t * NTN;
                const bool ok = any && row < nrows && (wfull || ng * NTN + n < p.Np);
                wv[j] = buf_load16(rw, (wbase + ((unsigned)t * p.Np + (unsigned)n) * 64u) | (ok ? 0u : OOB));
            }
        }
    };
    w_issue(0, 0, true);
    uint4 pv[PF > 0 ? PF : 1];
    auto p_issue = [&](int ch2, bool any) {          // same-type (f16) storage only; every lane masked when !any (branch-free)
        if constexpr (PF > 0) {
            int s2i = 0;
            if (p.nseg > 1 && ch2 >= p.seg[1].chunk0) s2i = 1;
            if (p.nseg > 2 && ch2 >= p.seg[2].chunk0) s2i = 2;
            const ConvSeg& sg2 = p.seg[s2i];
            const int c02 = (ch2 - sg2.chunk0) * KCE;
            const rsrc_t rs = make_rsrc(sg2.ptr, sg2.bytes);
            const unsigned pitch = (unsigned)sg2.ld * sizeof(CT);
            const unsigned img0 = b_lo * sg2.bs_lo + b_hi * sg2.bs_hi;
            const unsigned img = p.s2d ? img0 + (unsigned)(ch2 - sg2.chunk0) * 16u + ((unsigned)(q >> 1) * 2u * p.W + (unsigned)(q & 1)) * pitch
                                       : img0 + c02 * (unsigned)sizeof(CT) + q * 16;
            const unsigned pixstep = p.s2d ? 2u : 1u, rowstep = p.s2d ? 4u * p.W : (unsigned)p.W;
            const bool qvalid = any && c02 + q * EPV < sg2.C;
            int py = py0, px = px0;
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int iy = iy0 + py, ix = ix0 + px;
                const bool inb = qvalid && pix0 + 64 * j < npatch && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                pv[j] = buf_load16(rs, (img + ((unsigned)iy * rowstep + (unsigned)ix * pixstep) * pitch) | (inb ? 0u : OOB));
                px += dpx; py += dpy;
                if (px >= PW) { px -= PW; ++py; }
            }
        }
    };
    auto p_commit = [&]() {
        if constexpr (PF > 0) {
            const float sl = p.pre_act == DSVC_ACT_RELU ? 0.f : (p.pre_act == DSVC_ACT_NONE ? 1.f : p.slope);
            int py = py0, px = px0;
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                uint4 v = pv[j];
                if (p.pre_act != DSVC_ACT_NONE) {
                    unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if constexpr (sizeof(CT) == 4) {
                            const float f = __builtin_bit_cast(float, u[k]);
                            u[k] = __builtin_bit_cast(unsigned, fmaxf(f, f * sl));
                        } else {
                            const float lo = h16_lo(u[k]), hi = h16_hi(u[k]);
                            u[k] = pk_h16(fmaxf(lo, lo * sl), fmaxf(hi, hi * sl));
                        }
                    }
                    v = make_uint4(u[0], u[1], u[2], u[3]);
                }
                if (pix0 + 64 * j < npatch) *(uint4*)(lds_a + lds_row(py, px) * ROWB + q * 16) = v;
                px += dpx; py += dpy;
                if (px >= PW) { px -= PW; ++py; }
            }
        }
    };
    p_issue(0, true);
#ifdef DSVC_PHASE_PROBE
    const bool probe = tid == 0;
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
#endif
    PROBE_STAMP(t_begin)
#ifdef DSVC_PP_CLOCK
    const unsigned long long ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr bool SKIPK = !FAST && sizeof(CT) == 4;     // see mfma_tap
    int kreal = 8;
    for (int ch = 0; ch < p.nchunks; ++ch) {
        PROBE_STAMP(t_a)
        int s = 0;
        if (p.nseg > 1 && ch >= p.seg[1].chunk0) s = 1;
        if (p.nseg > 2 && ch >= p.seg[2].chunk0) s = 2;
        const ConvSeg& sg = p.seg[s];
        const int c0 = (ch - sg.chunk0) * KCE;
        if constexpr (SKIPK) kreal = __builtin_amdgcn_readfirstlane((p.skipk && sg.C - c0 < 8) ? sg.C - c0 : 8);

        lds_barrier();  // all waves done with the previous chunk's LDS contents (LDS-only: global traffic stays in flight)
        PROBE_STAMP(t_b)
        PROBE_ADD(0, t_a, t_b)
        // ---- stage the input halo patch of this channel chunk ----
        // Loads of a batch are all issued before the first LDS store so their L2/HBM latencies overlap.
        bool staged = false;
        if constexpr (PF > 0) { p_commit(); staged = true; }
        if constexpr (PF == 0 && FAST && sizeof(CT) == 2) {
            // fp32 storage feeding the f16 MFMA (the entropy scan's fp32 latents): two 16-B loads per quarter, batched like
            // the same-type path, rounded to f16 (RNE, as pack_vec does) on the way into LDS
            if (sg.dtype == DSVC_F32) {
                staged = true;
                constexpr int PC = 4;
                const rsrc_t rs = make_rsrc(sg.ptr, sg.bytes);
                const unsigned pitch = (unsigned)sg.ld * 4u;
                const unsigned img = b_lo * sg.bs_lo + b_hi * sg.bs_hi + (unsigned)(c0 + q * EPV) * 4u;
                const unsigned m0 = c0 + q * EPV < sg.C ? 0u : OOB, m1 = c0 + q * EPV + 4 < sg.C ? 0u : OOB;   // C % 4 == 0
                const float sl = p.pre_act == DSVC_ACT_RELU ? 0.f : (p.pre_act == DSVC_ACT_NONE ? 1.f : p.slope);
                int py = py0, px = px0;
                for (int base = pix0; base < npatch; base += 64 * PC) {
                    uint4 lo[PC], hi[PC];
                    int lr[PC];
#pragma unroll
                    for (int j = 0; j < PC; ++j) {
                        const int iy = iy0 + py, ix = ix0 + px;
                        lr[j] = lds_row(py, px);
                        const bool inb = base + 64 * j < npatch && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                        const unsigned off = (img + ((unsigned)iy * p.W + (unsigned)ix) * pitch) | (inb ? 0u : OOB);
                        lo[j] = buf_load16(rs, off | m0);
                        hi[j] = buf_load16(rs, (off + 16u) | m1 | (inb ? 0u : OOB));
                        px += dpx; py += dpy;
                        if (px >= PW) { px -= PW; ++py; }
                    }
#pragma unroll
                    for (int j = 0; j < PC; ++j) {
                        float f[8] = {__builtin_bit_cast(float, lo[j].x), __builtin_bit_cast(float, lo[j].y),
                                      __builtin_bit_cast(float, lo[j].z), __builtin_bit_cast(float, lo[j].w),
                                      __builtin_bit_cast(float, hi[j].x), __builtin_bit_cast(float, hi[j].y),
                                      __builtin_bit_cast(float, hi[j].z), __builtin_bit_cast(float, hi[j].w)};
                        if (p.pre_act != DSVC_ACT_NONE) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) f[k] = fmaxf(f[k], f[k] * sl);
                        }
                        const uint4 v = pack_vec<CT, 8>(f);
                        if (base + 64 * j < npatch) *(uint4*)(lds_a + lr[j] * ROWB + q * 16) = v;
                    }
                }
            }
        }
        if (staged) {
        } else if constexpr (FAST && PF == 0) {  // storage type == compute type, 16-B aligned, whole chunks
            constexpr int PB = 6;
            const rsrc_t rs = make_rsrc(sg.ptr, sg.bytes);      // out-of-image pixels read as zeros (bounds-checked loads)
            const unsigned pitch = (unsigned)sg.ld * sizeof(CT);
            // plain: quarter q = channels c0 + q*EPV of the pixel.  space-to-depth view: quarter q = image pixel
            // (2Y + q/2, 2X + q%2), channels (chunk index in the segment) * EPV .. of a (2H x 2W) image
            const unsigned img0 = b_lo * sg.bs_lo + b_hi * sg.bs_hi;
            const unsigned img = p.s2d ? img0 + (unsigned)(ch - sg.chunk0) * 16u +
                                             ((unsigned)(q >> 1) * 2u * p.W + (unsigned)(q & 1)) * pitch
                                       : img0 + c0 * (unsigned)sizeof(CT) + q * 16;
            const unsigned pixstep = p.s2d ? 2u : 1u, rowstep = p.s2d ? 4u * p.W : (unsigned)p.W;
            const bool qvalid = c0 + q * EPV < sg.C;           // a short last chunk: the missing 16-B quarters read as zeros
            int py = py0, px = px0;
            for (int base = pix0; base < npatch; base += 64 * PB) {
                uint4 v[PB];
                int lr[PB];
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    const int iy = iy0 + py, ix = ix0 + px;
                    lr[j] = lds_row(py, px);
                    const bool inb = qvalid && base + 64 * j < npatch && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    // (offset | OOB, not a select of offsets: hipcc turns the select into exec-mask branches and then waits for
                    //  ALL outstanding loads at every merge point, which serialises the batch)
                    v[j] = buf_load16(rs, (img + ((unsigned)iy * rowstep + (unsigned)ix * pixstep) * pitch) | (inb ? 0u : OOB));
                    px += dpx; py += dpy;
                    if (px >= PW) { px -= PW; ++py; }
                }
                if (p.pre_act != DSVC_ACT_NONE) {  // ResBlock1's leading ReLU / LeakyReLU, applied on the packed vector
                    const float sl = p.pre_act == DSVC_ACT_RELU ? 0.f : p.slope;
#pragma unroll
                    for (int j = 0; j < PB; ++j) {
                        unsigned u[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if constexpr (sizeof(CT) == 4) {
                                const float f = __builtin_bit_cast(float, u[k]);
                                u[k] = __builtin_bit_cast(unsigned, fmaxf(f, f * sl));
                            } else {
                                const float lo = h16_lo(u[k]), hi = h16_hi(u[k]);
                                u[k] = pk_h16(fmaxf(lo, lo * sl), fmaxf(hi, hi * sl));
                            }
                        }
                        v[j] = make_uint4(u[0], u[1], u[2], u[3]);
                    }
                }
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    if (base + 64 * j < npatch) *(uint4*)(lds_a + lr[j] * ROWB + q * 16) = v[j];
            }
        } else if (sizeof(CT) == 4 && sg.dtype == DSVC_F32 && p.skipk) {
            // element-staged fp32 segments (frames, flows: C % 4 != 0): bounds-checked DWORD loads, masked by the offset and issued four
            // pixels ahead of the first LDS store -- the per-element path below waits out one L2 round trip per pixel (3 -> 64 3x3 at
            // 1088x1920: 271 -> see tools/bench_narrow_f32.py).  Same values, same LDS image.
            constexpr int PBN = 4;
            const rsrc_t rs = make_rsrc(sg.ptr, sg.bytes);
            const unsigned img0 = (unsigned)b * (unsigned)p.H * (unsigned)p.W;
            const int cq = c0 + q * 4;
            // a 16-byte-aligned segment of whole 4-channel quarters beside the narrow ones (the 64-channel segment of 3+64+3 -> 96): one
            // 16-byte request per pixel quarter instead of four dword requests (wave-uniform choice per segment)
            const bool seg_vec = sg.vec_ok && (sg.C % 4) == 0;
            int py = py0, px = px0;
            for (int base = pix0; base < npatch; base += 64 * PBN) {
                unsigned u[PBN][4];
                int lr[PBN];
#pragma unroll
                for (int j = 0; j < PBN; ++j) {
                    const int iy = iy0 + py, ix = ix0 + px;
                    lr[j] = lds_row(py, px);
                    const bool inb = base + 64 * j < npatch && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    const unsigned e0 = ((img0 + (unsigned)iy * p.W + (unsigned)ix) * (unsigned)sg.ld + (unsigned)cq) * 4u;
                    if (seg_vec) {
                        const uint4 t = buf_load16(rs, (inb && cq < sg.C) ? e0 : OOB);
                        u[j][0] = t.x; u[j][1] = t.y; u[j][2] = t.z; u[j][3] = t.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            u[j][e] = __builtin_amdgcn_raw_buffer_load_b32(rs, (inb && cq + e < sg.C) ? e0 + 4u * e : OOB, 0, 0);
                    }
                    px += dpx; py += dpy;
                    if (px >= PW) { px -= PW; ++py; }
                }
                if (p.pre_act != DSVC_ACT_NONE) {
                    const float sl = p.pre_act == DSVC_ACT_RELU ? 0.f : p.slope;
#pragma unroll
                    for (int j = 0; j < PBN; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float f = __builtin_bit_cast(float, u[j][e]);
                            u[j][e] = __builtin_bit_cast(unsigned, fmaxf(f, f * sl));
                        }
                }
#pragma unroll
                for (int j = 0; j < PBN; ++j)
                    if (base + 64 * j < npatch) *(uint4*)(lds_a + lr[j] * ROWB + q * 16) = make_uint4(u[j][0], u[j][1], u[j][2], u[j][3]);
            }
        } else {
            int py = py0, px = px0;
            for (int pix = pix0; pix < npatch; pix += 64) {
                const int iy = iy0 + py, ix = ix0 + px;
                uint4 val = make_uint4(0, 0, 0, 0);
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                    float vals[EPV];
                    load_vec<EPV>(sg, (((int64_t)b * p.H + iy) * p.W + ix) * sg.ld, c0 + q * EPV, vals);
                    if (p.pre_act != DSVC_ACT_NONE) {
                        const float sl = p.pre_act == DSVC_ACT_RELU ? 0.f : p.slope;
#pragma unroll
                        for (int j = 0; j < EPV; ++j) vals[j] = fmaxf(vals[j], vals[j] * sl);
                    }
                    val = pack_vec<CT, EPV>(vals);
                }
                *(uint4*)(lds_a + lds_row(py, px) * ROWB + q * 16) = val;
                px += dpx; py += dpy;
                if (px >= PW) { px -= PW; ++py; }
            }
        }

        PROBE_STAMP(t_c)
        PROBE_ADD(1, t_b, t_c)
        int ky = 0, kx = 0;
        for (int tg0 = 0; tg0 < ntaps; tg0 += p.tapg) {
            const int tcur = min(p.tapg, ntaps - tg0);
            PROBE_STAMP(t_d0)
            if (tg0 > 0) lds_barrier();
            PROBE_STAMP(t_d)
            PROBE_ADD(0, t_d0, t_d)
            // ---- stage the weights of taps [tg0, tg0+tcur) for this chunk and N group (batched loads) ----
            if constexpr (WP > 0) {          // (compile-time choice: a run-time one merges two paths and hipcc then drains vmcnt at the join)
                {
                    const int nrows = tcur * NTN;
#pragma unroll
                    for (int j = 0; j < WP; ++j) {
                        const int row = (tid >> 2) + 64 * j;
                        if (row < nrows) *(uint4*)(lds_w + row * ROWB + q * 16) = wv[j];
                    }
                    const bool more_tg = tg0 + p.tapg < ntaps;
                    w_issue(more_tg ? ch : ch + 1, more_tg ? tg0 + p.tapg : 0, more_tg || ch + 1 < p.nchunks);
                    // the next chunk's patch, behind the weights; issued (masked) at every tap group so the code stays straight-line --
                    // VMEM returns in order, the real load of the last group lands last
                    p_issue(ch + 1, !more_tg && ch + 1 < p.nchunks);
                }
            } else {
                // bounds-checked buffer loads, masked by the offset: a conditional global load per vector made hipcc wait for every
                // load before issuing the next (s_waitcnt vmcnt(0) at each merge) -- nine serial L2 round trips per staging step,
                // 30 % of a workgroup's time (tools/phase_probe.py)
                constexpr int WB = 9;
                const unsigned wbase = (((unsigned)ch * ntaps + tg0) * p.Np + (unsigned)ng * NTN) * 64u + q * 16u;
                const int nrows = tcur * NTN;
                for (int base = tid >> 2; base < nrows; base += 64 * WB) {
                    uint4 v[WB];
#pragma unroll
                    for (int j = 0; j < WB; ++j) {
                        const int row = base + 64 * j;
                        const int t = row / NTN, n = row - t * NTN;
                        const bool ok = row < nrows && (wfull || ng * NTN + n < p.Np);
                        v[j] = buf_load16(rw, (wbase + ((unsigned)t * p.Np + (unsigned)n) * 64u) | (ok ? 0u : OOB));
                    }
#pragma unroll
                    for (int j = 0; j < WB; ++j)
                        if (base + 64 * j < nrows) *(uint4*)(lds_w + (base + 64 * j) * ROWB + q * 16) = v[j];
                }
            }
            PROBE_STAMP(t_e)
            PROBE_ADD(2, t_d, t_e)
            lds_barrier();
            PROBE_STAMP(t_f)
            PROBE_ADD(3, t_e, t_f)

            // ---- MFMA phase, software-pipelined over taps: while the MFMAs of one tap run, the LDS reads of the next
            // tap are already in flight.  Two named fragment sets (static register indexing), branch-free loop body
            // (the trailing prefetch re-reads the last tap), scheduling barriers pin the read-before-MFMA order.
            struct Frags { uint4 a[2][MT], b[2][NT]; };
            int rt = 0;                                   // next tap (within the staged group) to read
            auto read_tap = [&](Frags& f) {
                const int tap_off = (ky * PWL + (s2 ? (kx & 1) * PWh + (kx >> 1) : kx)) * ROWB;
                const char* wt = lds_w + rt * NTN * ROWB;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) f.b[ks][n] = *(const uint4*)(wt + b_off[n] + ks * 16);
#pragma unroll
                    for (int m = 0; m < MT; ++m) f.a[ks][m] = *(const uint4*)(lds_a + a_off[m] + tap_off + ks * 16);
                }
                if (rt + 1 < tcur) { ++rt; if (++kx == p.kw) { kx = 0; ++ky; } }
            };
            auto mfma_tap = [&](const Frags& f) {
                if constexpr (SKIPK) {
                    // element-staged fp32 launches carry 2- / 3-channel segments (frames, flows): MFMA 4 ks + j of a chunk multiplies channels
                    // 4 ks + j and 8 + 4 ks + j -- all zeros from channel `kreal` on; an all-zero fma step leaves the accumulator as it is, so
                    // skipping it keeps the bits (3 of 8 MFMAs per tap for a 3-channel segment)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (4 * ks + j < kreal) {
#pragma unroll
                                for (int m = 0; m < MT; ++m)
#pragma unroll
                                    for (int n = 0; n < NT; ++n)
                                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(f32x4, f.a[ks][m])[j], __builtin_bit_cast(f32x4, f.b[ks][n])[j], acc[m][n], 0, 0, 0);
                            }
                    return;
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            if constexpr (sizeof(CT) == 4) {
                                const f32x4 a4 = __builtin_bit_cast(f32x4, f.a[ks][m]);
                                const f32x4 b4 = __builtin_bit_cast(f32x4, f.b[ks][n]);
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b4[j], acc[m][n], 0, 0, 0);
                            } else {
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    __builtin_bit_cast(h16x8, REGEPI ? f.b[ks][n] : f.a[ks][m]),
                                    __builtin_bit_cast(h16x8, REGEPI ? f.a[ks][m] : f.b[ks][n]), acc[m][n], 0, 0, 0);
                            }
                        }
            };
            if constexpr (MT == 4 || (NT == 4 && sizeof(CT) == 2)) {
                // 16-row tiles (and 128-channel groups, to make room for the weight prefetch): 128+ accumulator registers leave no room for two whole taps of fragments; pipeline at the
                // granularity of one 16-deep k-step (6-7 reads ahead of 8-12 MFMAs) instead
                struct FragsK { uint4 a[MT], b[NT]; };
                auto read_step = [&](FragsK& f, int ks) {          // ks is a literal at every call site: straight-line code
                    const int tap_off = (ky * PWL + (s2 ? (kx & 1) * PWh + (kx >> 1) : kx)) * ROWB + ks * 16;
                    const char* wt = lds_w + rt * NTN * ROWB + ks * 16;
#pragma unroll
                    for (int n = 0; n < NT; ++n) f.b[n] = *(const uint4*)(wt + b_off[n]);
#pragma unroll
                    for (int m = 0; m < MT; ++m) f.a[m] = *(const uint4*)(lds_a + a_off[m] + tap_off);
                };
                auto mfma_step = [&](const FragsK& f) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                __builtin_bit_cast(h16x8, REGEPI ? f.b[n] : f.a[m]), __builtin_bit_cast(h16x8, REGEPI ? f.a[m] : f.b[n]), acc[m][n], 0, 0, 0);
                };
                FragsK g0, g1;
                read_step(g0, 0);
                for (int i = 0; i < tcur; ++i) {
                    read_step(g1, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_step(g0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (rt + 1 < tcur) { ++rt; if (++kx == p.kw) { kx = 0; ++ky; } }
                    read_step(g0, 0);                  // past the last tap: re-reads it (branch-free loop body)
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_step(g1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (++kx == p.kw) { kx = 0; ++ky; }
                PROBE_STAMP(t_g4)
                PROBE_ADD(4, t_f, t_g4)
                continue;
            }
            Frags fA, fB;
            read_tap(fA);
            for (int i = 0; i < (tcur >> 1); ++i) {
                read_tap(fB);
                __builtin_amdgcn_sched_barrier(0);
                mfma_tap(fA);
                __builtin_amdgcn_sched_barrier(0);
                read_tap(fA);
                __builtin_amdgcn_sched_barrier(0);
                mfma_tap(fB);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tcur & 1) mfma_tap(fA);
            if (++kx == p.kw) { kx = 0; ++ky; }            // step past the group's last tap for the next group
            PROBE_STAMP(t_g)
            PROBE_ADD(4, t_f, t_g)
        }
    }
    PROBE_STAMP(t_epi)
#ifdef DSVC_PP_CLOCK
    if (tid == 0) {
        const unsigned long long ck_t1 = __builtin_amdgcn_s_memtime(), ck_r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&g_mfma_clock[0], ck_t1 - ck_t0); atomicAdd(&g_mfma_clock[1], ck_r1 - ck_r0); atomicAdd(&g_mfma_clock[2], 1ull);
    }
#endif

    // ---- epilogue ----
    // MFMA C/D layout: col = lane&31 (output channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel).
    // Accumulators go through a wave-private LDS slab [pixel][channel] (<= 2 N-tiles per pass) so the fused
    // bias/activation/residual/store code is a short rolled loop moving 16-byte vectors along the NHWC channel axis.
    if constexpr (REGEPI) {
        // D[channel][pixel]: lane (r, h) holds pixel ox0 + r, channels (i & 3) + 8 * (i >> 2) + 4 * h of every 32-channel tile
        const EpiRsrc R = make_epi_rsrc(p);
        const rsrc_t rbias = make_rsrc(p.bias ? (const void*)(p.bias + (size_t)b * p.bias_bs) : (const void*)p.out, p.bias ? (unsigned)p.Cout * 4u : 0u);
        const float sl = p.act == DSVC_ACT_NONE ? 1.f : (p.act == DSVC_ACT_RELU ? 0.f : p.slope);      // act(v) = max(v, v * sl)
        const unsigned r1m = p.res1 ? 0u : OOB;
        const int ox = ox0 + r;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int ct = ng * NTN + n * 32;                       // first channel of this tile (uniform)
            const unsigned cm = ct < p.Cout ? 0u : OOB;
            const unsigned co0 = (unsigned)(ct + 8 * h);            // after the swap: channels co0..co0+7 and co0+16..co0+23
            const uint4 ba = buf_load16(rbias, (co0 * 4u) | cm), bb = buf_load16(rbias, (co0 * 4u + 16u) | cm);
            const uint4 bc = buf_load16(rbias, (co0 * 4u + 64u) | cm), bd = buf_load16(rbias, (co0 * 4u + 80u) | cm);
            const unsigned bw0[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w}, bw1[8] = {bc.x, bc.y, bc.z, bc.w, bd.x, bd.y, bd.z, bd.w};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int oy = oy0 + wave * MT + m;
                const unsigned pm = (oy < p.Ho && ox < p.Wo) ? cm : OOB;
                const unsigned opix = ((unsigned)b * p.Ho + (unsigned)oy) * p.Wo + (unsigned)ox;
                const uint4 ra = buf_load16(R.r1, ((opix * p.res1_ld + co0) * 2u) | pm | r1m);
                const uint4 rb = buf_load16(R.r1, ((opix * p.res1_ld + co0) * 2u + 32u) | pm | r1m);
                float v0[8], v1[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    {   // scalars first: hipcc 7.2 mis-casts vector elements
                        const float xq = acc[m][n][j], yq = acc[m][n][4 + j];
                        const v2u_t sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false);
                        const unsigned lo_ = sw.x, hi_ = sw.y;
                        v0[j] = __builtin_bit_cast(float, lo_); v0[4 + j] = __builtin_bit_cast(float, hi_);
                    }
                    {
                        const float xq = acc[m][n][8 + j], yq = acc[m][n][12 + j];
                        const v2u_t sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false);
                        const unsigned lo_ = sw.x, hi_ = sw.y;
                        v1[j] = __builtin_bit_cast(float, lo_); v1[4 + j] = __builtin_bit_cast(float, hi_);
                    }
                }
                const unsigned ru0[4] = {ra.x, ra.y, ra.z, ra.w}, ru1[4] = {rb.x, rb.y, rb.z, rb.w};
                unsigned o0[4], o1[4];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t0 = v0[j] + __builtin_bit_cast(float, bw0[j]), t1 = v1[j] + __builtin_bit_cast(float, bw1[j]);
                    v0[j] = fmaxf(t0, t0 * sl); v1[j] = fmaxf(t1, t1 * sl);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v0[2 * j] += h16_lo(ru0[j]); v0[2 * j + 1] += h16_hi(ru0[j]);
                    v1[2 * j] += h16_lo(ru1[j]); v1[2 * j + 1] += h16_hi(ru1[j]);
                    o0[j] = pk_h16(v0[2 * j], v0[2 * j + 1]);
                    o1[j] = pk_h16(v1[2 * j], v1[2 * j + 1]);
                }
                const v4u_t s0 = {o0[0], o0[1], o0[2], o0[3]}, s1 = {o1[0], o1[1], o1[2], o1[3]};
                __builtin_amdgcn_raw_buffer_store_b128(s0, R.out, ((opix * p.out_ld + co0) * 2u) | pm, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(s1, R.out, ((opix * p.out_ld + co0) * 2u + 32u) | pm, 0, 0);
            }
        }
        PROBE_FINISH
        return;
    }
    constexpr int EPN = NT >= 2 ? 2 : 1;
    constexpr int EPM = MT >= 2 ? 2 : 1;            // output rows per pass (the slab of a 16-row tile would not leave room for a second workgroup)
    constexpr int EPC = EPN * 32, EPLD = EPC + 4, GPP = EPC / 4;
    float* slab = (float*)smem + (size_t)wave * (EPM * 32 * EPLD);
    // per-item output placement (dsvc.h: out_bstride): this item's output / residual / bias become image 0 of the epilogue's view
    const int be = p.out_bs ? 0 : b;
    const EpiRsrc R = make_epi_rsrc(p, b);
    // The passes (<= 2 channel pairs x <= 2 row pairs) are a ROLLED loop around ONE copy of the fused epilogue; only the accumulator ->
    // slab staging of each pass is expanded with compile-time indices.  (Unrolling the passes around the inlined epilogue dispatch once
    // exceeded LLVM's pragma-unroll size limit: the loops stayed rolled, `acc` was indexed dynamically and went to scratch memory --
    // (MT * NT + 1) * 64 bytes per lane, 3x slower NT >= 3 slab variants; caught by tools/bench_subpel.py + kernel-resource-usage.)
    constexpr int NPN = (NT + EPN - 1) / EPN, NPM = MT / EPM;
    static_assert(MT % EPM == 0 && NPN <= 2 && NPM <= 2, "slab epilogue: at most 2 x 2 passes of whole row pairs");
    auto stage = [&](auto n0c, auto m0c) {
        constexpr int n0 = decltype(n0c)::value, m0 = decltype(m0c)::value;
#pragma unroll
        for (int e = 0; e < EPN; ++e) {
            if constexpr (true) {
                if (n0 + e < NT) {
#pragma unroll
                    for (int m = 0; m < EPM; ++m) {
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            slab[(m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * EPLD + e * 32 + r] = acc[m0 + m][n0 + e < NT ? n0 + e : 0][i];
                    }
                }
            }
        }
    };
    using std::integral_constant;
#pragma unroll 1
    for (int ps = 0; ps < NPN * NPM; ++ps) {
        const int n0 = (ps / NPM) * EPN, m0 = (ps % NPM) * EPM;      // scalars
        lds_barrier();
        if (ps == 0) stage(integral_constant<int, 0>{}, integral_constant<int, 0>{});
        if constexpr (NPN * NPM > 1) { if (ps == 1) stage(integral_constant<int, (1 / NPM) * EPN>{}, integral_constant<int, (1 % NPM) * EPM>{}); }
        if constexpr (NPN * NPM > 2) { if (ps == 2) stage(integral_constant<int, (2 / NPM) * EPN>{}, integral_constant<int, (2 % NPM) * EPM>{}); }
        if constexpr (NPN * NPM > 3) { if (ps == 3) stage(integral_constant<int, (3 / NPM) * EPN>{}, integral_constant<int, (3 % NPM) * EPM>{}); }
        lds_barrier();
        const int cpass = (NT - n0 < EPN ? NT - n0 : EPN) * 32;  // channels staged this pass
        run_epilogue<GPP, EPLD, EPM * 32, sizeof(CT) == 4, true>(p, R, slab, lane, be, oy0 + wave * MT + m0, ox0, ng * NTN + n0 * 32, cpass);
    }
    PROBE_FINISH
}

template <typename CT, int MT, int NT, bool FAST, bool REGEPI = false, int PF = 0>
int launch_cfg(hipStream_t st, const ConvParams& p, int ngroups, size_t lds) {
    static bool attr_set = false;
    auto kern = conv_mfma_kernel<CT, MT, NT, FAST, REGEPI, PF>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { dsvc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return DSVC_E_LAUNCH; }
        attr_set = true;
    }
    dim3 grid((unsigned)(p.B * p.tiles_x * p.tiles_y), (unsigned)ngroups);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    DSVC_CHECK_LAUNCH("conv_mfma_kernel");
    return DSVC_OK;
}

template <typename CT, int MT, bool FAST>
int launch_nt(hipStream_t st, const ConvParams& p, int NT, int ngroups, size_t lds) {
    switch (NT) {
        case 1: return launch_cfg<CT, MT, 1, FAST>(st, p, ngroups, lds);
        case 2: return launch_cfg<CT, MT, 2, FAST>(st, p, ngroups, lds);
        case 3: return launch_cfg<CT, MT, 3, FAST>(st, p, ngroups, lds);
        default: return launch_cfg<CT, MT, 4, FAST>(st, p, ngroups, lds);
    }
}

template <typename CT>
int launch_ct(hipStream_t st, const ConvParams& p, int MT, int NT, bool fast, int ngroups, size_t lds) {
    if constexpr (sizeof(CT) == 2) {
        if (p.regepi) {                       // register epilogue (see conv_mfma_kernel): large f16 convs only
            if (MT == 4) return launch_cfg<CT, 4, 2, true, true>(st, p, ngroups, lds);
            if (NT == 2) return p.patchpf ? launch_cfg<CT, 2, 2, true, true, 6>(st, p, ngroups, lds) : launch_cfg<CT, 2, 2, true, true>(st, p, ngroups, lds);
            if (NT == 3) return launch_cfg<CT, 2, 3, true, true>(st, p, ngroups, lds);
            return launch_cfg<CT, 2, 4, true, true>(st, p, ngroups, lds);
        }
        if (MT == 4) return launch_cfg<CT, 4, 2, true>(st, p, ngroups, lds);      // 16x32-pixel tiles: f16, vector staging, 64 output channels per group
        if (fast && MT == 1 && NT <= 3 && p.patchpf)         // small images (entropy scan, 1/8 - 1/16 resolution): 4x32 tiles + patch prefetch
            return NT == 1 ? launch_cfg<CT, 1, 1, true, false, 4>(st, p, ngroups, lds)
                 : NT == 2 ? launch_cfg<CT, 1, 2, true, false, 4>(st, p, ngroups, lds) : launch_cfg<CT, 1, 3, true, false, 4>(st, p, ngroups, lds);
        if (fast && MT == 2 && NT == 1 && p.patchpf) return launch_cfg<CT, 2, 1, true, false, 6>(st, p, ngroups, lds);
    }
    if constexpr (sizeof(CT) == 4) {
        if (fast && MT == 2 && NT == 2 && p.patchpf) return launch_cfg<CT, 2, 2, true, false, 6>(st, p, ngroups, lds);     // fp32: slab epilogue + patch prefetch
        if (fast && MT == 1 && NT <= 3 && p.patchpf)         // small images: 4x32 tiles + patch prefetch, as in f16 mode
            return NT == 1 ? launch_cfg<CT, 1, 1, true, false, 4>(st, p, ngroups, lds)
                 : NT == 2 ? launch_cfg<CT, 1, 2, true, false, 4>(st, p, ngroups, lds) : launch_cfg<CT, 1, 3, true, false, 4>(st, p, ngroups, lds);
        if (fast && MT == 2 && NT == 1 && p.patchpf) return launch_cfg<CT, 2, 1, true, false, 6>(st, p, ngroups, lds);
    }
    if (fast) return MT == 2 ? launch_nt<CT, 2, true>(st, p, NT, ngroups, lds) : launch_nt<CT, 1, true>(st, p, NT, ngroups, lds);
    return MT == 2 ? launch_nt<CT, 2, false>(st, p, NT, ngroups, lds) : launch_nt<CT, 1, false>(st, p, NT, ngroups, lds);
}

inline int chunk_elems(int compute) { return compute == DSVC_F32 ? 16 : 32; }

}  // namespace


static int conv2d_impl(void* stream, const dsvc_conv_desc* d);

// ---- per-launch timing of one conv SHAPE with HIP events on the launch stream (bench.py's roofline leg) --------------------
// Works the same whether the launch comes from Python, from a recorded scan step or from a frame segment.
#include <mutex>
#include <vector>
namespace {
struct ProfRec { hipEvent_t a, b; int nres; };
struct ProfState { bool active = false; int kh, stride, C, Cout, H, W; std::vector<ProfRec> recs; std::mutex mu; } g_prof;
}
extern "C" int dsvc_profile_begin(int32_t kh, int32_t stride, int32_t C, int32_t Cout, int32_t H, int32_t W) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (auto& r : g_prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.recs.clear();
    g_prof.kh = kh; g_prof.stride = stride; g_prof.C = C; g_prof.Cout = Cout; g_prof.H = H; g_prof.W = W;
    g_prof.active = true;
    return DSVC_OK;
}
extern "C" int64_t dsvc_profile_end(float* ms_out, int32_t* nres_out, int64_t cap) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.active = false;
    const int64_t n = (int64_t)g_prof.recs.size();
    if (ms_out == nullptr) return n;                 // size query (records are kept)
    for (int64_t i = 0; i < n && i < cap; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.recs[i].b) != hipSuccess || hipEventElapsedTime(&ms, g_prof.recs[i].a, g_prof.recs[i].b) != hipSuccess) {
            dsvc_set_error("profile_end: event %lld not usable", (long long)i);
            return DSVC_E_LAUNCH;
        }
        ms_out[i] = ms;
        if (nres_out) nres_out[i] = g_prof.recs[i].nres;
    }
    for (auto& r : g_prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.recs.clear();
    return n;
}

extern "C" int dsvc_conv2d(void* stream, const dsvc_conv_desc* d) {
    if (g_prof.active && d && d->nseg == 1 && d->kh == g_prof.kh && d->kw == g_prof.kh && d->stride == g_prof.stride && d->in[0].C == g_prof.C &&
        d->Cout == g_prof.Cout && d->H == g_prof.H && d->W == g_prof.W && !d->s2d) {
        ProfRec r{};
        if (hipEventCreate(&r.a) == hipSuccess && hipEventCreate(&r.b) == hipSuccess) {
            r.nres = (d->res1 ? 1 : 0) + (d->res2 ? 1 : 0);
            (void)hipEventRecord(r.a, (hipStream_t)stream);
            const int rc = conv2d_impl(stream, d);
            (void)hipEventRecord(r.b, (hipStream_t)stream);
            std::lock_guard<std::mutex> lk(g_prof.mu);
            g_prof.recs.push_back(r);
            return rc;
        }
    }
    return conv2d_impl(stream, d);
}

static int conv2d_impl(void* stream, const dsvc_conv_desc* d) {
    DSVC_REQUIRE(d != nullptr, "conv2d: null descriptor");
    DSVC_REQUIRE(d->nseg >= 1 && d->nseg <= 3, "conv2d: nseg=%d out of 1..3", d->nseg);
    DSVC_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "conv2d: bad shape B=%d H=%d W=%d Cout=%d", d->B, d->H, d->W, d->Cout);
    DSVC_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->kh <= 7 && d->kw <= 7, "conv2d: kernel %dx%d unsupported", d->kh, d->kw);
    DSVC_REQUIRE(d->stride == 1 || d->stride == 2, "conv2d: stride %d unsupported", d->stride);
    DSVC_REQUIRE(d->compute == DSVC_F32 || d->compute == DSVC_F16, "conv2d: bad compute type %d", d->compute);
    DSVC_REQUIRE(d->weight && d->out, "conv2d: null weight/out");
    DSVC_REQUIRE(d->shuffle == 1 || (d->shuffle == 2 && d->Cout % 4 == 0), "conv2d: shuffle=%d Cout=%d", d->shuffle, d->Cout);
    DSVC_REQUIRE(!(d->shuffle == 2 && (d->res2 || d->res1_scale)), "conv2d: shuffle with res2/gate unsupported");
    DSVC_REQUIRE(((uintptr_t)d->weight & 15) == 0, "conv2d: packed weights must be 16-byte aligned");

    ConvParams p{};
    const int kce = chunk_elems(d->compute);
    const bool per_item = d->batch_inner > 0;    // batch items with their own weights and input offsets (dsvc.h: batch_inner)
    if (per_item)
        DSVC_REQUIRE(!d->s2d && !d->post_weight && !d->res1_scale && d->weight_bstride >= 0 && d->weight_bstride % 16 == 0 && d->bias_bstride >= 0,
                     "conv2d: batch_inner excludes s2d / post_weight / res1_scale and needs 16-byte weight strides");
    int chunk0 = 0;
    for (int s = 0; s < d->nseg; ++s) {
        const dsvc_seg& in = d->in[s];
        DSVC_REQUIRE(in.ptr && in.C > 0 && in.ld >= (d->s2d ? in.C / 4 : in.C), "conv2d: segment %d ptr/C/ld invalid (C=%d ld=%d)", s, in.C, in.ld);
        DSVC_REQUIRE(in.dtype == DSVC_F32 || in.dtype == DSVC_F16, "conv2d: segment %d dtype", s);
        const int es = dsvc_esize(in.dtype);
        if (d->s2d)
            DSVC_REQUIRE(in.dtype == d->compute && in.C % kce == 0 && ((uintptr_t)in.ptr & 15) == 0 && ((int64_t)in.ld * es) % 16 == 0,
                         "conv2d: s2d segment %d needs storage == compute type, C %% %d == 0 and 16-byte aligned rows", s, kce);
        // extent of the underlying image: the view grid itself, or the (2H x 2W) image behind a space-to-depth view
        const int64_t ext = d->s2d ? (((int64_t)d->B * 4 * d->H * d->W - 1) * in.ld + in.C / 4) * es
                                   : (((int64_t)d->B * d->H * d->W - 1) * in.ld + in.C) * es;
        DSVC_REQUIRE(ext < 0x7FFFFFF0ll, "conv2d: input segment %d spans %lld bytes (limit 2 GiB)", s, (long long)ext);
        p.seg[s] = ConvSeg{in.ptr, in.C, in.ld, in.dtype,
                           (int)((((uintptr_t)in.ptr & 15) == 0) && (((int64_t)in.ld * es) % 16 == 0)), chunk0, (unsigned)ext,
                           (unsigned)((int64_t)(d->s2d ? 4 : 1) * d->H * d->W * in.ld * es), 0u};
        if (per_item) {                      // explicit batch-item offsets instead of the dense [B,H,W,ld] layout
            const int64_t lo = d->in_bstride[s][0], hi = d->in_bstride[s][1];
            DSVC_REQUIRE(lo >= 0 && hi >= 0 && lo % 16 == 0 && hi % 16 == 0, "conv2d: in_bstride of segment %d must be non-negative multiples of 16", s);
            const int64_t img = (((int64_t)d->H * d->W - 1) * in.ld + in.C) * es;
            const int nlo = d->B < d->batch_inner ? d->B : d->batch_inner, nhi = (d->B + d->batch_inner - 1) / d->batch_inner;
            const int64_t e2 = (nlo - 1) * lo + (nhi - 1) * hi + img;
            DSVC_REQUIRE(e2 < 0x7FFFFFF0ll, "conv2d: input segment %d spans %lld bytes (limit 2 GiB)", s, (long long)e2);
            p.seg[s].bytes = (unsigned)e2; p.seg[s].bs_lo = (unsigned)lo; p.seg[s].bs_hi = (unsigned)hi;
        }
        chunk0 += (in.C + kce - 1) / kce;
    }
    p.b_inner = per_item ? d->batch_inner : 0x7FFFFFFF;
    p.w_bs = per_item ? d->weight_bstride : 0; p.bias_bs = per_item ? (int)d->bias_bstride : 0;
    const bool item_out = per_item && d->out_bstride > 0;
    DSVC_REQUIRE(per_item || (d->out_bstride == 0 && d->res1_bstride == 0), "conv2d: out_bstride / res1_bstride need batch_inner > 0");
    DSVC_REQUIRE(!item_out || (d->shuffle == 1 && !d->res2 && d->res1_bstride >= 0 && (d->res1 == nullptr || d->res1_bstride > 0)),
                 "conv2d: out_bstride excludes shuffle / res2 and needs res1_bstride with res1");
    p.out_bs = item_out ? d->out_bstride : 0; p.res1_bs = item_out ? d->res1_bstride : 0;
    p.nseg = d->nseg; p.nchunks = chunk0;
    p.B = d->B; p.H = d->H; p.W = d->W;
    p.s2d = d->s2d ? 1 : 0;
    DSVC_REQUIRE(!d->s2d || (d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1), "conv2d: s2d needs a 3x3 stride-1 pad-1 kernel");
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
    p.Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    p.Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    DSVC_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv2d: empty output");
    p.Cout = d->Cout; p.Np = (d->Cout + 31) / 32 * 32;
    p.w = d->weight; p.bias = d->bias;
    {
        const int64_t wb = (int64_t)chunk0 * d->kh * d->kw * p.Np * 64;       // packed weights of one conv (dsvc_conv_packed_bytes)
        DSVC_REQUIRE(wb < 0x7FFFFFF0ll, "conv2d: packed weights of %lld bytes exceed 2 GiB", (long long)wb);
        p.w_bytes = (unsigned)wb;
    }
    p.pre_act = d->pre_act; p.act = d->act; p.slope = d->slope;
    p.res1 = d->res1; p.res1_ld = d->res1_ld; p.res1_dtype = d->res1_dtype; p.res1_scale = d->res1_scale;
    p.res2 = d->res2; p.res2_ld = d->res2_ld; p.res2_dtype = d->res2_dtype;
    p.post_w = d->post_weight; p.post_bias = d->post_bias;
    p.out = d->out; p.out_ld = d->out_ld; p.out_dtype = d->out_dtype; p.shuffle = d->shuffle;
    // 4-channel vector access (16 B fp32 / 8 B f16) is legal when base and pixel pitch keep that alignment
    auto vec4_ok = [](const void* ptr, int ld, int dtype) {
        const int bytes = 4 * dsvc_esize(dtype);
        return ptr != nullptr && ((uintptr_t)ptr % bytes) == 0 && (ld % 4) == 0;
    };
    p.res1_vec = vec4_ok(d->res1, d->res1_ld, d->res1_dtype) && (!item_out || d->res1_bstride % (4 * dsvc_esize(d->res1_dtype)) == 0);
    p.res2_vec = vec4_ok(d->res2, d->res2_ld, d->res2_dtype);
    p.out_vec = vec4_ok(d->out, d->out_ld, d->out_dtype) && (!item_out || d->out_bstride % (4 * dsvc_esize(d->out_dtype)) == 0);
    {
        const int sh = d->shuffle;
        const int64_t opix = (int64_t)(item_out ? 1 : d->B) * p.Ho * sh * p.Wo * sh, oc = d->Cout / (sh * sh);      // per-item placement: one image per item
        const int64_t oe = ((opix - 1) * d->out_ld + oc) * dsvc_esize(d->out_dtype);
        const int64_t r1e = d->res1 ? ((opix - 1) * d->res1_ld + oc) * dsvc_esize(d->res1_dtype) : 0;
        const int64_t r2e = d->res2 ? ((opix - 1) * d->res2_ld + oc) * dsvc_esize(d->res2_dtype) : 0;
        DSVC_REQUIRE(oe < 0x7FFFFFF0ll && r1e < 0x7FFFFFF0ll && r2e < 0x7FFFFFF0ll, "conv2d: output/residual tensor above 2 GiB");
        p.out_bytes = (unsigned)oe; p.res1_bytes = (unsigned)r1e; p.res2_bytes = (unsigned)r2e;
        p.epi_vec = d->shuffle == 1 && (d->Cout % 4) == 0 && p.out_vec && (!d->res1 || p.res1_vec) && (!d->res2 || p.res2_vec);
        // PixelShuffle(2): vector stores of 4 consecutive OUTPUT channels (conv_common.h: epilogue_vec_shuffle)
        if (d->shuffle == 2 && (d->Cout % 16) == 0 && p.out_vec && (!d->res1 || p.res1_vec) && !item_out && g_dsvc_opt.shuffle_vec &&
            (d->act == DSVC_ACT_NONE || d->act == DSVC_ACT_RELU || d->act == DSVC_ACT_LRELU))
            p.epi_vec = 1;
        p.epi_f32 = g_dsvc_opt.epi_f32 && d->out_dtype == DSVC_F32 && (!d->res1 || d->res1_dtype == DSVC_F32) && (!d->res2 || d->res2_dtype == DSVC_F32);
    }
    bool fast = true;
    for (int s = 0; s < d->nseg; ++s)
        fast = fast && p.seg[s].vec_ok &&
               ((p.seg[s].dtype == d->compute && (p.seg[s].C % (kce / 4)) == 0) ||
                (d->compute == DSVC_F16 && p.seg[s].dtype == DSVC_F32 && (p.seg[s].C % 4) == 0 && !d->s2d));   // converted while staged

    // ---- f16 3x3 C_in<=64 convs at large resolution: persistent kernel with LDS-resident weights ----
    DsvcOptions opt = g_dsvc_opt;
    if (p.s2d || per_item) {                // only the general kernel's staging understands the space-to-depth view / item offsets
        DSVC_REQUIRE(fast, "conv2d: s2d and batch_inner need the vector staging path (16-byte aligned rows, whole 4/8-channel quarters)");
        opt.specialised = 0;
    }
    if (p.post_w) {                          // fused trailing 1x1 conv: only the two-group 3x3 kernel implements it
        const int rc = dsvc_conv3x3_pp_try((hipStream_t)stream, p, d);
        DSVC_REQUIRE(rc != 1, "conv2d: post_weight (fused 1x1) is not supported for this shape / type combination");
        if (rc == DSVC_OK) { dsvc_count(DSVC_CNT_PP); dsvc_count(DSVC_CNT_PP_POST); }
        return rc;
    }
    if (opt.specialised && opt.headtap) {   // 3x3 f16 heads with Cout <= 3 of large images: per-input-pixel (tap, output) GEMM on the matrix pipe + 9-point gather-sum
        const int rc = dsvc_conv_headtap_try((hipStream_t)stream, p, d);
        if (rc == DSVC_OK) dsvc_count(DSVC_CNT_HEADTAP);
        if (rc != 1) return rc;
    }
    if (opt.specialised && opt.cin8) {      // f16 3x3 / 7x7 convs over one 8-channel input: the x-taps of a kernel row packed into the MFMA's K axis
        const int rc = dsvc_conv_cin8_try((hipStream_t)stream, p, d);
        if (rc == DSVC_OK) dsvc_count(DSVC_CNT_CIN8);
        if (rc != 1) return rc;
    }
    if (opt.specialised && opt.rr7) {       // f16 7x7 convs 32 -> 16 / 32 / 64 and 64 -> 32: four output rows per wave, pixel fragments reused across kernel rows in registers
        const int rc = dsvc_conv7x7_rr_try((hipStream_t)stream, p, d);
        if (rc == DSVC_OK) dsvc_count(DSVC_CNT_RR7);
        if (rc != 1) return rc;
    }
    if (opt.specialised && opt.head) {      // narrow f16 output heads (Cout <= 4) of large images: dot products on the vector ALU
        const int rc = dsvc_conv_head_try((hipStream_t)stream, p, d);
        if (rc == DSVC_OK) dsvc_count(DSVC_CNT_HEAD);
        if (rc != 1) return rc;
    }
    if (opt.specialised && opt.head_f32) {  // ... and of fp32 images: the MFMA's own fma chain on the vector ALU, bit-identical
        const int rc = dsvc_conv_head_f32_try((hipStream_t)stream, p, d);
        if (rc == DSVC_OK) dsvc_count(DSVC_CNT_HEAD);
        if (rc != 1) return rc;
    }
    if (opt.specialised && opt.conv1x1) {   // f16 pointwise convs: streaming kernel, no LDS staging
        const int rc = dsvc_conv1x1_try((hipStream_t)stream, p, d);
        if (rc == DSVC_OK) dsvc_count(DSVC_CNT_CONV1X1);
        if (rc != 1) return rc;
    }
    if (opt.specialised && opt.conv1x1_f32) {   // fp32 pointwise convs over 64 / 96 / 128 channels of whole images: streaming kernel, bit-identical to the general one
        const int rc = dsvc_conv1x1_f32_try((hipStream_t)stream, p, d);
        if (rc == DSVC_OK) dsvc_count(DSVC_CNT_CONV1X1);
        if (rc != 1) return rc;
    }
    if (opt.specialised && opt.pp) {        // 64-channel f16 ResBlock family: two phase-skewed wave groups per CU
        const int rc = dsvc_conv3x3_pp_try((hipStream_t)stream, p, d);
        if (rc == DSVC_OK) dsvc_count(DSVC_CNT_PP);
        if (rc != 1) return rc;
    }
    if (opt.specialised && opt.pps) {       // f16 3x3 convs with 3+ input chunks and 64 output channels: two wave groups, weights streamed per chunk
        const int rc = dsvc_conv3x3_pps_try((hipStream_t)stream, p, d);
        if (rc == DSVC_OK) dsvc_count(DSVC_CNT_PPS);
        if (rc != 1) return rc;
    }
    if (opt.specialised && opt.ws) {        // other large f16 3x3 convs with <= 64 input channels: LDS-resident weights
        const int rc = dsvc_conv3x3_ws_try((hipStream_t)stream, p, d);
        if (rc == DSVC_OK) dsvc_count(DSVC_CNT_WS);
        if (rc != 1) return rc;
    }
    // ---- tile configuration ----
    const int ntiles32 = p.Np / 32;
    int NT = ntiles32 >= 4 ? 4 : ntiles32;           // 1,2,3,4 (96 ch -> 3)
    int MT = (d->stride == 2) ? 1 : 2;
    auto nblocks = [&](int mt, int nt) {
        return (int64_t)p.B * ((p.Ho + 4 * mt - 1) / (4 * mt)) * ((p.Wo + TW - 1) / TW) * ((ntiles32 + nt - 1) / nt);
    };
    // small problems (entropy-parameter nets at 1/16 res): trade operand reuse for more workgroups
    // 400, not 512: at 1/4 resolution (272x480) the 8x32 tiling gives 510 workgroups -- two per CU, which is fine -- and
    // halves the weight re-staging per pixel (128->128 3x3: 90 -> 70 us); below ~1.5 workgroups per CU go finer
    static const int mt_thr = getenv("DSVC_MT_THR") ? atoi(getenv("DSVC_MT_THR")) : 400;
    if (nblocks(MT, NT) < mt_thr && MT == 2) MT = 1;
    while (nblocks(MT, NT) < mt_thr && NT > 1) NT = (NT + 1) / 2;
    // large f16 stride-1 convs with 64 output channels per group: 16x32-pixel tiles -- 0.75 instead of 1.0 ds_read_b128 per
    // MFMA and half the weight staging per pixel.  Worth only ~4 % (3x3 64+64->64 @1088x1920: 491 -> 471 us): the PMC
    // counters put the LDS at ~35 % busy and the MFMA pipe at 36 %, i.e. the kernel waits on barriers and staging latency
    // with two workgroups per CU, not on LDS bandwidth
    // (96 output channels: 16 rows spill -- 192 accumulator registers --, and a 12-row variant measured 237 vs 152 us)
    bool same_type = true;
    for (int s = 0; s < d->nseg; ++s) same_type = same_type && p.seg[s].dtype == d->compute;
    // 3x3 convs over f16 storage take the patch-prefetch variant of the 8x32 tiling instead (64+64->64: 452 vs 466 us)
    const bool pf_shape = opt.patch_prefetch && opt.reg_epilogue && same_type && d->kh == 3 && d->kw == 3 && d->out_dtype == DSVC_F16;
    if (MT == 2 && NT == 2 && fast && d->compute == DSVC_F16 && d->stride == 1 && !pf_shape && nblocks(4, NT) >= opt.tall_min_tiles) {
        const size_t a4 = (size_t)(15 + d->kh) * (31 + d->kw) * ROWB;
        if (a4 + (size_t)d->kw * NT * 32 * ROWB <= 78 * 1024) MT = 4;           // still two workgroups per CU
    }
    const int TH = 4 * MT;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    const int ngroups = (ntiles32 + NT - 1) / NT;
    const int PH = (TH - 1) * d->stride + d->kh, PW = (TW - 1) * d->stride + d->kw;
    const int PWL = d->stride == 2 ? 2 * ((PW + 1) / 2) : PW;      // LDS row pitch in pixels (stride 2: even | odd columns)
    const size_t lds_a = (size_t)PH * PWL * ROWB;
    const int ntaps = d->kh * d->kw;
    const size_t budget = 78 * 1024;  // two workgroups per CU (one per CU: 96->96 154 -> 213 us, measured)
    p.tapg = ntaps;
    if (lds_a + (size_t)p.tapg * NT * 32 * ROWB > budget) p.tapg = d->kw;
    if (lds_a + (size_t)p.tapg * NT * 32 * ROWB > 160 * 1024) p.tapg = 1;
    {   // the weight prefetch registers of the kernel variant (WP in conv_mfma_kernel) hold one whole staging step
        const int wp = (MT == 4 || (d->compute == DSVC_F32 && NT == 4)) ? 0 : (NT == 1 ? 5 : NT == 2 ? 9 : NT == 3 ? 5 : 6);
        if (wp > 0 && (p.tapg * NT + 1) / 2 > wp) p.tapg = 2 * wp / NT;
    }
    const size_t lds_main = lds_a + (size_t)p.tapg * NT * 32 * ROWB;

    {
        auto al16 = [](const void* q, int ld) { return ((uintptr_t)q & 15) == 0 && (ld % 8) == 0; };
        p.regepi = opt.reg_epilogue && !item_out && fast && d->compute == DSVC_F16 && (MT == 4 || (MT == 2 && NT >= 2)) && d->out_dtype == DSVC_F16 &&
                   d->shuffle == 1 && (d->Cout % 32) == 0 && al16(d->out, d->out_ld) && !d->res2 && !d->res1_scale &&
                   (!d->res1 || (d->res1_dtype == DSVC_F16 && al16(d->res1, d->res1_ld))) &&
                   (d->act == DSVC_ACT_NONE || d->act == DSVC_ACT_RELU || (d->act == DSVC_ACT_LRELU && d->slope >= 0.f && d->slope <= 1.f));
    }
    {   // patch prefetch variant: f16 storage everywhere, one staging batch (<= 6 vectors per thread) per chunk
        // (96 output channels: 256 registers with spills, 142 vs 129 us; with the k-step fragment pipeline to free registers 137 vs
        //  121 us for the tap-level pipeline without patch prefetch -- not taken)
        const bool f32_pf = d->compute == DSVC_F32 && fast && d->kh == 3 && d->kw == 3 && d->stride == 1 && !p.s2d;
        p.skipk = opt.skip_zero_k != 0;
        p.patchpf = (p.regepi || f32_pf) && opt.patch_prefetch && MT == 2 && NT == 2 && same_type && PH * PW <= 6 * 64;
        // small f16 launches (a few dozen to a few hundred workgroups, each a serial chain of stage -> MFMA per 32-channel chunk: the
        // entropy scan and the 1/8 - 1/16 resolution layers, ~250 launches per 1080p frame): the next chunk's patch loads ride under
        // the current chunk's weight commit and MFMAs (tools/phase_probe_tiny.py: patch staging was 25-30 % of a workgroup's clocks)
        // (fp32 mode: the same variants over 16-channel chunks -- same bytes per pixel row, same register budget; staging only, bit-identical)
        if (opt.patch_prefetch >= 1 && opt.small_pf && (d->compute == DSVC_F16 || opt.small_pf >= 1) && fast && same_type && p.nchunks >= 2 && !p.regepi && !p.patchpf &&
            ((MT == 1 && NT <= 3 && PH * PW <= 4 * 64) || (MT == 2 && NT == 1 && PH * PW <= 6 * 64)))
            p.patchpf = 1;
    }
    const size_t lds_epi = p.regepi ? 0 : (size_t)4 * (MT >= 2 ? 2 : 1) * 32 * ((NT >= 2 ? 64 : 32) + 4) * sizeof(float);
    const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    DSVC_REQUIRE(lds <= 160 * 1024, "conv2d: LDS need %zu exceeds 160 KiB", lds);

    hipStream_t st = (hipStream_t)stream;
    dsvc_count(DSVC_CNT_GENERIC);
    if (d->compute == DSVC_F32) return launch_ct<float>(st, p, MT, NT, fast, ngroups, lds);
    return launch_ct<h16>(st, p, MT, NT, fast, ngroups, lds);
}

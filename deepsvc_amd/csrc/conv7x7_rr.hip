// conv7x7_rr.hip -- f16 7x7 stride-1 convs over whole 32-channel chunks with ROW REUSE in registers: ME_Spynet's inner convs 32 -> 64, 64 -> 32
// and 32 -> 16 on every pyramid level (modules.py:127-131).
//
// In the general kernel a 7x7 conv with one or two 32-column output tiles is bound by LDS reads inside its MFMA phase: a pixel fragment is
// used once per tap (1.5 ds_read_b128 per MFMA; 566 us for 64 -> 32 at 1088x1920 = 0.30 of the matrix peak, 300 us for 32 -> 16).  A 7x7 kernel
// offers a reuse the 3x3 one hardly does: patch row p serves output rows p - 6 .. p, so a wave that owns FOUR consecutive output rows of a
// 32-pixel-wide tile and walks the ten patch rows of one (chunk, kx, k-step) column reads each pixel fragment ONCE for up to four MFMAs,
// with the seven weight fragments (ky = 0 .. 6) of that column held in registers: 10 pixel reads + 7 weight loads per 28 MFMAs = 0.61 operand
// reads per MFMA instead of 1.5.
//   * weights never touch LDS: lane (r, h) of the fragment (chunk, ky, kx, k-step, N-tile) is one 16-byte load from the MFMA-packed buffer
//     ([chunk][tap][Np rows] x 32 halfs; the four waves of a workgroup load the same lines: L1 / L2 hits), issued one half-iteration ahead
//     into a ping-pong pair of register sets (k-step 0 / 1 of the next kx column);
//   * the patch holds ONE 32-channel chunk of (16 + 6) x (32 + 6) pixels at an 80-byte pitch (conflict-free ds_read_b128): 67 KB, two
//     workgroups per CU -- one stages / stores while the other multiplies; a 64-channel input is two stages accumulated in the same registers;
//   * operand roles are swapped (D[channel][pixel]) and the epilogue is conv_cin8.hip's: v_permlane32_swap -> 8 consecutive channels per lane,
//     bias preloaded, activation as one v_med3, whole output rows through a wave-private LDS tile; 64 output channels = two passes over the patch.
// Summation order differs from the general kernel's (kx outermost; fp32 sums either way): f16-mode tolerance.
#include <climits>
#include "conv_common.h"

using namespace dsvc_conv;

namespace {

constexpr int RR_TH = 16, RR_TW = 32, RR_PH = RR_TH + 6, RR_PW = RR_TW + 6, RR_PITCH = 80, RR_NV = RR_PH * RR_PW * 4;
constexpr int RR_NLD = (RR_NV + 255) / 256;                 // 14 patch vectors per thread
constexpr int RR_TP = 80;                                   // output tile pitch (one 32-channel pass of a pixel row + 16 bytes)

// NCH: 32-channel input chunks (1 or 2).  COUT8: output channels / 8 (2, 4 or 8).
template <int NCH, int COUT8>
__global__ void __launch_bounds__(256, 2) conv7x7_rr_kernel(const ConvParams p, float act_slope) {
    DSVC_F16_SAT();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = (COUT8 + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    char* patch = smem;                                        // [22][38] pixels x 80 B
    char* otile = smem + RR_PH * RR_PW * RR_PITCH + wave * (32 * RR_TP);
    auto tile_sync = [&]() { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); };
    int bid = blockIdx.x;
    const int tx0 = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty0 = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int oy0 = ty0 * RR_TH, ox0 = tx0 * RR_TW;
    const rsrc_t rin = make_rsrc(p.seg[0].ptr, p.seg[0].bytes);
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes);
    const rsrc_t rout = make_rsrc(p.out, p.out_bytes);
    const unsigned ipitch = (unsigned)p.seg[0].ld * 2u, opitch = (unsigned)p.out_ld * 2u;
    const char* arow = patch + ((4 * wave) * RR_PW + r) * RR_PITCH + h * 16;
    // weight fragment (chunk kc, column kx, k-step ks, N-tile n): rows ky = 0 .. 6
    auto load_b = [&](int kc, int kx, int ks, int n, bool live, uint4 (&bw)[7]) {
        const unsigned base = (unsigned)(((kc * 49 + kx) * p.Np + n * 32 + r) * 64 + ks * 32 + h * 16);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) bw[ky] = buf_load16(rw, live ? base + (unsigned)(ky * 7 * p.Np * 64) : OOB);
    };

#pragma unroll 1
    for (int n = 0; n < NT; ++n) {
        f32x16 acc[4];
        {
            float bq[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {                      // accumulator row (reg i) = channel (i & 3) + 8 (i >> 2) + 4 h
                const int co = n * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                bq[i] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
            }
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[y][i] = bq[i];
        }
#pragma unroll 1
        for (int kc = 0; kc < NCH; ++kc) {
            uint4 ba[7], bb[7];
            load_b(kc, 0, 0, n, true, ba);
            load_b(kc, 0, 1, n, true, bb);
            if (NCH > 1 || n == 0) {
                if (kc > 0 || n > 0) lds_barrier();             // every wave has read the previous chunk's patch
                // ---- stage this chunk's patch: vector v -> patch pixel v / 4, 16-byte piece v % 4 (padding reads zeros) ----
                uint4 v[RR_NLD];
#pragma unroll
                for (int i = 0; i < RR_NLD; ++i) {
                    const int vi = tid + 256 * i;
                    const int pix = vi >> 2, piece = vi & 3;
                    const int py = pix / RR_PW, px = pix - py * RR_PW;
                    const int iy = oy0 - 3 + py, ix = ox0 - 3 + px;
                    const bool inb = vi < RR_NV && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    v[i] = buf_load16(rin, (((unsigned)((b * p.H + iy) * p.W + ix) * ipitch + (unsigned)(kc * 64 + piece * 16)) & 0x7FFFFFFFu) | (inb ? 0u : OOB));
                }
#pragma unroll
                for (int i = 0; i < RR_NLD; ++i) {
                    const int vi = tid + 256 * i;
                    if (vi < RR_NV) *(uint4*)(patch + (vi >> 2) * RR_PITCH + (vi & 3) * 16) = v[i];
                }
                lds_barrier();
            }
#pragma unroll 1
            for (int kx = 0; kx < 7; ++kx) {
                // k-step 0 with ba, then refill ba for the next column; k-step 1 with bb, then refill bb
#define RR_HALF(BW, KS)                                                                                                   \
    _Pragma("unroll") for (int pr = 0; pr < 10; ++pr) {                                                                   \
        const uint4 a = *(const uint4*)(arow + (pr * RR_PW + kx) * RR_PITCH + (KS) * 32);                                 \
        _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                                     \
            if (pr - y >= 0 && pr - y < 7)                                                                                \
                acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, BW[pr - y]), __builtin_bit_cast(h16x8, a), acc[y], 0, 0, 0); \
    }
                // the refills are pinned where they are written: left to itself the scheduler sinks all fourteen loads to the end of the body, and
                // the next iteration's first MFMA then waits for a whole L2 round trip (14 round trips per chunk)
                RR_HALF(ba, 0)
                __builtin_amdgcn_sched_barrier(0);
                load_b(kc, kx + 1, 0, n, kx < 6, ba);
                __builtin_amdgcn_sched_barrier(0);
                RR_HALF(bb, 1)
                __builtin_amdgcn_sched_barrier(0);
                load_b(kc, kx + 1, 1, n, kx < 6, bb);
                __builtin_amdgcn_sched_barrier(0);
#undef RR_HALF
            }
        }
        // ---- epilogue of this 32-channel pass: lane (r, h) -> 8 consecutive channels of pixel r per half P; whole rows through the wave's tile ----
        constexpr int ppn = COUT8 < 4 ? COUT8 : 4;             // 16-byte pieces of a pixel row one pass holds (16, 32 or 2 x 32 output channels)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xq = acc[y][8 * P + j], yq = acc[y][8 * P + 4 + j];   // scalars first: hipcc 7.2 mis-casts vector elements
                    const v2u_t sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false);
                    const unsigned lo_ = sw.x, hi_ = sw.y;
                    o[j] = __builtin_bit_cast(float, lo_); o[4 + j] = __builtin_bit_cast(float, hi_);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = __builtin_amdgcn_fmed3f(o[j], o[j] * act_slope, __builtin_inff());
                v4u_t ov;
                ov.x = pk_h16(o[0], o[1]); ov.y = pk_h16(o[2], o[3]); ov.z = pk_h16(o[4], o[5]); ov.w = pk_h16(o[6], o[7]);
                *(v4u_t*)(otile + r * RR_TP + (2 * P + h) * 16) = ov;
            }
            tile_sync();
            const int oy = oy0 + 4 * wave + y;
            const unsigned rowpix = (unsigned)((b * p.Ho + oy) * p.Wo + ox0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {                      // 32 pixels x up to 4 pieces = 2 instructions of 64 lanes
                const int vv = i * 64 + lane, px = vv / ppn, pc = vv - px * ppn;
                const uint4 ou = *(const uint4*)(otile + (px & 31) * RR_TP + pc * 16);
                v4u_t ov; ov.x = ou.x; ov.y = ou.y; ov.z = ou.z; ov.w = ou.w;
                const bool ok = px < 32 && oy < p.Ho && ox0 + px < p.Wo;
                __builtin_amdgcn_raw_buffer_store_b128(ov, rout, ok ? (rowpix + (unsigned)px) * opitch + (unsigned)(n * 4 + pc) * 16u : OOB, 0, 0);
            }
            tile_sync();
        }
    }
}

// ---- 32 -> 16: the same row-reuse walk on v_mfma_f32_16x16x32_f16 ----
// A 32-row weight tile is half padding when the conv has 16 outputs.  On the 16x16x32 shape (A = 16 weight rows x 32 channels, B = 32 channels x 16
// pixels, 16 cycles) ONE MFMA consumes a whole 32-channel tap for 16 pixels: 392 MFMAs of 16 cycles per wave and tile instead of 392 of 32.  Lane
// (c, q) = (lane % 16, lane / 16) reads the 16 bytes of channels 8q .. 8q + 7 of its pixel / of weight row c, and ends up with channels 4q .. 4q + 3
// of pixel c: one 8-byte store, 16 pixels x 32 B = 512 contiguous bytes per instruction -- no LDS in the epilogue.
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) conv7x7_rr16_kernel(const ConvParams p, float act_slope) {
    DSVC_F16_SAT();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    char* patch = smem;
    int bid = blockIdx.x;
    const int tx0 = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty0 = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int oy0 = ty0 * RR_TH, ox0 = tx0 * RR_TW;
    const rsrc_t rin = make_rsrc(p.seg[0].ptr, p.seg[0].bytes);
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes);
    const rsrc_t rout = make_rsrc(p.out, p.out_bytes);
    const unsigned ipitch = (unsigned)p.seg[0].ld * 2u, opitch = (unsigned)p.out_ld * 2u;
    auto load_w = [&](int kx, uint4 (&w)[7]) {                // weight row (tap (ky, kx), output c), channels 8q .. 8q + 7
        const unsigned base = (unsigned)((kx * p.Np + c) * 64 + q * 16);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) w[ky] = buf_load16(rw, base + (unsigned)(ky * 7 * p.Np * 64));
    };
    uint4 wa[7], wb[7];
    load_w(0, wa);
    {   // ---- stage the patch (as in conv7x7_rr_kernel) ----
        uint4 v[RR_NLD];
#pragma unroll
        for (int i = 0; i < RR_NLD; ++i) {
            const int vi = tid + 256 * i;
            const int pix = vi >> 2, piece = vi & 3;
            const int py = pix / RR_PW, px = pix - py * RR_PW;
            const int iy = oy0 - 3 + py, ix = ox0 - 3 + px;
            const bool inb = vi < RR_NV && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            v[i] = buf_load16(rin, (((unsigned)((b * p.H + iy) * p.W + ix) * ipitch + (unsigned)(piece * 16)) & 0x7FFFFFFFu) | (inb ? 0u : OOB));
        }
#pragma unroll
        for (int i = 0; i < RR_NLD; ++i) {
            const int vi = tid + 256 * i;
            if (vi < RR_NV) *(uint4*)(patch + (vi >> 2) * RR_PITCH + (vi & 3) * 16) = v[i];
        }
        lds_barrier();
    }
    f32x4_t acc[4][2];
    {
        float bq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[j] = (p.bias && 4 * q + j < p.Cout) ? p.bias[4 * q + j] : 0.f;      // accumulator row j = channel 4q + j
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[y][hf][j] = bq[j];
    }
    const char* arow = patch + ((4 * wave) * RR_PW + c) * RR_PITCH + q * 16;
#define RR16_COL(W, KX)                                                                                                      \
    _Pragma("unroll") for (int pr = 0; pr < 10; ++pr)                                                                        \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                                   \
            const uint4 a = *(const uint4*)(arow + (pr * RR_PW + (KX) + 16 * hf) * RR_PITCH);                                \
            _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                                    \
                if (pr - y >= 0 && pr - y < 7)                                                                               \
                    acc[y][hf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, W[pr - y]), __builtin_bit_cast(h16x8, a), acc[y][hf], 0, 0, 0); \
        }
#pragma unroll 1
    for (int kx = 0; kx < 6; kx += 2) {                        // columns kx (wa) and kx + 1 (wb); each refill is issued one column ahead
        load_w(kx + 1, wb);
        __builtin_amdgcn_sched_barrier(0);
        RR16_COL(wa, kx)
        __builtin_amdgcn_sched_barrier(0);
        load_w(kx + 2, wa);
        __builtin_amdgcn_sched_barrier(0);
        RR16_COL(wb, kx + 1)
        __builtin_amdgcn_sched_barrier(0);
    }
    RR16_COL(wa, 6)
#undef RR16_COL
    // ---- epilogue: lane (c, q) holds channels 4q .. 4q + 3 of pixel c of each 16-pixel half: activation, f16 pack, one 8-byte store ----
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        const int oy = oy0 + 4 * wave + y;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int ox = ox0 + 16 * hf + c;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = __builtin_amdgcn_fmed3f(acc[y][hf][j], acc[y][hf][j] * act_slope, __builtin_inff());
            v2u_t ov;
            ov.x = pk_h16(o[0], o[1]); ov.y = pk_h16(o[2], o[3]);
            const bool ok = oy < p.Ho && ox < p.Wo;
            __builtin_amdgcn_raw_buffer_store_b64(ov, rout, (((unsigned)((b * p.Ho + oy) * p.Wo + ox) * opitch + (unsigned)q * 8u) & 0x7FFFFFFFu) | (ok ? 0u : OOB), 0, 0);
        }
    }
}

int launch_rr16(hipStream_t st, const ConvParams& p, int ntiles, float act_slope) {
    constexpr size_t lds = (size_t)RR_PH * RR_PW * RR_PITCH;       // 66.9 KB: two workgroups per CU
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv7x7_rr16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { dsvc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return DSVC_E_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL(conv7x7_rr16_kernel, dim3(ntiles), dim3(256), lds, st, p, act_slope);
    DSVC_CHECK_LAUNCH("conv7x7_rr16_kernel");
    return DSVC_OK;
}

template <int NCH, int COUT8>
int launch_rr(hipStream_t st, const ConvParams& p, int ntiles, float act_slope) {
    constexpr size_t lds = (size_t)RR_PH * RR_PW * RR_PITCH + (size_t)4 * 32 * RR_TP;       // 66.9 + 10.2 KB: two workgroups per CU
    auto kern = conv7x7_rr_kernel<NCH, COUT8>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { dsvc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return DSVC_E_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), lds, st, p, act_slope);
    DSVC_CHECK_LAUNCH("conv7x7_rr_kernel");
    return DSVC_OK;
}

}  // namespace

// returns 1 when the shape is not covered (the caller goes on to the other kernels), else DSVC_OK / error
int dsvc_conv7x7_rr_try(hipStream_t st, const ConvParams& p_in, const dsvc_conv_desc* d) {
    if (d->compute != DSVC_F16 || d->kh != 7 || d->kw != 7 || d->stride != 1 || d->pad != 3 || d->shuffle != 1) return 1;
    if (p_in.nseg != 1 || d->res1 || d->res2 || d->res1_scale || d->pre_act != DSVC_ACT_NONE || d->post_weight || p_in.s2d) return 1;
    if (p_in.w_bs != 0 || p_in.out_bs != 0 || (p_in.b_inner != INT_MAX && d->B > 1)) return 1;        // dense batches with shared weights only
    const ConvSeg& sg = p_in.seg[0];
    if (sg.dtype != DSVC_F16 || (sg.C != 32 && sg.C != 64) || (sg.ld % 8) != 0 || ((uintptr_t)sg.ptr & 15) != 0) return 1;
    const bool shape = (sg.C == 32 && (d->Cout == 64 || d->Cout == 16 || d->Cout == 32)) || (sg.C == 64 && d->Cout == 32);
    if (!shape) return 1;
    if (d->out_dtype != DSVC_F16 || ((uintptr_t)d->out & 15) != 0 || (d->out_ld % 8) != 0) return 1;
    float act_slope = 1.f;                                      // act(x) = max(x, x * slope): none (1), ReLU (0), LeakyReLU
    if (d->act == DSVC_ACT_RELU) act_slope = 0.f;
    else if (d->act == DSVC_ACT_LRELU && d->slope >= 0.f && d->slope <= 1.f) act_slope = d->slope;
    else if (d->act != DSVC_ACT_NONE) return 1;
    if ((int64_t)d->B * d->H * d->W * sg.ld * 2 >= (1ll << 31) || (int64_t)d->B * d->H * d->W * d->out_ld * 2 >= (1ll << 31)) return 1;   // 32-bit byte offsets
    ConvParams p = p_in;
    p.tiles_x = (p.Wo + RR_TW - 1) / RR_TW;
    p.tiles_y = (p.Ho + RR_TH - 1) / RR_TH;
    const int64_t nt = (int64_t)p.B * p.tiles_x * p.tiles_y;
    if (nt < 128 || nt > INT_MAX) return 1;        // small images stay on the general kernel
    if (sg.C == 64) return launch_rr<2, 4>(st, p, (int)nt, act_slope);
    if (d->Cout == 64) return launch_rr<1, 8>(st, p, (int)nt, act_slope);
    if (d->Cout == 32) return launch_rr<1, 4>(st, p, (int)nt, act_slope);
    if (g_dsvc_opt.rr7 >= 2) return launch_rr<1, 2>(st, p, (int)nt, act_slope);       // (A/B: the 32-row-tile variant with half of its rows padding)
    return launch_rr16(st, p, (int)nt, act_slope);
}

// conv_cin8.hip -- f16 stride-1 convs over ONE 8-channel input (16-byte pixels) with the x-taps of a kernel row packed into the MFMA's K axis:
// ME_Spynet's first 7x7 conv 8 -> 32 on every pyramid level (modules.py:127-131: cat(ref, warped cur, flow) = 3 + 3 + 2 channels, merged into
// one f16 buffer by modules.Conv2d._pack_small_runs) and the 3x3 convs 8 -> 64 over a frame padded to 8 channels (modules.py:262-277 / 401).
//
// The general kernel pads every tap's 8 channels to a 32-channel chunk: 2 k-steps of 16 per tap, 98 per 7x7 tile row where 24.5 carry data
// (294 us for 8 -> 32 at 1088x1920, 0.07 of the matrix peak; 86 us for 3x3 8 -> 64, whose output alone streams in 50).  With 8 channels a pixel
// is 16 bytes and the pixels of an image row are CONTIGUOUS, so im2col along x costs nothing: k-step j of kernel row ky covers taps kx = 2j,
// 2j + 1 -- lane (r, h) of the pixel operand reads the 16 bytes of patch pixel (y + ky, x0 + r + 2j + h), consecutive lanes consecutive
// addresses (conflict-free ds_read_b128) -- and a 7x7 conv is 7 x 4 = 28 k-steps (taps 0 .. 6 and one zero column), a 3x3 conv 3 x 2 = 6.
//   * operand roles are swapped (D[channel][pixel]) so that, after one v_permlane32_swap per register pair, each lane holds 8 consecutive output
//     channels of its pixel (bias preloaded into the accumulators, activation as one v_med3: conv1x1.hip's register epilogue); a wave-private
//     LDS tile then turns those 16-byte pieces into whole output rows, 1 KB of consecutive bytes per store instruction;
//   * the weight fragments come straight from the MFMA-packed buffer ([tap][Np rows] x 32 halfs, channels 0 .. 7 of a row = one 16-byte k-half)
//     into LDS once per PERSISTENT workgroup, lane-contiguous (28 KB for 7x7 -> 32, 12 KB for 3x3 -> 64);
//   * a wave owns two output rows of an 8x32-pixel tile and walks the kernel rows: per ky it reads the weight fragments once and the pixel
//     fragments of both rows (1.5 LDS reads per MFMA for 7x7 -> 32, 1.0 for 3x3 -> 64);
//   * the next tile's patch vectors are loaded into registers while the current tile is multiplied; the patch's extra column (tap kx = KS of
//     the padded k-step) is staged as zeros, so no value outside the receptive field ever meets a weight -- not even a zero one.
// Summation order differs from the general kernel's (taps pairwise inside one MFMA; fp32 sums either way): f16-mode tolerance.
#include <climits>
#include "conv_common.h"

using namespace dsvc_conv;

namespace {

constexpr int C8_TH = 8, C8_TW = 32;

// KS: 3 or 7.  NT: 32-channel output tiles (Cout = 32 NT).
template <int KS, int NT>
__global__ void __launch_bounds__(256) conv_cin8_kernel(const ConvParams p, int ntiles, float act_slope) {
    DSVC_F16_SAT();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KX = (KS + 1) / 2;                           // k-steps per kernel row (two taps each)
    constexpr int PH = C8_TH + KS - 1, PW = C8_TW + KS, NV = PH * PW, NLD = (NV + 255) / 256;
    constexpr int NF = KS * KX * NT;                           // weight fragments of 1 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    char* lds_w = smem;                                        // [ky][j][n] x 64 lanes x 16 B
    char* patch = smem + NF * 1024;                            // [PH][PW] x 16 B
    constexpr int PPO = NT * 4, TP = (PPO + 1) * 16;           // 16-byte pieces per output pixel row; output tile pitch
    constexpr int REGION = NV * 16 > 4 * 32 * TP ? NV * 16 : 4 * 32 * TP;
    float* lds_bias = (float*)(patch + REGION);                // Np floats
    char* otile = patch + wave * (32 * TP);                    // wave-private output tile (one output row of 32 pixels), over the dead patch
    auto tile_sync = [&]() { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); };   // DS ops of a wave execute in order: only the compiler must not reorder
    // ---- weight fragments: lane (r, h) of fragment (ky, j, n) = channels 0 .. 7 of packed row (tap (ky, 2j + h), output 32 n + r) ----
    for (int f = tid; f < NF * 64; f += 256) {
        const int l = f & 63, fr = f >> 6;
        const int n = fr % NT, j = (fr / NT) % KX, ky = fr / (NT * KX);
        const int kx = 2 * j + (l >> 5);
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        if (kx < KS) w = *(const uint4*)((const char*)p.w + ((size_t)(ky * KS + kx) * p.Np + n * 32 + (l & 31)) * 64);
        *(uint4*)(lds_w + f * 16) = w;
    }
    for (int c = tid; c < p.Np; c += 256) lds_bias[c] = p.bias ? p.bias[c] : 0.f;

    const rsrc_t rin = make_rsrc(p.seg[0].ptr, p.seg[0].bytes);
    const rsrc_t rout = make_rsrc(p.out, p.out_bytes);
    const unsigned opitch = (unsigned)p.out_ld * 2u;
    auto tile_of = [&](int t, int& b, int& oy0, int& ox0) {
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        b = t / p.tiles_y; oy0 = ty * C8_TH; ox0 = tx * C8_TW;
    };
    // patch vector v = tid + 256 i -> patch pixel (v / PW, v % PW); the last column is the zero tap of the padded k-step
    auto issue = [&](int t, uint4 (&v)[NLD]) {
        int b, oy0, ox0;
        const bool live = t < ntiles;
        tile_of(live ? t : 0, b, oy0, ox0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int vi = tid + 256 * i;
            const int py = vi / PW, px = vi - py * PW;
            const int iy = oy0 - KS / 2 + py, ix = ox0 - KS / 2 + px;
            const bool inb = live && vi < NV && px < PW - 1 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            // (offset | OOB, not a select: hipcc turns a select of offsets into exec-mask branches around each load and sinks them behind the MFMA phase)
            v[i] = buf_load16(rin, ((unsigned)(((b * p.H + iy) * p.W + ix) * 16) & 0x7FFFFFFFu) | (inb ? 0u : OOB));
        }
    };
    uint4 v[NLD];
    issue((int)blockIdx.x, v);
    for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int vi = tid + 256 * i;
            if (vi < NV) *(uint4*)(patch + vi * 16) = v[i];
        }
        lds_barrier();                                         // (first pass: also the weights and the bias)
        issue(t + (int)gridDim.x, v);
        __builtin_amdgcn_sched_barrier(0);                     // the prefetch is issued HERE: it has the whole MFMA phase to land
        f32x16 acc[2][NT];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {               // accumulator row (reg i) = channel (i & 3) + 8 (i >> 2) + 4 h
                    const float4 bq = *(const float4*)(lds_bias + n * 32 + 8 * qd + 4 * h);
                    acc[rr][n][4 * qd] = bq.x; acc[rr][n][4 * qd + 1] = bq.y; acc[rr][n][4 * qd + 2] = bq.z; acc[rr][n][4 * qd + 3] = bq.w;
                }
        const char* prow = patch + ((2 * wave) * PW + r + h) * 16;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            uint4 bw[KX][NT];
#pragma unroll
            for (int j = 0; j < KX; ++j)
#pragma unroll
                for (int n = 0; n < NT; ++n) bw[j][n] = *(const uint4*)(lds_w + ((ky * KX + j) * NT + n) * 1024 + lane * 16);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int j = 0; j < KX; ++j) {
                    const uint4 a = *(const uint4*)(prow + ((rr + ky) * PW + 2 * j) * 16);
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[rr][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, bw[j][n]), __builtin_bit_cast(h16x8, a), acc[rr][n], 0, 0, 0);
                }
        }
        lds_barrier();                                         // every wave has read the patch: the output tiles / the next tile's patch replace it
        // ---- epilogue: lane (r, h) -> 8 consecutive channels of pixel r per (N-tile, half P): activation, f16 pack; a wave-private LDS tile
        //      (pitch = row + 16 bytes: conflict-free both ways) turns the 32-byte pieces into WHOLE output rows: every store instruction
        //      writes 1 KB of consecutive bytes (16 pixels x 64 B or 8 x 128 B; element-piece stores ran 3x3 8 -> 64 at 2.8 TB/s) ----
        int b, oy0, ox0;
        tile_of(t, b, oy0, ox0);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int P = 0; P < 2; ++P) {
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xq = acc[rr][n][8 * P + j], yq = acc[rr][n][8 * P + 4 + j];   // scalars first: hipcc 7.2 mis-casts vector elements
                        const v2u_t sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false);
                        const unsigned lo_ = sw.x, hi_ = sw.y;
                        o[j] = __builtin_bit_cast(float, lo_); o[4 + j] = __builtin_bit_cast(float, hi_);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = __builtin_amdgcn_fmed3f(o[j], o[j] * act_slope, __builtin_inff());
                    v4u_t ov;
                    ov.x = pk_h16(o[0], o[1]); ov.y = pk_h16(o[2], o[3]); ov.z = pk_h16(o[4], o[5]); ov.w = pk_h16(o[6], o[7]);
                    *(v4u_t*)(otile + r * TP + (n * 4 + 2 * P + h) * 16) = ov;
                }
            tile_sync();
            const int oy = oy0 + 2 * wave + rr;
            const unsigned rowpix = (unsigned)((b * p.Ho + oy) * p.Wo + ox0);
#pragma unroll
            for (int i = 0; i < PPO / 2; ++i) {                // 32 pixels x PPO pieces = PPO / 2 instructions of 64 lanes
                const int vv = i * 64 + lane, px = vv / PPO, pc = vv - px * PPO;
                const uint4 ou = *(const uint4*)(otile + px * TP + pc * 16);
                v4u_t ov; ov.x = ou.x; ov.y = ou.y; ov.z = ou.z; ov.w = ou.w;
                const bool ok = oy < p.Ho && ox0 + px < p.Wo;
                __builtin_amdgcn_raw_buffer_store_b128(ov, rout, ok ? (rowpix + (unsigned)px) * opitch + (unsigned)pc * 16u : OOB, 0, 0);
            }
            tile_sync();
        }
        lds_barrier();                                         // the output tiles share the patch's LDS
    }
}

template <int KS, int NT>
int launch_cin8(hipStream_t st, const ConvParams& p, int ntiles, float act_slope) {
    constexpr int KX = (KS + 1) / 2;
    constexpr size_t patch = (size_t)(C8_TH + KS - 1) * (C8_TW + KS) * 16, otiles = (size_t)4 * 32 * (NT * 4 + 1) * 16;
    const size_t lds = (size_t)KS * KX * NT * 1024 + (patch > otiles ? patch : otiles) + (size_t)p.Np * sizeof(float);
    const int grid = ntiles < 256 * 4 ? ntiles : 256 * 4;      // four workgroups per CU (<= 39 KB of LDS each)
    hipLaunchKernelGGL((conv_cin8_kernel<KS, NT>), dim3(grid), dim3(256), lds, st, p, ntiles, act_slope);
    DSVC_CHECK_LAUNCH("conv_cin8_kernel");
    return DSVC_OK;
}

}  // namespace

// returns 1 when the shape is not covered (the caller goes on to the other kernels), else DSVC_OK / error
int dsvc_conv_cin8_try(hipStream_t st, const ConvParams& p_in, const dsvc_conv_desc* d) {
    if (d->compute != DSVC_F16 || d->kh != d->kw || (d->kh != 3 && d->kh != 7) || d->stride != 1 || d->pad != d->kh / 2 || d->shuffle != 1) return 1;
    if (p_in.nseg != 1 || d->res1 || d->res2 || d->res1_scale || d->pre_act != DSVC_ACT_NONE || d->post_weight || p_in.s2d) return 1;
    if (p_in.w_bs != 0 || p_in.out_bs != 0 || (p_in.b_inner != INT_MAX && d->B > 1)) return 1;        // dense batches with shared weights only
    const ConvSeg& sg = p_in.seg[0];
    if (sg.dtype != DSVC_F16 || sg.C != 8 || sg.ld != 8 || ((uintptr_t)sg.ptr & 15) != 0) return 1;     // contiguous 16-byte pixels
    if (!((d->kh == 7 && d->Cout == 32) || (d->kh == 3 && (d->Cout == 32 || d->Cout == 64)))) return 1;
    if (d->out_dtype != DSVC_F16 || ((uintptr_t)d->out & 15) != 0 || (d->out_ld % 8) != 0) return 1;
    float act_slope = 1.f;                                      // act(x) = max(x, x * slope): none (1), ReLU (0), LeakyReLU
    if (d->act == DSVC_ACT_RELU) act_slope = 0.f;
    else if (d->act == DSVC_ACT_LRELU && d->slope >= 0.f && d->slope <= 1.f) act_slope = d->slope;
    else if (d->act != DSVC_ACT_NONE) return 1;
    if ((int64_t)d->B * d->H * d->W * 16 >= (1ll << 31) || (int64_t)d->B * d->H * d->W * d->out_ld * 2 >= (1ll << 31)) return 1;   // 32-bit byte offsets
    ConvParams p = p_in;
    p.tiles_x = (p.Wo + C8_TW - 1) / C8_TW;
    p.tiles_y = (p.Ho + C8_TH - 1) / C8_TH;
    const int64_t nt = (int64_t)p.B * p.tiles_x * p.tiles_y;
    if (nt < 64 || nt > INT_MAX) return 1;         // tiny images stay on the general kernel
    if (d->kh == 7) return launch_cin8<7, 1>(st, p, (int)nt, act_slope);
    return d->Cout == 64 ? launch_cin8<3, 2>(st, p, (int)nt, act_slope) : launch_cin8<3, 1>(st, p, (int)nt, act_slope);
}

// conv1x1_f32.hip -- fp32 (parity mode) pointwise convolution as a streaming kernel: conv1 / conv3 of every ResBottleneckBlock
// (modules.py:176,178; image_model.py:25,27) and the 1x1 shortcuts, 64 / 96 / 128 channels, whole images.
//
// In fp32 mode these launches went through the general implicit-GEMM kernel: halo-free patches staged through LDS behind workgroup
// barriers, 16 pixels x 1 channel per lane in the accumulators and a slab pass to get back to channel rows -- 1.0-3.1 TB/s on an
// operation that moves 8-12 bytes per MAC column (tools/conv_time_census.py with CENSUS_MODE=fp32: 25 ms of a 186 ms frame).
//
// Same structure as the f16 streaming kernel (conv1x1.hip), re-derived for 16-byte = 4-channel pieces:
//   * the image is a flat list of pixels; a wave owns 32 consecutive pixels at a time (persistent waves, grid-stride);
//   * global memory is touched in whole pixel rows (a load / store instruction moves 1 KB of complete rows); a wave-private LDS tile
//     (pitch = row + 16 bytes: conflict-free both ways) transposes between that layout and the MFMA operand layout; the next unit's
//     rows -- and its residual rows -- are in flight while the current unit's MFMAs run; no workgroup barrier after the prologue;
//   * weights ([chunk of 16 input channels][output channel] rows of 64 bytes, the general kernel's packing) sit in LDS for the whole launch;
//   * operand roles are swapped (D[channel][pixel] = W x X): a lane's 16 accumulator registers of an N-tile are four 4-channel vectors
//     of ITS pixel, so the epilogue writes 16-byte pieces straight into the tile's row layout.
//
// Bit-identical to the general kernel, which the parity fixtures pin: v_mfma_f32_32x32x2_f32 with the same channel pair (c, c + 8) of
// each 16-channel chunk per instruction, chunks / pairs in the same order, accumulators starting at zero; bias, activation and the
// residual are applied in the general epilogue's order (conv_common.h: epilogue_vec).  Swapping the operand roles only transposes D --
// asserted element for element by tests/test_gpu_ops.py::test_fp32_pointwise_kernel_matches_general_kernel.
#include <cstdlib>
#include "conv_common.h"

using namespace dsvc_conv;

namespace {

// KC: 16-channel input chunks (of up to two segments: the virtual torch.cat of the up-path shortcuts); NT: 32-channel output tiles;
// HAS_GATE: out = act(conv) + gate[c] * res1 (the SE gate of modules.py:226-247, batch 1)
template <int KC, int NT, bool HAS_R1, bool HAS_GATE>
__global__ void __launch_bounds__(256) conv1x1_f32_kernel(const ConvParams p, int nunits, int act, float slope) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PPI = KC * 4, PPO = NT * 8;                        // 16-byte pieces per input / output pixel row
    constexpr int TP = ((PPI > PPO ? PPI : PPO) + 1) * 16;          // tile pitch in bytes
    constexpr int NLI = PPI / 2, NLO = PPO / 2;                      // pieces per lane per unit (32 pixels over 64 lanes)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    char* lds_w = smem;                                              // [chunk][Np rows] at the 80-byte pitch
    float* lds_bias = (float*)(smem + (size_t)KC * p.Np * ROWB);
    char* tile = smem + (size_t)KC * p.Np * ROWB + (size_t)2 * p.Np * sizeof(float) + (size_t)wave * (32 * TP);
    {
        const int nrows = KC * p.Np, q = tid & 3;
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
    const int pp0 = two ? p.seg[0].C / 4 : PPI;                      // pieces of a row served by segment 0
    const unsigned pitch1 = (unsigned)p.seg[two ? 1 : 0].ld * 4;
    const rsrc_t rout = make_rsrc(p.out, p.out_bytes);
    const rsrc_t rr1 = make_rsrc(HAS_R1 ? p.res1 : p.out, HAS_R1 ? p.res1_bytes : p.out_bytes);
    const unsigned pitch = (unsigned)p.seg[0].ld * 4, opitch = (unsigned)p.out_ld * 4, rpitch = (unsigned)p.res1_ld * 4;
    const unsigned npix = (unsigned)p.Wo;                           // host presents the tensor as one row of B*H*W pixels
    const int nwaves = gridDim.x * 4;
    int unit = blockIdx.x * 4 + wave;
    auto load_g = [&](int u, uint4 (&g)[NLI]) {                      // input rows of unit u: vector v = 64 i + lane -> (pixel v / PPI, piece v % PPI)
        const bool uok = u < nunits;
#pragma unroll
        for (int i = 0; i < NLI; ++i) {
            const int v = i * 64 + lane, px = v / PPI, pc = v - px * PPI;
            const unsigned pix = (unsigned)u * 32 + px;
            const unsigned m = (uok && pix < npix) ? 0u : OOB;
            // (two segments: BOTH descriptors are read, each lane masked out of the one that is not its own -- a per-lane choice of descriptor
            //  compiles to a waterfall loop over the descriptor values around every load)
            if (!two) g[i] = buf_load16(rin, (pix * pitch + (unsigned)pc * 16u) | m);
            else {
                const uint4 a = buf_load16(rin, (pix * pitch + (unsigned)pc * 16u) | m | (pc < pp0 ? 0u : OOB));
                const uint4 b = buf_load16(rin1, (pix * pitch1 + (unsigned)(pc - pp0) * 16u) | m | (pc < pp0 ? OOB : 0u));
                g[i] = make_uint4(a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w);
            }
        }
    };
    auto load_gr = [&](int u, uint4 (&g)[NLO]) {                     // residual rows in the layout of the output rows
        const bool uok = u < nunits;
#pragma unroll
        for (int i = 0; i < NLO; ++i) {
            const int v = i * 64 + lane, px = v / PPO, pc = v - px * PPO;
            const unsigned pix = (unsigned)u * 32 + px;
            g[i] = buf_load16(rr1, (pix * rpitch + (unsigned)pc * 16u) | ((uok && pix < npix) ? 0u : OOB));
        }
    };
    auto tile_sync = [&]() { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); };   // DS ops of a wave execute in order: only the compiler must not reorder
    uint4 a_nxt[NLI], r_nxt[HAS_R1 ? NLO : 1];
    load_g(unit, a_nxt);
    if constexpr (HAS_R1) load_gr(unit, r_nxt);
    const int a_lane = r * TP + h * 32, w_lane = r * ROWB + h * 32;
    for (; unit < nunits; unit += nwaves) {
        // rows -> tile
#pragma unroll
        for (int i = 0; i < NLI; ++i) { const int v = i * 64 + lane, px = v / PPI, pc = v - px * PPI; *(uint4*)(tile + px * TP + pc * 16) = a_nxt[i]; }
        tile_sync();
        load_g(unit + nwaves, a_nxt);                              // prefetch (masked past the end)
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
        // lane (r, h), k-step ks of chunk c: channels 16 c + 8 h + 4 ks + j (j = 0..3) of pixel r / of output channel 32 n + r:
        // MFMA j pairs channel 16 c + 4 ks + j (k slot 0) with channel 16 c + 8 + 4 ks + j (k slot 1), as conv_mfma_kernel<float> does
#pragma unroll
        for (int c = 0; c < KC; ++c) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f32x4 a4 = __builtin_bit_cast(f32x4, *(const uint4*)(tile + a_lane + c * 64 + ks * 16));
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f32x4 b4 = __builtin_bit_cast(f32x4, *(const uint4*)(lds_w + (c * p.Np + n * 32) * ROWB + w_lane + ks * 16));
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[j], a4[j], acc[n], 0, 0, 0);
                }
            }
        }
        tile_sync();                                                // every operand read of the tile is done
        // accumulator register i of N-tile n = output channel 32 n + (i & 3) + 8 (i >> 2) + 4 h of pixel r: piece 8 n + 2 (i >> 2) + h
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq = *(const float4*)(lds_bias + n * 32 + 8 * g + 4 * h);
                const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = acc[n][4 * g + j] + bb[j];
                    const float neg = act == DSVC_ACT_RELU ? 0.f : t * slope;
                    v[j] = act == DSVC_ACT_NONE ? t : (t > 0.f ? t : neg);
                }
                *(float4*)(tile + r * TP + (n * 8 + 2 * g + h) * 16) = make_float4(v[0], v[1], v[2], v[3]);
            }
        tile_sync();
        // tile -> whole output rows (+ residual, which was loaded in this very layout)
#pragma unroll
        for (int i = 0; i < NLO; ++i) {
            const int v = i * 64 + lane, px = v / PPO, pc = v - px * PPO;
            float4 o = *(const float4*)(tile + px * TP + pc * 16);
            if constexpr (HAS_R1) {
                const uint4 rv = r_nxt[i];
                const unsigned ru[4] = {rv.x, rv.y, rv.z, rv.w};
                if constexpr (HAS_GATE) {        // the general epilogue's v += gate * r1 + r2 with r2 = +0: the product is rounded, then added
                    const float4 gt = *(const float4*)(lds_bias + p.Np + pc * 4);
                    o.x += __fadd_rn(__fmul_rn(gt.x, __builtin_bit_cast(float, ru[0])), 0.f); o.y += __fadd_rn(__fmul_rn(gt.y, __builtin_bit_cast(float, ru[1])), 0.f);
                    o.z += __fadd_rn(__fmul_rn(gt.z, __builtin_bit_cast(float, ru[2])), 0.f); o.w += __fadd_rn(__fmul_rn(gt.w, __builtin_bit_cast(float, ru[3])), 0.f);
                } else {
                    o.x += __builtin_bit_cast(float, ru[0]); o.y += __builtin_bit_cast(float, ru[1]);
                    o.z += __builtin_bit_cast(float, ru[2]); o.w += __builtin_bit_cast(float, ru[3]);
                }
            }
            const unsigned opx = (unsigned)unit * 32 + px;
            v4u_t ov;
            ov.x = __builtin_bit_cast(unsigned, o.x); ov.y = __builtin_bit_cast(unsigned, o.y);
            ov.z = __builtin_bit_cast(unsigned, o.z); ov.w = __builtin_bit_cast(unsigned, o.w);
            __builtin_amdgcn_raw_buffer_store_b128(ov, rout, (opx * opitch + (unsigned)pc * 16u) | (opx < npix ? 0u : OOB), 0, 0);
        }
        if constexpr (HAS_R1) load_gr(unit + nwaves, r_nxt);
        tile_sync();
    }
}

template <int KC, int NT>
int launch_f32(hipStream_t st, const ConvParams& p, int nunits, int act, float slope) {
    constexpr int PPI = KC * 4, PPO = NT * 8, TP = ((PPI > PPO ? PPI : PPO) + 1) * 16;
    const size_t lds = (size_t)KC * p.Np * ROWB + (size_t)2 * p.Np * sizeof(float) + (size_t)4 * 32 * TP;
    const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : ((int)((160 * 1024) / lds) > 4 ? 4 : (int)((160 * 1024) / lds));
    int grid = (nunits + 3) / 4;
    if (grid > 256 * per_cu) grid = 256 * per_cu;
#define C1F_LAUNCH(R1, GATE)                                                                                                      \
    do {                                                                                                                          \
        auto kern = conv1x1_f32_kernel<KC, NT, R1, GATE>;                                                                             \
        static bool attr_set = false;                                                                                             \
        if (!attr_set && lds > 64 * 1024) {                                                                                       \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
            if (e != hipSuccess) { dsvc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return DSVC_E_LAUNCH; }       \
            attr_set = true;                                                                                                      \
        }                                                                                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p, nunits, act, slope);                                          \
    } while (0)
    if (p.res1 && p.res1_scale) C1F_LAUNCH(true, true);
    else if (p.res1) C1F_LAUNCH(true, false);
    else C1F_LAUNCH(false, false);
#undef C1F_LAUNCH
    DSVC_CHECK_LAUNCH("conv1x1_f32_kernel");
    return DSVC_OK;
}

template <int KC>
int launch_f32_nt(hipStream_t st, const ConvParams& p, int nunits, int NT, int act, float slope) {
    switch (NT) {
        case 1: return launch_f32<KC, 1>(st, p, nunits, act, slope);
        case 2: return launch_f32<KC, 2>(st, p, nunits, act, slope);
        case 3: return launch_f32<KC, 3>(st, p, nunits, act, slope);
        default: return launch_f32<KC, 4>(st, p, nunits, act, slope);
    }
}

}  // namespace

// returns 1 when the shape is not covered (caller falls back to the general kernel), else DSVC_OK / error
int dsvc_conv1x1_f32_try(hipStream_t st, const ConvParams& p_in, const dsvc_conv_desc* d) {
    if (d->compute != DSVC_F32 || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->shuffle != 1) return 1;
    if (d->nseg > 2 || d->pre_act != DSVC_ACT_NONE || d->res2 || d->post_weight || (d->res1_scale && (d->B > 1 || !d->res1))) return 1;
    if (p_in.out_bs || p_in.w_bs || p_in.res1_bs) return 1;       // per-item weights / placement: the general kernel's
    if (d->act != DSVC_ACT_NONE && d->act != DSVC_ACT_RELU && d->act != DSVC_ACT_LRELU) return 1;
    auto v16 = [](const void* ptr, int ld) { return ((uintptr_t)ptr & 15) == 0 && (ld % 4) == 0; };
    int ctot = 0, ldin = 0;
    for (int s = 0; s < d->nseg; ++s) {                        // whole 16-channel chunks of aligned fp32 rows
        const ConvSeg& g = p_in.seg[s];
        if (g.dtype != DSVC_F32 || !v16(g.ptr, g.ld) || (g.C % 16) != 0) return 1;
        ctot += g.C;
        ldin = g.ld > ldin ? g.ld : ldin;
    }
    if (ctot != 64 && ctot != 96 && ctot != 128) return 1;
    if (d->Cout != 32 && d->Cout != 64 && d->Cout != 96 && d->Cout != 128) return 1;
    if (d->out_dtype != DSVC_F32 || !v16(d->out, d->out_ld)) return 1;
    if (d->res1 && (d->res1_dtype != DSVC_F32 || !v16(d->res1, d->res1_ld))) return 1;
    const int64_t npix = (int64_t)d->B * d->H * d->W;
    if (npix < 16 * 1024) return 1;                            // small images (the entropy-parameter nets) stay on the general kernel
    const int64_t ldmax = ldin > d->out_ld ? (ldin > d->res1_ld ? ldin : d->res1_ld) : (d->out_ld > d->res1_ld ? d->out_ld : d->res1_ld);
    if ((npix + 64) * ldmax * 4 >= ((int64_t)1 << 31)) return 1;  // byte offsets are 31-bit (the masked value has bit 31 set)
    ConvParams p = p_in;
    p.B = 1; p.Ho = 1; p.Wo = (int)npix;
    const int nunits = (int)((npix + 31) / 32);
    const int NT = d->Cout / 32;
    switch (ctot / 16) {
        case 4: return launch_f32_nt<4>(st, p, nunits, NT, d->act, d->slope);
        case 6: return launch_f32_nt<6>(st, p, nunits, NT, d->act, d->slope);
        default: return launch_f32_nt<8>(st, p, nunits, NT, d->act, d->slope);
    }
}

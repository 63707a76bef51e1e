// conv3x3_direct.hip -- bf16-MFMA 3x3 stride-1 convolution for SMALL images: the entropy-parameter / latent-residual /
// hyper-synthesis nets of the slice scan (image_model.py:96-146,176-196 -- cc_mean / cc_scale / lrp transforms, h_s)
// at 1/16 .. 1/64 of the frame resolution: ~390 launches per P-frame, each a few MFLOP..2 GFLOP on L2-resident data.
//
// At that size a conv is pure LATENCY: the generic kernel spends 19-29 us per launch walking its
// stage-to-LDS / barrier / MFMA sequence once per 32-channel chunk (rocprofv3, profiles/), with ~70 % of the CUs idle.
// This kernel has no LDS and no barriers:
//   * one WAVE owns 32 consecutive pixels of one image row x 32 output channels (an MFMA 32x32 tile); four
//     independent waves per workgroup; the N-tiles of the same pixels sit in the same workgroup (L1 reuse);
//   * MFMA fragments come STRAIGHT from global memory in operand layout: lane (r, h) loads the 16 B holding input
//     channels [16h + 8ks, +8) of pixel x0 + r + kx - 1 (bounds-checked buffer loads: zero padding, ragged row ends
//     and short channel chunks read zeros; fp32 segments take two loads and are rounded to bf16 in registers), and
//     the matching 16 B of the packed weight row of output channel n0 + r;
//   * all loads are independent and the whole K loop is STRAIGHT-LINE code (the chunk count is a template parameter,
//     segment / dtype choices are selects, never branches): hipcc then emits counted vmcnt waits, so the fragments of
//     the next two stages (stage = one kernel row of one chunk: 3 taps x 2 k-halves) stay in flight while the 6 MFMAs
//     of the current one run -- a first version with a rolled loop and uniform `if`s drained vmcnt(0) at every
//     control-flow merge and paid one L2 round trip (~3 us) per stage;
//   * operand roles are swapped (A = weights, B = pixels) so that D[channel][pixel] leaves pixel r on the lane and,
//     after one v_permlane32_swap per register, 8 consecutive channels per lane: the fused epilogue (bias,
//     activation, residual, cast) works on registers and stores 16/32-byte vectors.
#include <cstdlib>
#include "conv_common.h"

using namespace dsvc_conv;

namespace {

struct DirectArgs {
    int nitems, units_x, ngroups;
};

template <int NCH, bool ALLBF16>
__global__ void __launch_bounds__(256) conv3x3_direct_kernel(const ConvParams p, const DirectArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.nitems) return;                                  // no barriers in this kernel: waves are independent
    const int r = lane & 31, h = lane >> 5;
    const int ng = item % a.ngroups, unit = item / a.ngroups;
    const int ux = unit % a.units_x, rest = unit / a.units_x;
    const int y = rest % p.H, b = rest / p.H;
    const int x0 = ux * 32, n0 = ng * 32;
    const int H = p.H, W = p.W;

    // weight fragment of (chunk, tap, k-half): row n0 + r of the packed [chunk][tap][Np] x 64 B table
    const rsrc_t rw = make_rsrc(p.w, (unsigned)p.nchunks * 9u * (unsigned)p.Np * 64u);
    const unsigned w_lane = (unsigned)(n0 + r) * 64u + (unsigned)h * 32u;

    // per-lane x coordinates of the three kernel columns (loop invariant)
    int xk[3];
    bool xok[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        xk[kx] = x0 + r + kx - 1;
        xok[kx] = xk[kx] >= 0 && xk[kx] < W;
    }

    // One stage = one kernel row ky of one 32-channel chunk: 3 taps x 2 k-halves = 6 MFMAs.
    struct Stage { uint4 a0[6], a1[6], w[6]; };                    // a1: second half of an fp32 fragment
    auto issue = [&](int st, Stage& S) {
        const int ch = st / 3, ky = st - ch * 3;
        const ConvSeg& sg = p.seg[(p.nseg > 2 && ch >= p.seg[2].chunk0) ? 2 : (p.nseg > 1 && ch >= p.seg[1].chunk0) ? 1 : 0];
        const rsrc_t rin = make_rsrc(sg.ptr, sg.bytes);
        const int cb = (ch - sg.chunk0) * 32;                        // first channel of the chunk inside the segment
        const int yy = y + ky - 1;
        const bool yok = yy >= 0 && yy < H;
        const bool f32 = !ALLBF16 && sg.dtype == DSVC_F32;
        const unsigned esz = f32 ? 4u : 2u;
        const unsigned rowpix = (unsigned)((b * H + yy) * W);       // scalar
        const unsigned ldb = (unsigned)sg.ld * esz;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const unsigned pixoff = __umul24(rowpix + (unsigned)xk[kx], ldb);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int c = cb + 16 * h + 8 * ks;                  // this lane's 8 input channels
                const bool ok = yok && xok[kx] && c < sg.C;          // C % 8 == 0 on this path
                // masked lanes get bit 31 OR-ed into an offset that is computed unconditionally (extents are < 2 GiB,
                // so the result is out of range): a select of two constants, where `ok ? off : OOB` made hipcc wrap the
                // address arithmetic in exec-mask branches -- and drain vmcnt(0) at every one of their merges
                const unsigned off = pixoff + (unsigned)c * esz;
                S.a0[kx * 2 + ks] = buf_load16(rin, off | (ok ? 0u : OOB));
                if constexpr (!ALLBF16) S.a1[kx * 2 + ks] = buf_load16(rin, (off + 16) | ((ok && f32) ? 0u : OOB));
                S.w[kx * 2 + ks] = buf_load16(rw, w_lane + (unsigned)((ch * 9 + ky * 3 + kx) * p.Np) * 64u + (unsigned)ks * 16u);
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    auto compute = [&](int st, const Stage& S) {
        bool f32 = false;
        if constexpr (!ALLBF16) {
            const int ch = st / 3;
            const ConvSeg& sg = p.seg[(p.nseg > 2 && ch >= p.seg[2].chunk0) ? 2 : (p.nseg > 1 && ch >= p.seg[1].chunk0) ? 1 : 0];
            f32 = sg.dtype == DSVC_F32;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            uint4 af = S.a0[j];
            if constexpr (!ALLBF16) {                                // 8 fp32 -> 8 bf16 (RNE), selected, not branched
                const uint4 lo = S.a0[j], hi = S.a1[j];
                const unsigned u[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};   // scalars before bit_cast
                const unsigned raw[4] = {lo.x, lo.y, lo.z, lo.w};
                unsigned pk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned cv = (unsigned)f32_to_bf16_bits(__builtin_bit_cast(float, u[2 * k])) |
                                        ((unsigned)f32_to_bf16_bits(__builtin_bit_cast(float, u[2 * k + 1])) << 16);
                    pk[k] = f32 ? cv : raw[k];
                }
                af = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, S.w[j]), __builtin_bit_cast(bf16x8, af), acc, 0, 0, 0);
        }
    };

    // bias of this lane's 2 x 8 output channels, requested before the K loop so that it costs no round trip later
    float bias[16];
#pragma unroll
    for (int P = 0; P < 2; ++P)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = n0 + 16 * P + 8 * h + j;
            bias[8 * P + j] = (p.bias && c < p.Cout) ? p.bias[c] : 0.f;
        }

    constexpr int NST = NCH * 3;
    Stage S[3];                                                      // fully unrolled below: static register sets
    issue(0, S[0]);
    if constexpr (NST > 1) issue(1, S[1]);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        if (st + 2 < NST) issue(st + 2, S[(st + 2) % 3]);
        compute(st, S[st % 3]);
    }

    // ---- epilogue: D[channel][pixel] -> 8 consecutive channels per lane -> bias, activation, residual, store ----
    const int ox = x0 + r;
    const bool pix_ok = ox < W;
    const int64_t opix = ((int64_t)b * H + y) * W + ox;
#pragma unroll
    for (int P = 0; P < 2; ++P) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xq = acc[8 * P + j], yq = acc[8 * P + 4 + j];   // scalars first: hipcc 7.2 mis-casts vector elements
            const v2u_t sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false);
            const unsigned lo = sw.x, hi = sw.y;
            v[j] = __builtin_bit_cast(float, lo); v[4 + j] = __builtin_bit_cast(float, hi);
        }
        const int co = n0 + 16 * P + 8 * h;                          // this lane's channels co .. co+7
        if (!pix_ok || co >= p.Cout) continue;
        const int nv = min(8, p.Cout - co);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = apply_act(v[j] + bias[8 * P + j], p.act, p.slope);
        }
        if (p.res1) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nv) v[j] += load_elem(p.res1, p.res1_dtype, opix * p.res1_ld + co + j);
        }
        const int64_t o = opix * p.out_ld + co;
        if (nv == 8 && p.out_vec && (p.out_ld & 7) == 0 && ((uintptr_t)p.out & 15) == 0) {     // co % 8 == 0 always
            if (p.out_dtype == DSVC_F32) {
                float4* dst = (float4*)((float*)p.out + o);
                dst[0] = make_float4(v[0], v[1], v[2], v[3]);
                dst[1] = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                *(uint4*)((unsigned short*)p.out + o) = pack_vec<__bf16, 8>(v);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nv) store_elem(p.out, p.out_dtype, o + j, v[j]);
        }
    }
}

bool seg_vec16(const ConvSeg& sg) {
    const int esz = sg.dtype == DSVC_F32 ? 4 : 2;
    return ((uintptr_t)sg.ptr & 15) == 0 && ((sg.ld * esz) & 15) == 0 && (sg.C & 7) == 0;
}

}  // namespace

// returns 1 when the shape is not covered (caller continues with the other kernels), else DSVC_OK / error
int dsvc_conv3x3_direct_try(hipStream_t st, const ConvParams& p, const dsvc_conv_desc* d) {
    if (d->compute != DSVC_BF16 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->shuffle != 1) return 1;
    if (d->pre_act != DSVC_ACT_NONE || d->res1_scale || d->res2) return 1;
    const int64_t npix = (int64_t)p.B * p.H * p.W;
    const int64_t max_pix = g_dsvc_opt.direct_max_pix;
    if (npix > max_pix) return 1;                                     // larger images: LDS-tiled kernels reuse operands better
    bool allbf16 = true;
    for (int s = 0; s < p.nseg; ++s) {
        if (!seg_vec16(p.seg[s])) return 1;
        allbf16 = allbf16 && p.seg[s].dtype == DSVC_BF16;
    }
    DirectArgs a;
    a.units_x = (p.W + 31) / 32;
    a.ngroups = p.Np / 32;
    a.nitems = p.B * p.H * a.units_x * a.ngroups;
    const int grid = (a.nitems + 3) / 4;
#define DSVC_DIRECT_CASE(N)                                                                                        \
    case N:                                                                                                        \
        if (allbf16) hipLaunchKernelGGL((conv3x3_direct_kernel<N, true>), dim3(grid), dim3(256), 0, st, p, a);     \
        else hipLaunchKernelGGL((conv3x3_direct_kernel<N, false>), dim3(grid), dim3(256), 0, st, p, a);            \
        break;
    switch (p.nchunks) {
        DSVC_DIRECT_CASE(1) DSVC_DIRECT_CASE(2) DSVC_DIRECT_CASE(3) DSVC_DIRECT_CASE(4)
        DSVC_DIRECT_CASE(5) DSVC_DIRECT_CASE(6) DSVC_DIRECT_CASE(7) DSVC_DIRECT_CASE(8)
        default: return 1;
    }
#undef DSVC_DIRECT_CASE
    DSVC_CHECK_LAUNCH("conv3x3_direct_kernel");
    return DSVC_OK;
}

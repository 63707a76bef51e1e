// conv_common.h -- parameter block and fused epilogue shared by the conv kernels (conv_mfma.hip, conv3x3_ws.hip / conv3x3_pp.hip / conv1x1.hip).
#pragma once
#include "common.h"

namespace dsvc_conv {

constexpr int ROWB = 80;   // LDS row pitch in bytes
constexpr int TW = 32;     // tile width in output pixels == MFMA M

struct ConvSeg {
    const void* ptr;
    int C, ld, dtype, vec_ok, chunk0;
    unsigned bytes;      // extent from ptr to the end of the last pixel's slice (buffer-resource num_records)
    unsigned bs_lo, bs_hi;   // byte offset of batch item b = (b % b_inner) * bs_lo + (b / b_inner) * bs_hi (dense: b * image bytes)
};

struct ConvParams {
    ConvSeg seg[3];
    int nseg, nchunks;
    int B, H, W, Ho, Wo;
    int kh, kw, stride, pad;
    int Cout, Np;
    const void* w;
    const float* bias;
    int pre_act, act;
    float slope;
    const void* res1; int res1_ld, res1_dtype, res1_vec; const float* res1_scale;
    const void* res2; int res2_ld, res2_dtype, res2_vec;
    void* out; int out_ld, out_dtype, out_vec, shuffle;
    int tiles_x, tiles_y, tapg;
    unsigned out_bytes, res1_bytes, res2_bytes;   // extents for the buffer resources of the vector epilogue
    int epi_vec;                                   // 1: epilogue may use 4-channel vector buffer ops everywhere
    int epi_f32;                                   // 1: output and every residual are fp32 (fp32 mode): the vector epilogue issues only the fp32 instructions
    int s2d;                                       // 1: inputs are a space-to-depth view of a (2H x 2W) image (dsvc.h)
    const void* post_w; const float* post_bias;    // fused trailing 1x1 conv (dsvc.h: post_weight), conv3x3_pp only
    int b_inner;                                   // see ConvSeg::bs_lo (dense batch: INT_MAX)
    int skipk;                                     // 1: element-staged fp32 launches skip the all-zero MFMAs of narrow segments (conv_mfma_kernel: SKIPK)
    int patchpf;                                   // 1: ... with the next chunk's input patch prefetched into registers
    int regepi;                                    // 1: conv_mfma_kernel's register epilogue variant
    unsigned w_bytes;                              // extent of one item's packed weights (buffer-resource bound)
    long long w_bs; int bias_bs;                   // per-batch-item weights (dsvc.h: weight_bstride): bytes / floats per item, 0 = shared
    long long out_bs, res1_bs;                     // per-item output / residual placement (dsvc.h: out_bstride), 0 = dense batch
};

// ---- raw buffer resources: hardware bounds checking instead of branches ------------------------------------------
// A lane that must not touch memory -- masked pixel, absent operand, or the view of the other element type -- gets
// the out-of-range offset OOB (loads return 0, stores are dropped).  The code stays straight-line,
// so hipcc's waitcnt insertion emits counted waits at first use instead of vmcnt(0) at every control-flow merge.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
constexpr unsigned OOB = 0x80000000u;   // beyond any extent (< 2 GiB enforced on the host); offset + 16 cannot wrap

__device__ __forceinline__ rsrc_t make_rsrc(const void* ptr, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(rsrc_t r, unsigned off) {
    const v4u_t t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_uint4(t.x, t.y, t.z, t.w);
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the vector-memory queue, so
// prefetch loads and epilogue stores stay in flight across it (cdna_hip_programming.md, "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct EpiRsrc {      // one descriptor per operand (an absent residual borrows the output's; its offsets are all OOB)
    rsrc_t out, r1, r2;
    bool out_f32, r1_f32, r1_h16, r2_f32, r2_h16;
    // the epilogue's view of output / first residual / bias: the launch's, or -- per-item placement (dsvc.h: out_bstride) -- this item's,
    // with bias_bs = 0 and batch index 0.  Kept here as three scalars: a modified COPY of ConvParams lands on the stack (256 B of scratch
    // per thread plus the spills it provokes -- the slab-epilogue variants ran 3x slower for a while, tools/bench_subpel.py).
    void* outp; const void* r1p; const float* bias; int bias_bs;
};
__device__ __forceinline__ EpiRsrc make_epi_rsrc(const ConvParams& p, int item = 0) {
    EpiRsrc R;
    const bool placed = p.out_bs != 0;
    R.outp = placed ? (void*)((char*)p.out + (size_t)item * p.out_bs) : p.out;
    R.r1p = (placed && p.res1) ? (const void*)((const char*)p.res1 + (size_t)item * p.res1_bs) : p.res1;
    R.bias = (placed && p.bias) ? p.bias + (size_t)item * p.bias_bs : p.bias;
    R.bias_bs = placed ? 0 : p.bias_bs;
    R.out = make_rsrc(R.outp, p.out_bytes);
    R.r1 = make_rsrc(p.res1 ? R.r1p : R.outp, p.res1 ? p.res1_bytes : p.out_bytes);
    R.r2 = make_rsrc(p.res2 ? p.res2 : R.outp, p.res2 ? p.res2_bytes : p.out_bytes);
    R.out_f32 = p.out_dtype == DSVC_F32;
    R.r1_f32 = p.res1 && p.res1_dtype == DSVC_F32; R.r1_h16 = p.res1 && p.res1_dtype == DSVC_F16;
    R.r2_f32 = p.res2 && p.res2_dtype == DSVC_F32; R.r2_h16 = p.res2 && p.res2_dtype == DSVC_F16;
    return R;
}


// ---- fused epilogue pass over one wave-private LDS slab, specialised per activation -------------------------
// Lane -> (channel group g = lane % GPP fixed for the whole pass, pixel lanes step 64/GPP): bias/gate are loaded once,
// residuals and outputs move as 16-byte (fp32) / 8-byte (f16) vectors along the channel axis.
template <int ACT>
__device__ __forceinline__ float act_t(float v, float slope) {
    if constexpr (ACT == DSVC_ACT_RELU) return v > 0.f ? v : 0.f;
    else if constexpr (ACT == DSVC_ACT_LRELU) return v > 0.f ? v : v * slope;
    else if constexpr (ACT == DSVC_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    else if constexpr (ACT == DSVC_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    else if constexpr (ACT == DSVC_ACT_HALF_TANH) return 0.5f * tanhf(v);
    else return v;
}

__device__ __forceinline__ void load4(const void* base, int dtype, int vec, int64_t off, int nv, float (&o)[4]) {
    if (vec && nv == 4) {
        if (dtype == DSVC_F32) {
            const float4 t = *(const float4*)((const float*)base + off);
            o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
        } else {
            const uint2 t = *(const uint2*)((const unsigned short*)base + off);
            o[0] = h16_lo(t.x); o[1] = h16_hi(t.x);
            o[2] = h16_lo(t.y); o[3] = h16_hi(t.y);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = j < nv ? load_elem(base, dtype, off + j) : 0.f;
    }
}

template <int ACT, int GPP, int EPLD>
__device__ __forceinline__ void epilogue_pass(const ConvParams& p, const EpiRsrc& R, const float* slab, int lane, int npl, int b, int oy_base,
                                              int ox0, int co0, int cpass) {
    const int g = lane % GPP;
    const int co = co0 + g * 4;
    if (g * 4 >= cpass || co >= p.Cout) return;
    const int nv = min(4, p.Cout - co);
    float bias[4] = {0.f, 0.f, 0.f, 0.f}, gate[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (R.bias && j < nv) bias[j] = R.bias[b * R.bias_bs + co + j];
        if (p.res1_scale && j < nv) gate[j] = p.res1_scale[(int64_t)b * p.Cout + co + j];
    }
    for (int pl = lane / GPP; pl < npl; pl += 64 / GPP) {
        const int oy = oy_base + (pl >> 5), ox = ox0 + (pl & 31);
        if (oy >= p.Ho || ox >= p.Wo) continue;
        const float4 a4 = *(const float4*)(slab + pl * EPLD + g * 4);
        float v[4] = {act_t<ACT>(a4.x + bias[0], p.slope), act_t<ACT>(a4.y + bias[1], p.slope),
                      act_t<ACT>(a4.z + bias[2], p.slope), act_t<ACT>(a4.w + bias[3], p.slope)};
        const int64_t opix = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
        if (p.shuffle == 1) {
            if (p.res1) {
                float r[4];
                load4(R.r1p, p.res1_dtype, p.res1_vec, opix * p.res1_ld + co, nv, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += gate[j] * r[j];
            }
            if (p.res2) {
                float r[4];
                load4(p.res2, p.res2_dtype, p.res2_vec, opix * p.res2_ld + co, nv, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += r[j];
            }
            const int64_t oo = opix * p.out_ld + co;
            if (p.out_vec && nv == 4) {
                if (p.out_dtype == DSVC_F32) {
                    *(float4*)((float*)R.outp + oo) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint2 u;
                    u.x = pk_h16(v[0], v[1]);
                    u.y = pk_h16(v[2], v[3]);
                    *(uint2*)((unsigned short*)R.outp + oo) = u;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nv) store_elem(R.outp, p.out_dtype, oo + j, v[j]);
            }
        } else {  // PixelShuffle(2): conv channel co+j = cq*4 + dy*2 + dx  (co % 4 == 0, Cout % 4 == 0)
            const int cq = co >> 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t spix = ((int64_t)b * (2 * p.Ho) + 2 * oy + (j >> 1)) * (2 * p.Wo) + 2 * ox + (j & 1);
                float o = v[j];
                if (p.res1) o += load_elem(R.r1p, p.res1_dtype, spix * p.res1_ld + cq);
                store_elem(R.outp, p.out_dtype, spix * p.out_ld + cq, o);
            }
        }
    }
}


// ---- straight-line vector epilogue: NPL pixels of a wave-private slab, 4 channels per lane -------------------------
// ALLF32: output and residuals are known to be fp32 -- the loads / stores of the other element type are not even issued (a masked
// buffer instruction still costs its issue slot: ~150-250 cycles each beside the other workgroup's MFMA stream, tools/phase_probe_fp32.py)
template <int ACT, bool HAS_RES, int GPP, int EPLD, int NPL, bool ALLF32 = false>
__device__ __forceinline__ void epilogue_vec(const ConvParams& p, const EpiRsrc& R, const float* slab, int lane, int b,
                                             int oy_base, int ox0, int co0, int cpass) {
    const int g = lane % GPP;
    const int co = co0 + g * 4;
    const bool lane_ok = g * 4 < cpass && co < p.Cout;     // Cout % 4 == 0 on this path
    float bias[4] = {0.f, 0.f, 0.f, 0.f}, gate[4] = {1.f, 1.f, 1.f, 1.f};
    if (lane_ok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (R.bias) bias[j] = R.bias[b * R.bias_bs + co + j];
            if (p.res1_scale) gate[j] = p.res1_scale[(int64_t)b * p.Cout + co + j];
        }
    }
    // Address math: the pixel row / column start of iteration `it` is wave-uniform (scalar unit); each lane only adds
    // its constant (lane / GPP) * ld + co.  No per-lane multiplies inside the loop.
    constexpr int STEP = 64 / GPP;
    const int lp = lane / GPP;
    const unsigned lane_o = (unsigned)lp * p.out_ld + co, lane_1 = (unsigned)lp * p.res1_ld + co, lane_2 = (unsigned)lp * p.res2_ld + co;
#pragma unroll 2
    for (int it = 0; it < NPL / STEP; ++it) {   // 2 iterations in flight overlap the residual loads at modest register cost
        const int oy = oy_base + ((it * STEP) >> 5), xs = ox0 + ((it * STEP) & 31);   // scalars
        const bool ok = lane_ok && oy < p.Ho && xs + lp < p.Wo;
        const unsigned spix = ((unsigned)b * p.Ho + oy) * p.Wo + xs;                    // scalar
        const float4 a4 = *(const float4*)(slab + (lp + it * STEP) * EPLD + g * 4);
        float v[4] = {act_t<ACT>(a4.x + bias[0], p.slope), act_t<ACT>(a4.y + bias[1], p.slope),
                      act_t<ACT>(a4.z + bias[2], p.slope), act_t<ACT>(a4.w + bias[3], p.slope)};
        if constexpr (HAS_RES) {
            const unsigned e1 = spix * p.res1_ld + lane_1, e2 = spix * p.res2_ld + lane_2;
            const v4u_t f1 = __builtin_amdgcn_raw_buffer_load_b128(R.r1, (ok && R.r1_f32) ? e1 * 4 : OOB, 0, 0);
            const v4u_t f2 = __builtin_amdgcn_raw_buffer_load_b128(R.r2, (ok && R.r2_f32) ? e2 * 4 : OOB, 0, 0);
            v2u_t h1 = {0u, 0u}, h2 = {0u, 0u};
            if constexpr (!ALLF32) {
                h1 = __builtin_amdgcn_raw_buffer_load_b64(R.r1, (ok && R.r1_h16) ? e1 * 2 : OOB, 0, 0);
                h2 = __builtin_amdgcn_raw_buffer_load_b64(R.r2, (ok && R.r2_h16) ? e2 * 2 : OOB, 0, 0);
            }
            // NB: lanes are copied to scalars first -- __builtin_bit_cast applied directly to an ext-vector element
            // (f1.y) is compiled as a cast of element 0 by hipcc 7.2.
            const unsigned a1[4] = {f1.x, f1.y, f1.z, f1.w}, a2[4] = {f2.x, f2.y, f2.z, f2.w};
            const unsigned h1x = h1.x, h1y = h1.y, h2x = h2.x, h2y = h2.y;
            const float b1[4] = {h16_lo(h1x), h16_hi(h1x), h16_lo(h1y), h16_hi(h1y)};
            const float b2[4] = {h16_lo(h2x), h16_hi(h2x), h16_lo(h2y), h16_hi(h2y)};
            float r1[4], r2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r1[j] = __builtin_bit_cast(float, a1[j]) + b1[j];
                r2[j] = __builtin_bit_cast(float, a2[j]) + b2[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += gate[j] * r1[j] + r2[j];
        }
        const unsigned eo = spix * p.out_ld + lane_o;
        v4u_t of;
        of.x = __builtin_bit_cast(unsigned, v[0]); of.y = __builtin_bit_cast(unsigned, v[1]);
        of.z = __builtin_bit_cast(unsigned, v[2]); of.w = __builtin_bit_cast(unsigned, v[3]);
        v2u_t ob;
        ob.x = pk_h16(v[0], v[1]);
        ob.y = pk_h16(v[2], v[3]);
        __builtin_amdgcn_raw_buffer_store_b128(of, R.out, (ok && R.out_f32) ? eo * 4 : OOB, 0, 0);
        if constexpr (!ALLF32) __builtin_amdgcn_raw_buffer_store_b64(ob, R.out, (ok && !R.out_f32) ? eo * 2 : OOB, 0, 0);
    }
}

// ---- the same for PixelShuffle(2) outputs (subpel convs: image_model.py:25-27, modules.py:127-131) -------------------------------
// conv channel co = 4*cq + 2*dy + dx lands on output pixel (2*oy + dy, 2*ox + dx), channel cq.  Lane g of a pixel's GPP lanes takes the
// sub-pixel s = g & 3 and the FOUR conv channels 16*(g >> 2) + 4*j + s (j = 0..3) of the staged pass: four consecutive output channels of
// one output pixel, i.e. one 8-byte (f16) / 16-byte (fp32) vector instead of four scattered elements (the element-wise path ran the
// 1088x1920 subpel convs at 0.05-0.07 of the MFMA peak: tools/conv_census.py).  Slab reads are 4 x ds_read_b32, conflict-free at the
// EPLD = 68 pitch.  Host guarantees: Cout % 16 == 0, no gate / second residual, vector-aligned output and residual.
template <int ACT, bool HAS_RES, int GPP, int EPLD, int NPL>
__device__ __forceinline__ void epilogue_vec_shuffle(const ConvParams& p, const EpiRsrc& R, const float* slab, int lane, int b,
                                                     int oy_base, int ox0, int co0, int cpass) {
    const int g = lane % GPP, sub = g & 3, cqg = g >> 2;
    const int cl = 16 * cqg + sub;                          // first conv channel of this lane inside the pass
    const bool lane_ok = 16 * cqg < cpass && co0 + 16 * cqg < p.Cout;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (lane_ok && R.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = R.bias[b * R.bias_bs + co0 + cl + 4 * j];
    }
    constexpr int STEP = 64 / GPP;
    const int lp = lane / GPP;
    const int Wo2 = 2 * p.Wo;
    const unsigned lpix = (unsigned)((sub >> 1) * Wo2 + 2 * lp + (sub & 1));      // this lane's output pixel relative to the iteration's
    const unsigned cq = (unsigned)((co0 >> 2) + 4 * cqg);
    const unsigned lane_o = lpix * p.out_ld + cq, lane_1 = lpix * p.res1_ld + cq;
#pragma unroll 2
    for (int it = 0; it < NPL / STEP; ++it) {
        const int oy = oy_base + ((it * STEP) >> 5), xs = ox0 + ((it * STEP) & 31);   // scalars
        const bool ok = lane_ok && oy < p.Ho && xs + lp < p.Wo;
        const unsigned spix = ((unsigned)b * (2 * p.Ho) + 2 * oy) * Wo2 + 2 * xs;       // scalar
        const float* sl = slab + (lp + it * STEP) * EPLD + cl;
        float v[4] = {act_t<ACT>(sl[0] + bias[0], p.slope), act_t<ACT>(sl[4] + bias[1], p.slope),
                      act_t<ACT>(sl[8] + bias[2], p.slope), act_t<ACT>(sl[12] + bias[3], p.slope)};
        if constexpr (HAS_RES) {
            const unsigned e1 = spix * p.res1_ld + lane_1;
            const v4u_t f1 = __builtin_amdgcn_raw_buffer_load_b128(R.r1, (ok && R.r1_f32) ? e1 * 4 : OOB, 0, 0);
            const v2u_t h1 = __builtin_amdgcn_raw_buffer_load_b64(R.r1, (ok && R.r1_h16) ? e1 * 2 : OOB, 0, 0);
            const unsigned a1[4] = {f1.x, f1.y, f1.z, f1.w};
            const unsigned h1x = h1.x, h1y = h1.y;
            const float b1[4] = {h16_lo(h1x), h16_hi(h1x), h16_lo(h1y), h16_hi(h1y)};
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += __builtin_bit_cast(float, a1[j]) + b1[j];
        }
        const unsigned eo = spix * p.out_ld + lane_o;
        v4u_t of;
        of.x = __builtin_bit_cast(unsigned, v[0]); of.y = __builtin_bit_cast(unsigned, v[1]);
        of.z = __builtin_bit_cast(unsigned, v[2]); of.w = __builtin_bit_cast(unsigned, v[3]);
        v2u_t ob;
        ob.x = pk_h16(v[0], v[1]);
        ob.y = pk_h16(v[2], v[3]);
        __builtin_amdgcn_raw_buffer_store_b128(of, R.out, (ok && R.out_f32) ? eo * 4 : OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(ob, R.out, (ok && !R.out_f32) ? eo * 2 : OOB, 0, 0);
    }
}

// one epilogue pass: vector path (straight-line, buffer ops) when the host proved alignment, else the general path.
// F32K: the caller is an fp32-compute kernel -- only those instantiate the all-fp32 variants; SHUF: the caller can be launched with
// PixelShuffle(2) (the generic kernel only).  Every variant is inlined into its caller, and the slab-epilogue kernels sit at the
// register limit: offering the fp32-only variants to the f16 kernels as well made conv3x3_ws spill and pushed the generic kernel's
// accumulators to scratch (build.py now refuses a library with scratch use).
template <int GPP, int EPLD, int NPL, bool F32K = false, bool SHUF = false>
__device__ __forceinline__ void run_epilogue(const ConvParams& p, const EpiRsrc& R, const float* slab, int lane, int b,
                                             int oy_base, int ox0, int co0, int cpass) {
    if constexpr (SHUF) {
        if (p.epi_vec && p.shuffle == 2) {      // subpel output (host: only with activation none / ReLU / LeakyReLU)
            const bool res = p.res1 != nullptr;
            if (p.act == DSVC_ACT_RELU) {
                if (res) epilogue_vec_shuffle<DSVC_ACT_RELU, true, GPP, EPLD, NPL>(p, R, slab, lane, b, oy_base, ox0, co0, cpass);
                else epilogue_vec_shuffle<DSVC_ACT_RELU, false, GPP, EPLD, NPL>(p, R, slab, lane, b, oy_base, ox0, co0, cpass);
            } else if (p.act == DSVC_ACT_LRELU) {
                if (res) epilogue_vec_shuffle<DSVC_ACT_LRELU, true, GPP, EPLD, NPL>(p, R, slab, lane, b, oy_base, ox0, co0, cpass);
                else epilogue_vec_shuffle<DSVC_ACT_LRELU, false, GPP, EPLD, NPL>(p, R, slab, lane, b, oy_base, ox0, co0, cpass);
            } else {
                if (res) epilogue_vec_shuffle<DSVC_ACT_NONE, true, GPP, EPLD, NPL>(p, R, slab, lane, b, oy_base, ox0, co0, cpass);
                else epilogue_vec_shuffle<DSVC_ACT_NONE, false, GPP, EPLD, NPL>(p, R, slab, lane, b, oy_base, ox0, co0, cpass);
            }
            return;
        }
    }
    const bool vec = p.epi_vec && (SHUF || p.shuffle == 1);
    const bool res = p.res1 || p.res2;
#define DSVC_EPI_CASE(A)                                                                                            \
    case A:                                                                                                         \
        if (!vec) epilogue_pass<A, GPP, EPLD>(p, R, slab, lane, NPL, b, oy_base, ox0, co0, cpass);                  \
        else {                                                                                                      \
            bool done = false;                                                                                      \
            if constexpr (F32K) {                                                                                   \
                if (p.epi_f32) {                                                                                    \
                    if (res) epilogue_vec<A, true, GPP, EPLD, NPL, true>(p, R, slab, lane, b, oy_base, ox0, co0, cpass); \
                    else epilogue_vec<A, false, GPP, EPLD, NPL, true>(p, R, slab, lane, b, oy_base, ox0, co0, cpass);   \
                    done = true;                                                                                    \
                }                                                                                                   \
            }                                                                                                       \
            if (!done) {                                                                                            \
                if (res) epilogue_vec<A, true, GPP, EPLD, NPL>(p, R, slab, lane, b, oy_base, ox0, co0, cpass);      \
                else epilogue_vec<A, false, GPP, EPLD, NPL>(p, R, slab, lane, b, oy_base, ox0, co0, cpass);         \
            }                                                                                                       \
        }                                                                                                           \
        break;
    switch (p.act) {
        DSVC_EPI_CASE(DSVC_ACT_RELU)
        DSVC_EPI_CASE(DSVC_ACT_LRELU)
        DSVC_EPI_CASE(DSVC_ACT_GELU)
        DSVC_EPI_CASE(DSVC_ACT_SIGMOID)
        DSVC_EPI_CASE(DSVC_ACT_HALF_TANH)
        default:
        DSVC_EPI_CASE(DSVC_ACT_NONE)
    }
#undef DSVC_EPI_CASE
}


// ---- direct epilogue for the SWAPPED operand order (A = weights, B = pixels) -------------------------------------------
// D[row = output channel][col = pixel]: lane (r = lane&31, h = lane>>5) owns pixel r of the M-tile and, per register
// group g = reg>>2, the 4 consecutive channels cbase + 8g + 4h .. +3 -- exactly one 16-B (fp32) / 8-B (f16) vector
// of the NHWC output.  No LDS transposition, no barrier; straight-line buffer ops (masked lanes / absent operands via OOB).
template <int ACT, bool HAS_RES>
__device__ __forceinline__ void epilogue_direct_tile(const ConvParams& p, const EpiRsrc& R, const rsrc_t& rbias, const f32x16& acc,
                                                     unsigned opix, bool pix_ok, int cbase, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int co = cbase + 8 * g + 4 * h;
        const bool ok = pix_ok && co < p.Cout;              // Cout % 4 == 0 on this path
        const v4u_t bw = __builtin_amdgcn_raw_buffer_load_b128(rbias, co < p.Cout ? (unsigned)co * 4 : OOB, 0, 0);
        const unsigned bu[4] = {bw.x, bw.y, bw.z, bw.w};
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_t<ACT>(acc[4 * g + j] + __builtin_bit_cast(float, bu[j]), p.slope);
        if constexpr (HAS_RES) {
            const unsigned e1 = opix * p.res1_ld + co, e2 = opix * p.res2_ld + co;
            const v4u_t f1 = __builtin_amdgcn_raw_buffer_load_b128(R.r1, (ok && R.r1_f32) ? e1 * 4 : OOB, 0, 0);
            const v2u_t h1 = __builtin_amdgcn_raw_buffer_load_b64(R.r1, (ok && R.r1_h16) ? e1 * 2 : OOB, 0, 0);
            const v4u_t f2 = __builtin_amdgcn_raw_buffer_load_b128(R.r2, (ok && R.r2_f32) ? e2 * 4 : OOB, 0, 0);
            const v2u_t h2 = __builtin_amdgcn_raw_buffer_load_b64(R.r2, (ok && R.r2_h16) ? e2 * 2 : OOB, 0, 0);
            const unsigned a1[4] = {f1.x, f1.y, f1.z, f1.w}, a2[4] = {f2.x, f2.y, f2.z, f2.w};
            const unsigned h1x = h1.x, h1y = h1.y, h2x = h2.x, h2y = h2.y;
            const float b1[4] = {h16_lo(h1x), h16_hi(h1x), h16_lo(h1y), h16_hi(h1y)};
            const float b2[4] = {h16_lo(h2x), h16_hi(h2x), h16_lo(h2y), h16_hi(h2y)};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] += (__builtin_bit_cast(float, a1[j]) + b1[j]) + (__builtin_bit_cast(float, a2[j]) + b2[j]);
        }
        const unsigned eo = opix * p.out_ld + co;
        v4u_t of;
        of.x = __builtin_bit_cast(unsigned, v[0]); of.y = __builtin_bit_cast(unsigned, v[1]);
        of.z = __builtin_bit_cast(unsigned, v[2]); of.w = __builtin_bit_cast(unsigned, v[3]);
        v2u_t ob;
        ob.x = pk_h16(v[0], v[1]);
        ob.y = pk_h16(v[2], v[3]);
        __builtin_amdgcn_raw_buffer_store_b128(of, R.out, (ok && R.out_f32) ? eo * 4 : OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(ob, R.out, (ok && !R.out_f32) ? eo * 2 : OOB, 0, 0);
    }
}

__device__ __forceinline__ void run_epilogue_direct(const ConvParams& p, const EpiRsrc& R, const rsrc_t& rbias, const f32x16& acc,
                                                    unsigned opix, bool pix_ok, int cbase, int h) {
#define DSVC_EPD_CASE(A)                                                                                         \
    case A:                                                                                                      \
        if (p.res1 || p.res2) epilogue_direct_tile<A, true>(p, R, rbias, acc, opix, pix_ok, cbase, h);            \
        else epilogue_direct_tile<A, false>(p, R, rbias, acc, opix, pix_ok, cbase, h);                            \
        break;
    switch (p.act) {
        DSVC_EPD_CASE(DSVC_ACT_RELU)
        DSVC_EPD_CASE(DSVC_ACT_LRELU)
        DSVC_EPD_CASE(DSVC_ACT_GELU)
        DSVC_EPD_CASE(DSVC_ACT_SIGMOID)
        DSVC_EPD_CASE(DSVC_ACT_HALF_TANH)
        default:
            if (p.res1 || p.res2) epilogue_direct_tile<DSVC_ACT_NONE, true>(p, R, rbias, acc, opix, pix_ok, cbase, h);
            else epilogue_direct_tile<DSVC_ACT_NONE, false>(p, R, rbias, acc, opix, pix_ok, cbase, h);
            break;
    }
#undef DSVC_EPD_CASE
}

template <typename CT> struct CTraits;
template <> struct CTraits<float> { static constexpr int KCE = 16, EPV = 4; };
template <> struct CTraits<h16> { static constexpr int KCE = 32, EPV = 8; };

template <int EPV>
__device__ __forceinline__ void load_vec(const ConvSeg& sg, int64_t off, int c, float (&v)[EPV]) {
    const int nvalid = sg.C - c;
    if (nvalid >= EPV && sg.vec_ok) {
        if (sg.dtype == DSVC_F32) {
            const float4* p = (const float4*)((const float*)sg.ptr + off + c);
#pragma unroll
            for (int j = 0; j < EPV / 4; ++j) {
                float4 t = p[j];
                v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
            }
        } else {
            const unsigned short* p = (const unsigned short*)sg.ptr + off + c;
            if (EPV == 8) {
                uint4 t = *(const uint4*)p;
                unsigned u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[2 * j] = h16_lo(u[j]);
                    v[2 * j + 1] = h16_hi(u[j]);
                }
            } else {
                uint2 t = *(const uint2*)p;
                unsigned u[2] = {t.x, t.y};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    v[2 * j] = h16_lo(u[j]);
                    v[2 * j + 1] = h16_hi(u[j]);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < EPV; ++j) v[j] = (j < nvalid) ? load_elem(sg.ptr, sg.dtype, off + c + j) : 0.f;
    }
}

template <typename CT, int EPV>
__device__ __forceinline__ uint4 pack_vec(const float (&v)[EPV]) {
    if constexpr (sizeof(CT) == 4) {
        return make_uint4(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                          __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]));
    } else {
        unsigned u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            u[j] = pk_h16(v[2 * j], v[2 * j + 1]);
        return make_uint4(u[0], u[1], u[2], u[3]);
    }
}

}  // namespace dsvc_conv

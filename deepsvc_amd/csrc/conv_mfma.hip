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
//   K is walked in 64-byte "chunks" of input channels (16 fp32 / 32 bf16).  For one chunk the
//   input HALO PATCH ((TH-1)*s+kh) x ((TW-1)*s+kw) pixels is staged ONCE into LDS and every tap
//   reads it at a shifted pixel offset -- HBM/L2 sees each input element once per tile, not kh*kw times.
//   LDS rows (one pixel of the patch, or one output channel of a tap's weights) are 64 B of data
//   + 16 B pad = 80 B: consecutive rows advance 20 banks, which makes the 16-lane groups of
//   ds_read_b128 conflict-free (MI355X_MICROARCH.md, LDS).  A lane reads 16 B: lane (r=l&31,h=l>>5)
//   takes bytes [32h + 16ks, +16) of row r.  The same byte->k assignment is used for A and B, so
//   the K order inside a chunk is a free permutation.
//   fp32 compute: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain), 4 MFMAs per 16-B fragment pair.
//   bf16 compute: v_mfma_f32_32x32x16_bf16, 1 MFMA per 16-B fragment pair, fp32 accumulate.
//   Epilogue (fused): + bias, activation, + residual (optionally gated per channel: SELayer),
//   + second residual, optional PixelShuffle(2) scatter, cast to the output storage type.
#include "common.h"

namespace {

constexpr int ROWB = 80;   // LDS row pitch in bytes
constexpr int TW = 32;     // tile width in output pixels == MFMA M

struct ConvSeg {
    const void* ptr;
    int C, ld, dtype, vec_ok, chunk0;
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
    const void* res1; int res1_ld, res1_dtype; const float* res1_scale;
    const void* res2; int res2_ld, res2_dtype;
    void* out; int out_ld, out_dtype, shuffle;
    int tiles_x, tiles_y, tapg;
};

template <typename CT> struct CTraits;
template <> struct CTraits<float> { static constexpr int KCE = 16, EPV = 4; };
template <> struct CTraits<__bf16> { static constexpr int KCE = 32, EPV = 8; };

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
                    v[2 * j] = __builtin_bit_cast(float, u[j] << 16);
                    v[2 * j + 1] = __builtin_bit_cast(float, u[j] & 0xFFFF0000u);
                }
            } else {
                uint2 t = *(const uint2*)p;
                unsigned u[2] = {t.x, t.y};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    v[2 * j] = __builtin_bit_cast(float, u[j] << 16);
                    v[2 * j + 1] = __builtin_bit_cast(float, u[j] & 0xFFFF0000u);
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
            u[j] = (unsigned)f32_to_bf16_bits(v[2 * j]) | ((unsigned)f32_to_bf16_bits(v[2 * j + 1]) << 16);
        return make_uint4(u[0], u[1], u[2], u[3]);
    }
}

template <typename CT, int MT, int NT>
__global__ void __launch_bounds__(256, 2) conv_mfma_kernel(const ConvParams p) {
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

    const int oy0 = ty * TH, ox0 = tx * TW;
    const int PH = (TH - 1) * p.stride + p.kh, PW = (TW - 1) * p.stride + p.kw;
    const int iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
    const int npatch = PH * PW;
    char* lds_a = smem;
    char* lds_w = smem + (size_t)npatch * ROWB;
    const int ntaps = p.kh * p.kw;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    const int frag_off = h * 32;  // byte offset of this lane's K half inside a 64-B chunk row

    for (int ch = 0; ch < p.nchunks; ++ch) {
        int s = 0;
        if (p.nseg > 1 && ch >= p.seg[1].chunk0) s = 1;
        if (p.nseg > 2 && ch >= p.seg[2].chunk0) s = 2;
        const ConvSeg& sg = p.seg[s];
        const int c0 = (ch - sg.chunk0) * KCE;

        __syncthreads();  // all waves done with the previous chunk's LDS contents
        // ---- stage the input halo patch of this channel chunk ----
        for (int v = tid; v < npatch * 4; v += 256) {
            const int pix = v >> 2, q = v & 3;
            const int py = pix / PW, px = pix - py * PW;
            const int iy = iy0 + py, ix = ix0 + px;
            float vals[EPV];
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                const int64_t off = (((int64_t)b * p.H + iy) * p.W + ix) * sg.ld;
                load_vec<EPV>(sg, off, c0 + q * EPV, vals);
                if (p.pre_act != DSVC_ACT_NONE) {
#pragma unroll
                    for (int j = 0; j < EPV; ++j) vals[j] = apply_act(vals[j], p.pre_act, p.slope);
                }
            } else {
#pragma unroll
                for (int j = 0; j < EPV; ++j) vals[j] = 0.f;
            }
            *(uint4*)(lds_a + pix * ROWB + q * 16) = pack_vec<CT, EPV>(vals);
        }

        for (int tg0 = 0; tg0 < ntaps; tg0 += p.tapg) {
            const int tcur = min(p.tapg, ntaps - tg0);
            if (tg0 > 0) __syncthreads();
            // ---- stage the weights of taps [tg0, tg0+tcur) for this chunk and N group ----
            const char* wbase = (const char*)p.w + ((size_t)ch * ntaps + tg0) * p.Np * 64;
            for (int v = tid; v < tcur * NTN * 4; v += 256) {
                const int row = v >> 2, q = v & 3;
                const int t = row / NTN, n = row - t * NTN;
                const int ngl = ng * NTN + n;
                uint4 val = make_uint4(0, 0, 0, 0);
                if (ngl < p.Np) val = *(const uint4*)(wbase + ((size_t)t * p.Np + ngl) * 64 + q * 16);
                *(uint4*)(lds_w + row * ROWB + q * 16) = val;
            }
            __syncthreads();

            for (int t = 0; t < tcur; ++t) {
                const int tap = tg0 + t;
                const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    uint4 af[MT], bf[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        bf[n] = *(const uint4*)(lds_w + (t * NTN + n * 32 + r) * ROWB + frag_off + ks * 16);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int prow = (wave * MT + m) * p.stride + ky;
                        af[m] = *(const uint4*)(lds_a + (prow * PW + r * p.stride + kx) * ROWB + frag_off + ks * 16);
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            if constexpr (sizeof(CT) == 4) {
                                const f32x4 a4 = __builtin_bit_cast(f32x4, af[m]);
                                const f32x4 b4 = __builtin_bit_cast(f32x4, bf[n]);
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b4[j], acc[m][n], 0, 0, 0);
                            } else {
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                    __builtin_bit_cast(bf16x8, af[m]), __builtin_bit_cast(bf16x8, bf[n]), acc[m][n], 0, 0, 0);
                            }
                        }
                }
            }
        }
    }

    // ---- epilogue ----
    // MFMA C/D layout: col = lane&31 (output channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel).
    // Accumulators go through a wave-private LDS slab [pixel][channel] (<= 2 N-tiles per pass) so the
    // fused bias/activation/residual/store code exists once, in a rolled loop, and global traffic is
    // 16-byte vectors along the channel axis (NHWC contiguous).
    constexpr int EPN = NT >= 2 ? 2 : 1;
    constexpr int EPC = EPN * 32, EPLD = EPC + 4, GPP = EPC / 4;
    float* slab = (float*)smem + (size_t)wave * (MT * 32 * EPLD);
#pragma unroll
    for (int n0 = 0; n0 < NT; n0 += EPN) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPN; ++e) {
            if (n0 + e < NT) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        slab[(m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * EPLD + e * 32 + r] = acc[m][n0 + e < NT ? n0 + e : 0][i];
            }
        }
        __syncthreads();
        const int cpass = (NT - n0 < EPN ? NT - n0 : EPN) * 32;  // channels staged this pass
        for (int it = lane; it < MT * 32 * GPP; it += 64) {
            const int pl = it / GPP, g = it - pl * GPP;
            const int m = pl >> 5, row = pl & 31;
            const int oy = oy0 + wave * MT + m, ox = ox0 + row;
            const int co = ng * NTN + n0 * 32 + g * 4;
            if (g * 4 >= cpass || oy >= p.Ho || ox >= p.Wo || co >= p.Cout) continue;
            const int nv = min(4, p.Cout - co);
            const float4 a4 = *(const float4*)(slab + pl * EPLD + g * 4);
            float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bias = (p.bias && j < nv) ? p.bias[co + j] : 0.f;
                v[j] = apply_act(v[j] + bias, p.act, p.slope);
            }
            const int64_t opix = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
            if (p.shuffle == 1) {
                if (p.res1) {
                    const int64_t ro = opix * p.res1_ld + co;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < nv) {
                            const float gate = p.res1_scale ? p.res1_scale[(int64_t)b * p.Cout + co + j] : 1.f;
                            v[j] += gate * load_elem(p.res1, p.res1_dtype, ro + j);
                        }
                }
                if (p.res2) {
                    const int64_t ro = opix * p.res2_ld + co;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < nv) v[j] += load_elem(p.res2, p.res2_dtype, ro + j);
                }
                const int64_t oo = opix * p.out_ld + co;
                if (nv == 4 && p.out_dtype == DSVC_F32 && ((((uintptr_t)p.out >> 2) + oo) & 3) == 0) {
                    *(float4*)((float*)p.out + oo) = make_float4(v[0], v[1], v[2], v[3]);
                } else if (nv == 4 && p.out_dtype == DSVC_BF16 && ((((uintptr_t)p.out >> 1) + oo) & 3) == 0) {
                    uint2 u;
                    u.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
                    u.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
                    *(uint2*)((unsigned short*)p.out + oo) = u;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < nv) store_elem(p.out, p.out_dtype, oo + j, v[j]);
                }
            } else {  // PixelShuffle(2): conv channel co+j = cq*4 + dy*2 + dx  (co % 4 == 0, Cout % 4 == 0)
                const int cq = co >> 2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int dy = j >> 1, dx = j & 1;
                    const int64_t spix = ((int64_t)b * (2 * p.Ho) + 2 * oy + dy) * (2 * p.Wo) + 2 * ox + dx;
                    float o = v[j];
                    if (p.res1) o += load_elem(p.res1, p.res1_dtype, spix * p.res1_ld + cq);
                    store_elem(p.out, p.out_dtype, spix * p.out_ld + cq, o);
                }
            }
        }
    }
}

template <typename CT, int MT, int NT>
int launch_cfg(hipStream_t st, const ConvParams& p, int ngroups, size_t lds) {
    static bool attr_set = false;
    auto kern = conv_mfma_kernel<CT, MT, NT>;
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

template <typename CT, int MT>
int launch_nt(hipStream_t st, const ConvParams& p, int NT, int ngroups, size_t lds) {
    switch (NT) {
        case 1: return launch_cfg<CT, MT, 1>(st, p, ngroups, lds);
        case 2: return launch_cfg<CT, MT, 2>(st, p, ngroups, lds);
        case 3: return launch_cfg<CT, MT, 3>(st, p, ngroups, lds);
        default: return launch_cfg<CT, MT, 4>(st, p, ngroups, lds);
    }
}

inline int chunk_elems(int compute) { return compute == DSVC_F32 ? 16 : 32; }

}  // namespace

extern "C" int64_t dsvc_conv_packed_bytes(int32_t Cout, int32_t nseg, const int32_t* segC, int32_t kh, int32_t kw, int32_t compute) {
    if (Cout <= 0 || nseg < 1 || nseg > 3 || !segC || kh < 1 || kw < 1) return DSVC_E_ARG;
    const int kce = chunk_elems(compute);
    int64_t nch = 0;
    for (int s = 0; s < nseg; ++s) nch += (segC[s] + kce - 1) / kce;
    const int64_t Np = (Cout + 31) / 32 * 32;
    return nch * kh * kw * Np * 64;
}

extern "C" int dsvc_conv2d(void* stream, const dsvc_conv_desc* d) {
    DSVC_REQUIRE(d != nullptr, "conv2d: null descriptor");
    DSVC_REQUIRE(d->nseg >= 1 && d->nseg <= 3, "conv2d: nseg=%d out of 1..3", d->nseg);
    DSVC_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "conv2d: bad shape B=%d H=%d W=%d Cout=%d", d->B, d->H, d->W, d->Cout);
    DSVC_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->kh <= 7 && d->kw <= 7, "conv2d: kernel %dx%d unsupported", d->kh, d->kw);
    DSVC_REQUIRE(d->stride == 1 || d->stride == 2, "conv2d: stride %d unsupported", d->stride);
    DSVC_REQUIRE(d->compute == DSVC_F32 || d->compute == DSVC_BF16, "conv2d: bad compute type %d", d->compute);
    DSVC_REQUIRE(d->weight && d->out, "conv2d: null weight/out");
    DSVC_REQUIRE(d->shuffle == 1 || (d->shuffle == 2 && d->Cout % 4 == 0), "conv2d: shuffle=%d Cout=%d", d->shuffle, d->Cout);
    DSVC_REQUIRE(!(d->shuffle == 2 && (d->res2 || d->res1_scale)), "conv2d: shuffle with res2/gate unsupported");
    DSVC_REQUIRE(((uintptr_t)d->weight & 15) == 0, "conv2d: packed weights must be 16-byte aligned");

    ConvParams p{};
    const int kce = chunk_elems(d->compute);
    int chunk0 = 0;
    for (int s = 0; s < d->nseg; ++s) {
        const dsvc_seg& in = d->in[s];
        DSVC_REQUIRE(in.ptr && in.C > 0 && in.ld >= in.C, "conv2d: segment %d ptr/C/ld invalid (C=%d ld=%d)", s, in.C, in.ld);
        DSVC_REQUIRE(in.dtype == DSVC_F32 || in.dtype == DSVC_BF16, "conv2d: segment %d dtype", s);
        const int es = dsvc_esize(in.dtype);
        p.seg[s] = ConvSeg{in.ptr, in.C, in.ld, in.dtype,
                           (int)((((uintptr_t)in.ptr & 15) == 0) && (((int64_t)in.ld * es) % 16 == 0)), chunk0};
        chunk0 += (in.C + kce - 1) / kce;
    }
    p.nseg = d->nseg; p.nchunks = chunk0;
    p.B = d->B; p.H = d->H; p.W = d->W;
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
    p.Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    p.Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    DSVC_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv2d: empty output");
    p.Cout = d->Cout; p.Np = (d->Cout + 31) / 32 * 32;
    p.w = d->weight; p.bias = d->bias;
    p.pre_act = d->pre_act; p.act = d->act; p.slope = d->slope;
    p.res1 = d->res1; p.res1_ld = d->res1_ld; p.res1_dtype = d->res1_dtype; p.res1_scale = d->res1_scale;
    p.res2 = d->res2; p.res2_ld = d->res2_ld; p.res2_dtype = d->res2_dtype;
    p.out = d->out; p.out_ld = d->out_ld; p.out_dtype = d->out_dtype; p.shuffle = d->shuffle;

    // ---- tile configuration ----
    const int ntiles32 = p.Np / 32;
    int NT = ntiles32 >= 4 ? 4 : ntiles32;           // 1,2,3,4 (96 ch -> 3)
    int MT = (d->stride == 2) ? 1 : 2;
    auto nblocks = [&](int mt, int nt) {
        return (int64_t)p.B * ((p.Ho + 4 * mt - 1) / (4 * mt)) * ((p.Wo + TW - 1) / TW) * ((ntiles32 + nt - 1) / nt);
    };
    // small problems (entropy-parameter nets at 1/16 res): trade operand reuse for more workgroups
    if (nblocks(MT, NT) < 512 && MT == 2) MT = 1;
    while (nblocks(MT, NT) < 512 && NT > 1) NT = (NT + 1) / 2;
    const int TH = 4 * MT;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    const int ngroups = (ntiles32 + NT - 1) / NT;
    const int PH = (TH - 1) * d->stride + d->kh, PW = (TW - 1) * d->stride + d->kw;
    const size_t lds_a = (size_t)PH * PW * ROWB;
    const int ntaps = d->kh * d->kw;
    const size_t budget = 78 * 1024;  // two workgroups per CU
    p.tapg = ntaps;
    if (lds_a + (size_t)p.tapg * NT * 32 * ROWB > budget) p.tapg = d->kw;
    if (lds_a + (size_t)p.tapg * NT * 32 * ROWB > 160 * 1024) p.tapg = 1;
    const size_t lds_main = lds_a + (size_t)p.tapg * NT * 32 * ROWB;
    const size_t lds_epi = (size_t)4 * MT * 32 * ((NT >= 2 ? 64 : 32) + 4) * sizeof(float);
    const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    DSVC_REQUIRE(lds <= 160 * 1024, "conv2d: LDS need %zu exceeds 160 KiB", lds);

    hipStream_t st = (hipStream_t)stream;
    if (d->compute == DSVC_F32)
        return MT == 2 ? launch_nt<float, 2>(st, p, NT, ngroups, lds) : launch_nt<float, 1>(st, p, NT, ngroups, lds);
    return MT == 2 ? launch_nt<__bf16, 2>(st, p, NT, ngroups, lds) : launch_nt<__bf16, 1>(st, p, NT, ngroups, lds);
}

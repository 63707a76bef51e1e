// pointwise.hip -- HBM-bound resampling / layout / pointwise kernels of the hot path (NHWC).
// Each kernel is a grid-stride loop over (pixel, channel-vector) with channels fastest so that a
// wave touches contiguous bytes (coalesced) -- these ops are judged against the HBM roofline.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int64_t kIdx32 = (1ll << 32) - (1ll << 24);   // item counts below this take the 32-bit index decomposition (the grid-stride step stays below 2^24)
inline unsigned grid_for(int64_t n) {
    int64_t g = (n + kThreads - 1) / kThreads;
    const int64_t cap = 256 * 16;  // 16 workgroups per CU, grid-stride beyond
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---- torch_warp (modules.py:25-62) -----------------------------------------------------------
// grid_sample(bilinear, padding_mode='border', align_corners=True) with grid = base + flow/((size-1)/2):
// the sampling position in pixels is (x + fx, y + fy), clipped to [0, size-1]; taps beyond the edge
// carry zero weight.  One thread per (pixel, 4-channel group) when C % 4 == 0, else per element.
// IT: index type of the work-item decomposition.  `unsigned` whenever the item count fits (always, up to 4K): the chains of 64-bit
// div / mod by run-time values (4-6 per item, ~100 instructions each) had made the 64-channel warp VALU-bound at 183 us where its bytes
// move in ~100 us (rocprofv3, round 3); int64_t remains for larger problems.
template <int VEC, typename IT>
__global__ void warp_kernel(const void* in, int in_ld, int in_dt, const float* flow, int flow_ld,
                            void* out, int out_ld, int out_dt, int B, int H, int W, int C) {
    DSVC_F16_SAT();
    const int cg = (C + VEC - 1) / VEC;
    const IT total = (IT)B * H * W * cg;
    for (IT i = (IT)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IT)gridDim.x * blockDim.x) {
        const IT pixi = i / (IT)cg, rowi = pixi / (IT)W;
        const int g = (int)(i - pixi * cg);
        const int64_t pix = (int64_t)pixi;
        const int x = (int)(pixi - rowi * W);
        const int64_t b = (int64_t)(rowi / (IT)H);
        const int y = (int)(rowi - (IT)b * H);
        const float fx = flow[pix * flow_ld], fy = flow[pix * flow_ld + 1];
        float sx = fminf(fmaxf((float)x + fx, 0.f), (float)(W - 1));
        float sy = fminf(fmaxf((float)y + fy, 0.f), (float)(H - 1));
        const float x0f = floorf(sx), y0f = floorf(sy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float ax = sx - x0f, ay = sy - y0f;
        const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
        const float w00 = (1.f - ax) * (1.f - ay), w01 = ax * (1.f - ay), w10 = (1.f - ax) * ay, w11 = ax * ay;
        const int64_t base = b * H * W;
        const int64_t o00 = (base + (int64_t)y0 * W + x0) * in_ld, o01 = (base + (int64_t)y0 * W + x1) * in_ld;
        const int64_t o10 = (base + (int64_t)y1 * W + x0) * in_ld, o11 = (base + (int64_t)y1 * W + x1) * in_ld;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = g * VEC + j;
            if (c >= C) break;
            const float v = w00 * load_elem(in, in_dt, o00 + c) + w01 * load_elem(in, in_dt, o01 + c) +
                            w10 * load_elem(in, in_dt, o10 + c) + w11 * load_elem(in, in_dt, o11 + c);
            store_elem(out, out_dt, pix * out_ld + c, v);
        }
    }
}

// ---- F.interpolate(x2, bilinear, align_corners=False) * scale (modules.py:107-112,163) ----------
__global__ void upsample2x_kernel(const float* in, int in_ld, float* out, int out_ld, int B, int H, int W, int C, float scale) {
    const int Ho = 2 * H, Wo = 2 * W;
    const int64_t total = (int64_t)B * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int x = (int)(pix % Wo);
        const int y = (int)((pix / Wo) % Ho);
        const int64_t b = pix / ((int64_t)Wo * Ho);
        const float sx = fmaxf((x + 0.5f) * 0.5f - 0.5f, 0.f), sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f);
        const int x0 = (int)sx, y0 = (int)sy;
        const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
        const float lx = sx - x0, ly = sy - y0;
        const int64_t base = b * H * W;
        const float v00 = in[(base + (int64_t)y0 * W + x0) * in_ld + c], v01 = in[(base + (int64_t)y0 * W + x1) * in_ld + c];
        const float v10 = in[(base + (int64_t)y1 * W + x0) * in_ld + c], v11 = in[(base + (int64_t)y1 * W + x1) * in_ld + c];
        const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        out[pix * out_ld + c] = v * scale;
    }
}

__global__ void pool2x2_kernel(int mode, const void* in, int in_ld, void* out, int out_ld, int dt, int B, int H, int W, int C) {
    DSVC_F16_SAT();
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int x = (int)(pix % Wo);
        const int y = (int)((pix / Wo) % Ho);
        const int64_t b = pix / ((int64_t)Wo * Ho);
        const int64_t p00 = ((b * H + 2 * y) * W + 2 * x) * (int64_t)in_ld + c;
        const float a = load_elem(in, dt, p00), bb = load_elem(in, dt, p00 + in_ld);
        const float cc = load_elem(in, dt, p00 + (int64_t)W * in_ld), d = load_elem(in, dt, p00 + (int64_t)(W + 1) * in_ld);
        const float v = mode == 0 ? ((a + bb) + (cc + d)) * 0.25f : fmaxf(fmaxf(a, bb), fmaxf(cc, d));
        store_elem(out, dt, pix * out_ld + c, v);
    }
}

// 16 bytes per thread (8 f16 / 4 fp32 channels): the element-wise kernel above moves 2-4 bytes per lane (1.5 TB/s at 1080p)
template <bool H16, typename IT>
__global__ void pool2x2_vec_kernel(int mode, const void* in, int in_ld, void* out, int out_ld, int B, int H, int W, int C) {
    DSVC_F16_SAT();
    constexpr int V = H16 ? 8 : 4;
    const int Ho = H / 2, Wo = W / 2, CV = C / V;
    const IT total = (IT)B * Ho * Wo * CV;
    for (IT i = (IT)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IT)gridDim.x * blockDim.x) {
        const IT pixi = i / (IT)CV, rowi = pixi / (IT)Wo;
        const int c = (int)(i - pixi * CV) * V;
        const int64_t pix = (int64_t)pixi;
        const int x = (int)(pixi - rowi * Wo);
        const int64_t b = (int64_t)(rowi / (IT)Ho);
        const int y = (int)(rowi - (IT)b * Ho);
        const int64_t p00 = ((b * H + 2 * y) * W + 2 * x) * (int64_t)in_ld + c;
        const int64_t offs[4] = {p00, p00 + in_ld, p00 + (int64_t)W * in_ld, p00 + (int64_t)(W + 1) * in_ld};
        float v[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 u = H16 ? *(const uint4*)((const unsigned short*)in + offs[k]) : *(const uint4*)((const float*)in + offs[k]);
            const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (H16) {
                    v[k][2 * j] = h16_lo(w[j]);
                    v[k][2 * j + 1] = h16_hi(w[j]);
                } else {
                    v[k][j] = __builtin_bit_cast(float, w[j]);
                }
            }
        }
        float o[8];
#pragma unroll
        for (int j = 0; j < V; ++j)
            o[j] = mode == 0 ? ((v[0][j] + v[1][j]) + (v[2][j] + v[3][j])) * 0.25f : fmaxf(fmaxf(v[0][j], v[1][j]), fmaxf(v[2][j], v[3][j]));
        if constexpr (H16) {
            uint4 u;
            u.x = pk_h16(o[0], o[1]);
            u.y = pk_h16(o[2], o[3]);
            u.z = pk_h16(o[4], o[5]);
            u.w = pk_h16(o[6], o[7]);
            *(uint4*)((unsigned short*)out + pix * out_ld + c) = u;
        } else {
            *(float4*)((float*)out + pix * out_ld + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// nn.PixelShuffle(r) reading NCHW fp32, writing NHWC: out[b][y][x][c] = in[b][c*r*r + (y%r)*r + (x%r)][y/r][x/r]
__global__ void pixel_shuffle_nchw_kernel(const float* in, void* out, int out_ld, int out_dt, int B, int Co, int Hin, int Win, int r) {
    DSVC_F16_SAT();
    const int Ho = Hin * r, Wo = Win * r;
    const int64_t total = (int64_t)B * Ho * Wo * Co;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // iterate x fastest for coalesced reads of the NCHW source; writes are strided by out_ld
        const int x = (int)(i % Wo);
        const int c = (int)((i / Wo) % Co);
        const int y = (int)((i / ((int64_t)Wo * Co)) % Ho);
        const int64_t b = i / ((int64_t)Wo * Co * Ho);
        const int cin = c * r * r + (y % r) * r + (x % r);
        const float v = in[((b * (Co * r * r) + cin) * Hin + y / r) * (int64_t)Win + x / r];
        store_elem(out, out_dt, ((b * Ho + y) * Wo + x) * (int64_t)out_ld + c, v);
    }
}

// Same op, one thread per OUTPUT PIXEL and 8 channels: lanes walk the source x (64 consecutive floats of one plane per load,
// coalesced) and write their pixel's 8 channels as one 16-byte (f16) / two 16-byte (fp32) stores -- the element-wise kernel
// above wrote 2 bytes per thread at a 32-byte stride (216 us for the 1080p inter-layer feature, ~0.9 TB/s).
template <bool H16, typename IT>
__global__ void pixel_shuffle_nchw_vec8_kernel(const float* __restrict__ in, void* out, int out_ld, int B, int Co, int Hin, int Win, int r) {
    DSVC_F16_SAT();
    const IT total = (IT)B * Hin * r * r * Win * (Co / 8);
    const int64_t plane = (int64_t)Hin * Win;
    for (IT i = (IT)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IT)gridDim.x * blockDim.x) {
        IT t = i / (IT)Win, u;
        const int xi = (int)(i - t * Win);
        u = t / (IT)r; const int rx = (int)(t - u * r); t = u;
        u = t / (IT)r; const int ry = (int)(t - u * r); t = u;
        u = t / (IT)(Co / 8); const int c8 = (int)(t - u * (Co / 8)) * 8; t = u;
        u = t / (IT)Hin; const int yi = (int)(t - u * Hin);
        const int64_t b = (int64_t)u;
        const float* src = in + ((b * Co + c8) * r * r + ry * r + rx) * plane + (int64_t)yi * Win + xi;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)j * r * r * plane];
        const int64_t o = ((b * Hin * r + yi * r + ry) * (int64_t)Win * r + xi * r + rx) * out_ld + c8;
        if constexpr (H16) {
            uint4 u;
            u.x = pk_h16(v[0], v[1]);
            u.y = pk_h16(v[2], v[3]);
            u.z = pk_h16(v[4], v[5]);
            u.w = pk_h16(v[6], v[7]);
            *(uint4*)((unsigned short*)out + o) = u;
        } else {
            *(float4*)((float*)out + o) = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)((float*)out + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// layout transposes through an LDS tile so both sides are coalesced
__global__ void nchw_to_nhwc_kernel(const float* in, void* out, int out_ld, int out_dt, int B, int C, int64_t HW) {
    DSVC_F16_SAT();
    __shared__ float tile[32][33];
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int64_t b = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k;
        const int64_t pp = p0 + tx;
        tile[k][tx] = (c < C && pp < HW) ? in[(b * C + c) * HW + pp] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t pp = p0 + k;
        const int c = c0 + tx;
        if (c < C && pp < HW) store_elem(out, out_dt, (b * HW + pp) * out_ld + c, tile[tx][k]);
    }
}

__global__ void nhwc_to_nchw_kernel(const void* in, int in_ld, int in_dt, float* out, int B, int C, int64_t HW) {
    __shared__ float tile[32][33];
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int64_t b = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int64_t pp = p0 + k;
        const int c = c0 + tx;
        tile[k][tx] = (c < C && pp < HW) ? load_elem(in, in_dt, (b * HW + pp) * in_ld + c) : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k;
        const int64_t pp = p0 + tx;
        if (c < C && pp < HW) out[(b * C + c) * HW + pp] = tile[tx][k];
    }
}

struct Opnd { const void* p; int ld, dt; };
constexpr int kSeBlocks = 512;   // partial sums per image
// 8 consecutive channels as floats <-> one 16-B (f16) or two 16-B (fp32) vectors
__device__ __forceinline__ void load8(const void* p, int dt, int64_t off, float (&v)[8]) {
    if (dt == DSVC_F32) {
        const float4 a = *(const float4*)((const float*)p + off), b = *(const float4*)((const float*)p + off + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const uint4 t = *(const uint4*)((const unsigned short*)p + off);
        const unsigned u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = h16_lo(u[j]); v[2 * j + 1] = h16_hi(u[j]); }
    }
}
__device__ __forceinline__ void store8(void* p, int dt, int64_t off, const float (&v)[8]) {
    if (dt == DSVC_F32) {
        *(float4*)((float*)p + off) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)((float*)p + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        unsigned u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = pk_h16(v[2 * j], v[2 * j + 1]);
        *(uint4*)((unsigned short*)p + off) = make_uint4(u[0], u[1], u[2], u[3]);
    }
}
inline bool vec8_ok(const void* p, int ld, int dt, int C) {
    return p && (C % 8) == 0 && (ld % 8) == 0 && ((uintptr_t)p % 16) == 0 && (dt == DSVC_F32 || dt == DSVC_F16);
}

// warp, 8 channels per thread, 16-B accesses (the 64-channel feature warp of InterLayerPrediction, modules.py:429)
template <typename IT>
__global__ void warp8_kernel(const void* in, int in_ld, int in_dt, const float* flow, int flow_ld,
                             void* out, int out_ld, int out_dt, int B, int H, int W, int C) {
    DSVC_F16_SAT();
    const int cg = C / 8;
    const IT total = (IT)B * H * W * cg;
    for (IT i = (IT)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IT)gridDim.x * blockDim.x) {
        const IT pixi = i / (IT)cg, rowi = pixi / (IT)W;
        const int g = (int)(i - pixi * cg);
        const int64_t pix = (int64_t)pixi;
        const int x = (int)(pixi - rowi * W);
        const int64_t b = (int64_t)(rowi / (IT)H);
        const int y = (int)(rowi - (IT)b * H);
        const float2 f = *(const float2*)(flow + pix * flow_ld);
        const float sx = fminf(fmaxf((float)x + f.x, 0.f), (float)(W - 1)), sy = fminf(fmaxf((float)y + f.y, 0.f), (float)(H - 1));
        const float x0f = floorf(sx), y0f = floorf(sy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float ax = sx - x0f, ay = sy - y0f;
        const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
        const float w00 = (1.f - ax) * (1.f - ay), w01 = ax * (1.f - ay), w10 = (1.f - ax) * ay, w11 = ax * ay;
        const int64_t base = b * H * W;
        float a[8], bb[8], c[8], d[8], o[8];
        load8(in, in_dt, (base + (int64_t)y0 * W + x0) * in_ld + g * 8, a);
        load8(in, in_dt, (base + (int64_t)y0 * W + x1) * in_ld + g * 8, bb);
        load8(in, in_dt, (base + (int64_t)y1 * W + x0) * in_ld + g * 8, c);
        load8(in, in_dt, (base + (int64_t)y1 * W + x1) * in_ld + g * 8, d);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = w00 * a[j] + w01 * bb[j] + w10 * c[j] + w11 * d[j];
        store8(out, out_dt, pix * out_ld + g * 8, o);
    }
}

// elementwise family, 8 channels per thread (all operands 16-B aligned, C % 8 == 0)
__global__ void eltwise8_kernel(int op, int64_t n_pix, int C, Opnd a, Opnd b, Opnd c, Opnd d, Opnd e,
                                void* out, int out_ld, int out_dt) {
    DSVC_F16_SAT();
    const int cg = C / 8;
    const int64_t total = n_pix * cg;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cg) * 8;
        const int64_t pix = i / cg;
        float va[8], vb[8], vc[8], vd[8], ve[8], o[8];
        load8(a.p, a.dt, pix * a.ld + ch, va);
        if (op == DSVC_EW_ADD || op == DSVC_EW_SUB || op == DSVC_EW_BLEND2 || op == DSVC_EW_BLEND3) load8(b.p, b.dt, pix * b.ld + ch, vb);
        if (op == DSVC_EW_BLEND2 || op == DSVC_EW_BLEND3) load8(d.p, d.dt, pix * d.ld + ch, vd);
        if (op == DSVC_EW_BLEND3) { load8(c.p, c.dt, pix * c.ld + ch, vc); load8(e.p, e.dt, pix * e.ld + ch, ve); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            switch (op) {
                case DSVC_EW_ADD: o[j] = va[j] + vb[j]; break;
                case DSVC_EW_SUB: o[j] = va[j] - vb[j]; break;
                case DSVC_EW_BLEND2: o[j] = vd[j] * va[j] + (1.f - vd[j]) * vb[j]; break;
                case DSVC_EW_BLEND3: o[j] = vd[j] * va[j] + ve[j] * vb[j] + (1.f - vd[j] - ve[j]) * vc[j]; break;
                case DSVC_EW_CLAMP01: o[j] = fminf(fmaxf(va[j], 0.f), 1.f); break;
                default: o[j] = va[j]; break;
            }
        }
        store8(out, out_dt, pix * out_ld + ch, o);
    }
}

// SE partial sums, 8 channels per thread: block j sums pixels j, j+kSeBlocks, ... (deterministic order)
__global__ void se_partial8_kernel(const void* x, int ld, int dt, int64_t HW, int C, float* partial) {
    extern __shared__ float red[];
    const int b = blockIdx.y, j = blockIdx.x;
    const int cg = C / 8;                           // thread -> (pixel lane, channel octet); 256 % cg == 0 for C in {32,64,128}
    const int lanes = blockDim.x / cg;
    const int g = threadIdx.x % cg, pl = threadIdx.x / cg;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t p = (int64_t)j * lanes + pl; p < HW; p += (int64_t)kSeBlocks * lanes) {
        float v[8];
        load8(x, dt, ((int64_t)b * HW + p) * ld + g * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[(pl * cg + g) * 8 + k] = s[k];
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        float t = 0.f;
        for (int k = 0; k < lanes; ++k) t += red[(k * cg + c / 8) * 8 + (c & 7)];
        partial[((int64_t)b * kSeBlocks + j) * C + c] = t;
    }
}


__global__ void eltwise_kernel(int op, int64_t n_pix, int C, Opnd a, Opnd b, Opnd c, Opnd d, Opnd e,
                               void* out, int out_ld, int out_dt) {
    DSVC_F16_SAT();
    const int64_t total = n_pix * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % C);
        const int64_t pix = i / C;
        const float va = load_elem(a.p, a.dt, pix * a.ld + ch);
        float v;
        switch (op) {
            case DSVC_EW_ADD: v = va + load_elem(b.p, b.dt, pix * b.ld + ch); break;
            case DSVC_EW_SUB: v = va - load_elem(b.p, b.dt, pix * b.ld + ch); break;
            case DSVC_EW_BLEND2: {
                const float w = load_elem(d.p, d.dt, pix * d.ld + ch);
                v = w * va + (1.f - w) * load_elem(b.p, b.dt, pix * b.ld + ch);
                break;
            }
            case DSVC_EW_BLEND3: {
                const float w = load_elem(d.p, d.dt, pix * d.ld + ch), u = load_elem(e.p, e.dt, pix * e.ld + ch);
                v = w * va + u * load_elem(b.p, b.dt, pix * b.ld + ch) + (1.f - w - u) * load_elem(c.p, c.dt, pix * c.ld + ch);
                break;
            }
            case DSVC_EW_CLAMP01: v = fminf(fmaxf(va, 0.f), 1.f); break;
            default: v = va; break;
        }
        store_elem(out, out_dt, pix * out_ld + ch, v);
    }
}

// ---- SELayer (modules.py:217-230), deterministic two-stage mean ----------------------------------
__global__ void se_partial_kernel(const void* x, int ld, int dt, int64_t HW, int C, float* partial) {
    // block (j, b): pixels j, j+kSeBlocks, ...; thread t handles channel t % C, pixel lane t / C
    extern __shared__ float red[];
    const int b = blockIdx.y, j = blockIdx.x;
    const int lanes = blockDim.x / C;  // pixel lanes per block (C <= 256 and divides 256 on this path)
    const int c = threadIdx.x % C, pl = threadIdx.x / C;
    float s = 0.f;
    if (pl < lanes)
        for (int64_t p = (int64_t)j * lanes + pl; p < HW; p += (int64_t)kSeBlocks * lanes)
            s += load_elem(x, dt, ((int64_t)b * HW + p) * ld + c);
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int k = 0; k < lanes; ++k) t += red[k * C + threadIdx.x];
        partial[((int64_t)b * kSeBlocks + j) * C + threadIdx.x] = t;
    }
}

__global__ void __launch_bounds__(1024) se_gate_kernel(const float* partial, int64_t HW, int C, int hidden, const float* w1, const float* w2, float* gate) {
    extern __shared__ float sm[];  // mean[C] + hid[hidden]
    float* mean = sm;
    float* hid = sm + C;
    const int b = blockIdx.x;
    {   // blockDim / C threads per channel (1024 threads: 16 for C = 64), each a fixed strided subset of the partials whose loads are all
        // issued before the first add, combined in a fixed order (deterministic).  History: one thread per channel walking all 512
        // partials took 38 us of serial load latency per gate; 4 threads per channel with a rolled loop still took 38 us (rocprofv3).
        __shared__ double part[1024];
        const int lanes = blockDim.x / C, c = threadIdx.x % C, l = threadIdx.x / C;     // C divides 256 (checked on the host)
        constexpr int U = 8;
        double s = 0.0;
        if (l < lanes) {
            for (int j0 = l; j0 < kSeBlocks; j0 += lanes * U) {
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = j0 + u * lanes;
                    v[u] = j < kSeBlocks ? partial[((int64_t)b * kSeBlocks + j) * C + c] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) s += (double)v[u];
            }
        }
        part[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < C) {
            double t = 0.0;
            for (int k = 0; k < lanes; ++k) t += part[k * C + threadIdx.x];
            mean[threadIdx.x] = (float)(t / (double)HW);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < hidden; k += blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += w1[k * C + c] * mean[c];
        hid[k] = fmaxf(s, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < hidden; ++k) s += w2[c * hidden + k] * hid[k];
        gate[(int64_t)b * C + c] = 1.f / (1.f + expf(-s));
    }
}

// ---- deterministic reductions -------------------------------------------------------------------
constexpr int kRedBlocks = 1024;
__global__ void reduce_partial_kernel(int mode, int64_t n_pix, int C, Opnd a, Opnd b, float* partial) {
    __shared__ double red[kThreads];
    const int64_t total = n_pix * C;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % C);
        const int64_t pix = i / C;
        const float va = load_elem(a.p, a.dt, pix * a.ld + ch);
        if (mode == 0) s += va;
        else if (mode == 1) s += logf(va);
        else { const float dd = va - load_elem(b.p, b.dt, pix * b.ld + ch); s += dd * dd; }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = kThreads / 2; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) ((double*)partial)[blockIdx.x] = red[0];
}
__global__ void reduce_final_kernel(const float* partial, int n, double* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += ((const double*)partial)[i];
        out[0] = s;
    }
}

inline Opnd mk(const dsvc_operand* o) { return o ? Opnd{o->ptr, o->ld, o->dtype} : Opnd{nullptr, 0, 0}; }
inline bool dt_ok(int dt) { return dt == DSVC_F32 || dt == DSVC_F16; }

// ---- RefineNet1 temporal aggregation (semantic_layer.py:1166-1184) -----------------------------------------------------------
// Per pixel: n(v) = v / ||v||_2 over the C channels; a_l = mean_c (or sum_c) n(rx)[c] * n(rs_l)[c]; w = softmax_l(a);
// out[c] = x[c] + sum_l x[c] * w_l  (evaluated in that order: the weights sum to 1 only up to rounding), optional ReLU
// (OursResNetStudentP.act, semantic_layer.py:1082).  One wave per pixel, C/64 channels per lane (C = 256: one float4 per lane),
// wave-wide sums through DPP-free shuffles in a fixed order (deterministic).  HBM-bound: reads (2 + L) * C floats per pixel.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int MAXL>
__global__ void __launch_bounds__(256) refine_aggregate_kernel(const float* x, int x_ld, const float* rx, int rx_ld, const float* rs, int rs_ld,
                                                               int64_t rs_img, int L, int64_t npix, int C, int mean, int relu,
                                                               float* out, int out_ld) {
    const int lane = threadIdx.x & 63;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= npix) return;
    const int per = C / 64;                       // channels per lane (host checks C % 64 == 0, per <= 4)
    float xv[4], rv[4];
    float n2 = 0.f;
    for (int j = 0; j < per; ++j) {
        xv[j] = x[pix * x_ld + lane * per + j];
        rv[j] = rx[pix * rx_ld + lane * per + j];
        n2 += rv[j] * rv[j];
    }
    const float nx = sqrtf(wave_sum(n2));
    float a[MAXL];
    float amax = -__builtin_inff();
    for (int l = 0; l < L; ++l) {
        float sv[4], s2 = 0.f, dot = 0.f;
        for (int j = 0; j < per; ++j) {
            sv[j] = rs[l * rs_img + pix * rs_ld + lane * per + j];
            s2 += sv[j] * sv[j];
        }
        const float ns = sqrtf(wave_sum(s2));
        for (int j = 0; j < per; ++j) dot += (rv[j] / nx) * (sv[j] / ns);
        a[l] = wave_sum(dot) * (mean ? 1.f / (float)C : 1.f);
        amax = fmaxf(amax, a[l]);
    }
    float den = 0.f;
    for (int l = 0; l < L; ++l) { a[l] = expf(a[l] - amax); den += a[l]; }
    for (int j = 0; j < per; ++j) {
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc += xv[j] * (a[l] / den);
        float o = xv[j] + acc;
        if (relu) o = fmaxf(o, 0.f);
        out[pix * out_ld + lane * per + j] = o;
    }
}

}  // namespace

extern "C" int dsvc_warp_bilinear_border(void* stream, const void* in, int32_t in_ld, int32_t in_dtype,
                                         const float* flow, int32_t flow_ld, void* out, int32_t out_ld, int32_t out_dtype,
                                         int32_t B, int32_t H, int32_t W, int32_t C) {
    DSVC_REQUIRE(in && flow && out, "warp: null pointer");
    DSVC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && in_ld >= C && out_ld >= C && flow_ld >= 2, "warp: bad shape");
    DSVC_REQUIRE(dt_ok(in_dtype) && dt_ok(out_dtype), "warp: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    if (vec8_ok(in, in_ld, in_dtype, C) && vec8_ok(out, out_ld, out_dtype, C) && flow_ld % 2 == 0 && ((uintptr_t)flow % 8) == 0) {
        const int64_t n = (int64_t)B * H * W * (C / 8);
        if (n < kIdx32) hipLaunchKernelGGL(warp8_kernel<unsigned>, dim3(grid_for(n)), dim3(kThreads), 0, st, in, in_ld, in_dtype, flow, flow_ld, out, out_ld, out_dtype, B, H, W, C);
        else hipLaunchKernelGGL(warp8_kernel<int64_t>, dim3(grid_for(n)), dim3(kThreads), 0, st, in, in_ld, in_dtype, flow, flow_ld, out, out_ld, out_dtype, B, H, W, C);
    } else if (C % 4 == 0) {
        const int64_t n = (int64_t)B * H * W * (C / 4);
        if (n < kIdx32) hipLaunchKernelGGL((warp_kernel<4, unsigned>), dim3(grid_for(n)), dim3(kThreads), 0, st, in, in_ld, in_dtype, flow, flow_ld, out, out_ld, out_dtype, B, H, W, C);
        else hipLaunchKernelGGL((warp_kernel<4, int64_t>), dim3(grid_for(n)), dim3(kThreads), 0, st, in, in_ld, in_dtype, flow, flow_ld, out, out_ld, out_dtype, B, H, W, C);
    } else {
        const int64_t n = (int64_t)B * H * W * C;
        if (n < kIdx32) hipLaunchKernelGGL((warp_kernel<1, unsigned>), dim3(grid_for(n)), dim3(kThreads), 0, st, in, in_ld, in_dtype, flow, flow_ld, out, out_ld, out_dtype, B, H, W, C);
        else hipLaunchKernelGGL((warp_kernel<1, int64_t>), dim3(grid_for(n)), dim3(kThreads), 0, st, in, in_ld, in_dtype, flow, flow_ld, out, out_ld, out_dtype, B, H, W, C);
    }
    DSVC_CHECK_LAUNCH("warp_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_upsample2x_bilinear(void* stream, const float* in, int32_t in_ld, float* out, int32_t out_ld,
                                        int32_t B, int32_t H, int32_t W, int32_t C, float scale) {
    DSVC_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0 && in_ld >= C && out_ld >= C, "upsample2x: bad argument");
    const int64_t n = (int64_t)B * 4 * H * W * C;
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, in, in_ld, out, out_ld, B, H, W, C, scale);
    DSVC_CHECK_LAUNCH("upsample2x_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_pool2x2(void* stream, int32_t mode, const void* in, int32_t in_ld, void* out, int32_t out_ld,
                            int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C) {
    DSVC_REQUIRE(in && out && (mode == 0 || mode == 1) && dt_ok(dtype), "pool2x2: bad argument");
    DSVC_REQUIRE(B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C > 0 && in_ld >= C && out_ld >= C, "pool2x2: bad shape H=%d W=%d", H, W);
    const int64_t n = (int64_t)B * (H / 2) * (W / 2) * C;
    const int V = dtype == DSVC_F16 ? 8 : 4, es = dsvc_esize(dtype);
    if (C % V == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0 && ((int64_t)in_ld * es) % 16 == 0 && ((int64_t)out_ld * es) % 16 == 0) {
        const bool i32 = n / V < kIdx32;
        if (dtype == DSVC_F16) {
            if (i32) hipLaunchKernelGGL((pool2x2_vec_kernel<true, unsigned>), dim3(grid_for(n / V)), dim3(kThreads), 0, (hipStream_t)stream, mode, in, in_ld, out, out_ld, B, H, W, C);
            else hipLaunchKernelGGL((pool2x2_vec_kernel<true, int64_t>), dim3(grid_for(n / V)), dim3(kThreads), 0, (hipStream_t)stream, mode, in, in_ld, out, out_ld, B, H, W, C);
        } else {
            if (i32) hipLaunchKernelGGL((pool2x2_vec_kernel<false, unsigned>), dim3(grid_for(n / V)), dim3(kThreads), 0, (hipStream_t)stream, mode, in, in_ld, out, out_ld, B, H, W, C);
            else hipLaunchKernelGGL((pool2x2_vec_kernel<false, int64_t>), dim3(grid_for(n / V)), dim3(kThreads), 0, (hipStream_t)stream, mode, in, in_ld, out, out_ld, B, H, W, C);
        }
        DSVC_CHECK_LAUNCH("pool2x2_vec_kernel");
        return DSVC_OK;
    }
    hipLaunchKernelGGL(pool2x2_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, mode, in, in_ld, out, out_ld, dtype, B, H, W, C);
    DSVC_CHECK_LAUNCH("pool2x2_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_pixel_shuffle_nchw(void* stream, const float* in, void* out, int32_t out_ld, int32_t out_dtype,
                                       int32_t B, int32_t Cout, int32_t Hin, int32_t Win, int32_t r) {
    DSVC_REQUIRE(in && out && B > 0 && Cout > 0 && Hin > 0 && Win > 0 && r >= 1 && out_ld >= Cout && dt_ok(out_dtype), "pixel_shuffle: bad argument");
    const int64_t n = (int64_t)B * Cout * Hin * r * Win * r;
    if (Cout % 8 == 0 && ((uintptr_t)out & 15) == 0 && ((int64_t)out_ld * dsvc_esize(out_dtype)) % 16 == 0) {
        if (out_dtype == DSVC_F16) {
            if (n / 8 < kIdx32) hipLaunchKernelGGL((pixel_shuffle_nchw_vec8_kernel<true, unsigned>), dim3(grid_for(n / 8)), dim3(kThreads), 0, (hipStream_t)stream, in, out, out_ld, B, Cout, Hin, Win, r);
            else hipLaunchKernelGGL((pixel_shuffle_nchw_vec8_kernel<true, int64_t>), dim3(grid_for(n / 8)), dim3(kThreads), 0, (hipStream_t)stream, in, out, out_ld, B, Cout, Hin, Win, r);
        } else {
            if (n / 8 < kIdx32) hipLaunchKernelGGL((pixel_shuffle_nchw_vec8_kernel<false, unsigned>), dim3(grid_for(n / 8)), dim3(kThreads), 0, (hipStream_t)stream, in, out, out_ld, B, Cout, Hin, Win, r);
            else hipLaunchKernelGGL((pixel_shuffle_nchw_vec8_kernel<false, int64_t>), dim3(grid_for(n / 8)), dim3(kThreads), 0, (hipStream_t)stream, in, out, out_ld, B, Cout, Hin, Win, r);
        }
        DSVC_CHECK_LAUNCH("pixel_shuffle_nchw_vec8_kernel");
        return DSVC_OK;
    }
    hipLaunchKernelGGL(pixel_shuffle_nchw_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, in, out, out_ld, out_dtype, B, Cout, Hin, Win, r);
    DSVC_CHECK_LAUNCH("pixel_shuffle_nchw_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_nchw_to_nhwc(void* stream, const float* in, void* out, int32_t out_ld, int32_t out_dtype,
                                 int32_t B, int32_t C, int32_t H, int32_t W) {
    DSVC_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0 && out_ld >= C && dt_ok(out_dtype), "nchw_to_nhwc: bad argument");
    const int64_t HW = (int64_t)H * W;
    dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, out_ld, out_dtype, B, C, HW);
    DSVC_CHECK_LAUNCH("nchw_to_nhwc_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_nhwc_to_nchw(void* stream, const void* in, int32_t in_ld, int32_t in_dtype, float* out,
                                 int32_t B, int32_t C, int32_t H, int32_t W) {
    DSVC_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0 && in_ld >= C && dt_ok(in_dtype), "nhwc_to_nchw: bad argument");
    const int64_t HW = (int64_t)H * W;
    dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, in_ld, in_dtype, out, B, C, HW);
    DSVC_CHECK_LAUNCH("nhwc_to_nchw_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_eltwise(void* stream, int32_t op, int64_t n_pix, int32_t C, const dsvc_operand* a,
                            const dsvc_operand* b, const dsvc_operand* c, const dsvc_operand* d, const dsvc_operand* e,
                            void* out, int32_t out_ld, int32_t out_dtype) {
    DSVC_REQUIRE(a && a->ptr && out && n_pix > 0 && C > 0 && out_ld >= C && dt_ok(out_dtype), "eltwise: bad argument");
    DSVC_REQUIRE(op >= DSVC_EW_COPY && op <= DSVC_EW_CLAMP01, "eltwise: unknown op %d", op);
    const bool need_b = op == DSVC_EW_ADD || op == DSVC_EW_SUB || op == DSVC_EW_BLEND2 || op == DSVC_EW_BLEND3;
    DSVC_REQUIRE(!need_b || (b && b->ptr), "eltwise: operand b missing");
    DSVC_REQUIRE(!(op == DSVC_EW_BLEND2 || op == DSVC_EW_BLEND3) || (d && d->ptr), "eltwise: operand d missing");
    DSVC_REQUIRE(op != DSVC_EW_BLEND3 || (c && c->ptr && e && e->ptr), "eltwise: operands c/e missing");
    auto okv = [&](const dsvc_operand* o) { return !o || !o->ptr || vec8_ok(o->ptr, o->ld, o->dtype, C); };
    if (vec8_ok(out, out_ld, out_dtype, C) && okv(a) && okv(b) && okv(c) && okv(d) && okv(e))
        hipLaunchKernelGGL(eltwise8_kernel, dim3(grid_for(n_pix * (C / 8))), dim3(kThreads), 0, (hipStream_t)stream, op, n_pix, C,
                           mk(a), mk(b), mk(c), mk(d), mk(e), out, out_ld, out_dtype);
    else
        hipLaunchKernelGGL(eltwise_kernel, dim3(grid_for(n_pix * C)), dim3(kThreads), 0, (hipStream_t)stream, op, n_pix, C,
                           mk(a), mk(b), mk(c), mk(d), mk(e), out, out_ld, out_dtype);
    DSVC_CHECK_LAUNCH("eltwise_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_refine_aggregate(void* stream, const float* x, int32_t x_ld, const float* rx, int32_t rx_ld, const float* rs, int32_t rs_ld,
                                     int32_t L, int64_t n_pix, int32_t C, int32_t mean, int32_t relu, float* out, int32_t out_ld) {
    DSVC_REQUIRE(x && rx && rs && out && n_pix > 0 && L >= 1 && L <= 16, "refine_aggregate: bad argument (1 <= supports <= 16)");
    DSVC_REQUIRE(C > 0 && C % 64 == 0 && C <= 256 && x_ld >= C && rx_ld >= C && rs_ld >= C && out_ld >= C, "refine_aggregate: C=%d must be 64..256, multiple of 64", C);
    const int64_t blocks = (n_pix + 3) / 4;
    hipLaunchKernelGGL(refine_aggregate_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, x_ld, rx, rx_ld, rs, rs_ld,
                       n_pix * rs_ld, L, n_pix, C, mean, relu, out, out_ld);
    DSVC_CHECK_LAUNCH("refine_aggregate_kernel");
    return DSVC_OK;
}

extern "C" int64_t dsvc_se_workspace_floats(int32_t B, int32_t C) { return (int64_t)B * kSeBlocks * C; }

extern "C" int dsvc_se_gate(void* stream, const void* x, int32_t ld, int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C,
                            const float* w1, const float* w2, int32_t hidden, float* partial, float* gate) {
    DSVC_REQUIRE(x && w1 && w2 && partial && gate && dt_ok(dtype), "se_gate: null/bad argument");
    DSVC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C <= 256 && 256 % C == 0 && hidden > 0 && ld >= C, "se_gate: C=%d must divide 256", C);
    hipStream_t st = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W;
    if (vec8_ok(x, ld, dtype, C) && 256 % (C / 8) == 0)
        hipLaunchKernelGGL(se_partial8_kernel, dim3(kSeBlocks, B), dim3(256), 256 * 8 * sizeof(float), st, x, ld, dtype, HW, C, partial);
    else
        hipLaunchKernelGGL(se_partial_kernel, dim3(kSeBlocks, B), dim3(256), 256 * sizeof(float), st, x, ld, dtype, HW, C, partial);
    DSVC_CHECK_LAUNCH("se_partial_kernel");
    hipLaunchKernelGGL(se_gate_kernel, dim3(B), dim3(1024), (C + hidden) * sizeof(float), st, partial, HW, C, hidden, w1, w2, gate);
    DSVC_CHECK_LAUNCH("se_gate_kernel");
    return DSVC_OK;
}

extern "C" int64_t dsvc_reduce_workspace_floats(void) { return 2 * kRedBlocks; }

extern "C" int dsvc_reduce(void* stream, int32_t mode, int64_t n_pix, int32_t C, const dsvc_operand* a, const dsvc_operand* b,
                           float* partial, double* out) {
    DSVC_REQUIRE(a && a->ptr && partial && out && n_pix > 0 && C > 0 && mode >= 0 && mode <= 2, "reduce: bad argument");
    DSVC_REQUIRE(mode != 2 || (b && b->ptr), "reduce: operand b missing");
    DSVC_REQUIRE(((uintptr_t)partial & 7) == 0, "reduce: workspace must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int64_t g = (n_pix * C + kThreads - 1) / kThreads;
    const int nb = (int)(g < kRedBlocks ? g : kRedBlocks);
    hipLaunchKernelGGL(reduce_partial_kernel, dim3(nb), dim3(kThreads), 0, st, mode, n_pix, C, mk(a), mk(b), partial);
    DSVC_CHECK_LAUNCH("reduce_partial_kernel");
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, st, (const float*)partial, nb, out);
    DSVC_CHECK_LAUNCH("reduce_final_kernel");
    return DSVC_OK;
}

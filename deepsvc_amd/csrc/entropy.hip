// entropy.hip -- device side of the hyperprior / channel-conditional entropy model
// (compressai 1.2.1 semantics; reference call sites
// image_model.py:155-162,181-183,206-207,237-239,260,286,290).
//
// Symbol and index planes are int32 in the reference's flattening order [C][h][w]
// (image_model.py:241-242 `y_q_slice.reshape(-1)` of an NCHW tensor) so the host range coder
// consumes them as-is.  All rounding is round-half-to-even (torch.round) = rintf.
// Every kernel here is elementwise => deterministic => encoder and decoder see identical mu/index.
#include "common.h"

namespace {
constexpr int kThreads = 256;
inline unsigned grid_for(int64_t n) {
    int64_t g = (n + kThreads - 1) / kThreads;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

__device__ __forceinline__ float phi_c(float x) { return 0.5f * erfcf(-0.70710678118654752440f * x); }

// thread index i enumerates the PLANAR order (b, c, y, x) so symbol/index stores are coalesced;
// NHWC operands at 1/16 resolution are small (L2-resident), their strided reads are cheap.
__global__ void gc_quantize_kernel(const void* y, int y_ld, const void* mu, int mu_ld, const void* sc, int sc_ld, int dt,
                                   const float* table, int levels, int B, int H, int W, int C,
                                   int32_t* sym, int32_t* idx, void* yhat, int yhat_ld) {
    DSVC_F16_SAT();
    const int64_t HW = (int64_t)H * W, total = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % HW;
        const int c = (int)((i / HW) % C);
        const int64_t b = i / (HW * C);
        const int64_t pix = b * HW + p;
        const float m = load_elem(mu, dt, pix * mu_ld + c);
        if (sym || yhat) {
            const float q = rintf(load_elem(y, dt, pix * y_ld + c) - m);
            if (sym) sym[i] = (int32_t)q;
            if (yhat) store_elem(yhat, dt, pix * yhat_ld + c, q + m);
        }
        if (idx) {
            const float s = fmaxf(load_elem(sc, dt, pix * sc_ld + c), 0.11f);
            int k = levels - 1;
            for (int t = 0; t < levels - 1; ++t) k -= (s <= table[t]) ? 1 : 0;
            idx[i] = k;
        }
    }
}

__global__ void gc_dequantize_kernel(const int32_t* sym, const void* mu, int mu_ld, int dt, int B, int H, int W, int C,
                                     void* yhat, int yhat_ld) {
    DSVC_F16_SAT();
    const int64_t HW = (int64_t)H * W, total = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % HW;
        const int c = (int)((i / HW) % C);
        const int64_t b = i / (HW * C);
        const int64_t pix = b * HW + p;
        store_elem(yhat, dt, pix * yhat_ld + c, (float)sym[i] + load_elem(mu, dt, pix * mu_ld + c));
    }
}

__global__ void gc_likelihood_kernel(const void* y, int y_ld, const void* mu, int mu_ld, const void* sc, int sc_ld, int dt,
                                     int64_t n_pix, int C, void* yhat, int yhat_ld, float* lik, int lik_ld) {
    DSVC_F16_SAT();
    const int64_t total = n_pix * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const float m = load_elem(mu, dt, pix * mu_ld + c);
        const float yh = rintf(load_elem(y, dt, pix * y_ld + c) - m) + m;
        if (yhat) store_elem(yhat, dt, pix * yhat_ld + c, yh);
        if (lik) {
            const float v = fabsf(yh - m);
            const float s = fmaxf(load_elem(sc, dt, pix * sc_ld + c), 0.11f);
            const float l = phi_c((0.5f - v) / s) - phi_c((-0.5f - v) / s);
            lik[pix * lik_ld + c] = fmaxf(l, 1e-9f);
        }
    }
}

// factorised prior: per-channel 1-3-3-3-3-1 MLP on the cumulative logits (A.2).  params per channel:
// [0..2] M0(3x1) [3..11] M1(3x3) [12..20] M2 [21..29] M3 [30..32] M4(1x3) [33..35] b0 [36..38] b1 [39..41] b2
// [42..44] b3 [45] b4 [46..48] f0 [49..51] f1 [52..54] f2 [55..57] f3 [58] median; M = softplus(matrix),
// f = tanh(factor) precomputed on the host.
__device__ __forceinline__ float eb_logits(const float* P, float x) {
    float h[3], g[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { h[i] = P[i] * x + P[33 + i]; h[i] += P[46 + i] * tanhf(h[i]); }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const float* M = P + 3 + 9 * l;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            g[i] = M[3 * i] * h[0] + M[3 * i + 1] * h[1] + M[3 * i + 2] * h[2] + P[36 + 3 * l + i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) h[i] = g[i] + P[49 + 3 * l + i] * tanhf(g[i]);
    }
    return P[30] * h[0] + P[31] * h[1] + P[32] * h[2] + P[45];
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void eb_process_kernel(const void* z, int z_ld, int dt, const float* params, int B, int H, int W, int C, int mode,
                                  int32_t* sym, void* zhat, int zhat_ld, float* lik, int lik_ld) {
    DSVC_F16_SAT();
    const int64_t HW = (int64_t)H * W, total = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % HW;
        const int c = (int)((i / HW) % C);
        const int64_t b = i / (HW * C);
        const int64_t pix = b * HW + p;
        const float* P = params + (int64_t)c * DSVC_EB_PARAMS_PER_CH;
        const float med = P[58];
        const float q = rintf(load_elem(z, dt, pix * z_ld + c) - med);
        if (mode & 1) sym[i] = (int32_t)q;
        const float zh = q + med;
        if (mode & 2) store_elem(zhat, dt, pix * zhat_ld + c, zh);
        if (mode & 4) {
            const float lo = eb_logits(P, zh - 0.5f), up = eb_logits(P, zh + 0.5f);
            const float sum = lo + up;
            const float sg = sum > 0.f ? -1.f : (sum < 0.f ? 1.f : 0.f);
            lik[pix * lik_ld + c] = fmaxf(fabsf(sigmoidf(sg * up) - sigmoidf(sg * lo)), 1e-9f);
        }
    }
}

__global__ void eb_dequantize_kernel(const int32_t* sym, const float* params, int dt, int B, int H, int W, int C, void* zhat, int zhat_ld) {
    DSVC_F16_SAT();
    const int64_t HW = (int64_t)H * W, total = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % HW;
        const int c = (int)((i / HW) % C);
        const int64_t b = i / (HW * C);
        store_elem(zhat, dt, (b * HW + p) * zhat_ld + c, (float)sym[i] + params[(int64_t)c * DSVC_EB_PARAMS_PER_CH + 58]);
    }
}
inline bool dt_ok(int dt) { return dt == DSVC_F32 || dt == DSVC_F16; }
}  // namespace

extern "C" int dsvc_gc_quantize(void* stream, const void* y, int32_t y_ld, const void* mu, int32_t mu_ld,
                                const void* scale, int32_t scale_ld, int32_t dtype, const float* table, int32_t levels,
                                int32_t B, int32_t H, int32_t W, int32_t C,
                                int32_t* symbols, int32_t* indexes, void* y_hat, int32_t y_hat_ld) {
    DSVC_REQUIRE(mu && dt_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0, "gc_quantize: bad argument");
    DSVC_REQUIRE(!(symbols || y_hat) || y, "gc_quantize: y missing");
    DSVC_REQUIRE(!indexes || (scale && table && levels > 1), "gc_quantize: scale/table missing");
    const int64_t n = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(gc_quantize_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, y, y_ld, mu, mu_ld, scale, scale_ld,
                       dtype, table, levels, B, H, W, C, symbols, indexes, y_hat, y_hat_ld);
    DSVC_CHECK_LAUNCH("gc_quantize_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_gc_dequantize(void* stream, const int32_t* symbols, const void* mu, int32_t mu_ld, int32_t dtype,
                                  int32_t B, int32_t H, int32_t W, int32_t C, void* y_hat, int32_t y_hat_ld) {
    DSVC_REQUIRE(symbols && mu && y_hat && dt_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0, "gc_dequantize: bad argument");
    const int64_t n = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(gc_dequantize_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, symbols, mu, mu_ld, dtype, B, H, W, C, y_hat, y_hat_ld);
    DSVC_CHECK_LAUNCH("gc_dequantize_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_gc_likelihood(void* stream, const void* y, int32_t y_ld, const void* mu, int32_t mu_ld,
                                  const void* scale, int32_t scale_ld, int32_t dtype, int64_t n_pix, int32_t C,
                                  void* y_hat, int32_t y_hat_ld, float* lik, int32_t lik_ld) {
    DSVC_REQUIRE(y && mu && dt_ok(dtype) && n_pix > 0 && C > 0, "gc_likelihood: bad argument");
    DSVC_REQUIRE(!lik || scale, "gc_likelihood: scale missing");
    hipLaunchKernelGGL(gc_likelihood_kernel, dim3(grid_for(n_pix * C)), dim3(kThreads), 0, (hipStream_t)stream, y, y_ld, mu, mu_ld,
                       scale, scale_ld, dtype, n_pix, C, y_hat, y_hat_ld, lik, lik_ld);
    DSVC_CHECK_LAUNCH("gc_likelihood_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_eb_process(void* stream, const void* z, int32_t z_ld, int32_t dtype, const float* params,
                               int32_t B, int32_t H, int32_t W, int32_t C, int32_t mode,
                               int32_t* symbols, void* z_hat, int32_t z_hat_ld, float* lik, int32_t lik_ld) {
    DSVC_REQUIRE(z && params && dt_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0 && mode > 0 && mode < 8, "eb_process: bad argument");
    DSVC_REQUIRE(!(mode & 1) || symbols, "eb_process: symbols missing");
    DSVC_REQUIRE(!(mode & 2) || z_hat, "eb_process: z_hat missing");
    DSVC_REQUIRE(!(mode & 4) || lik, "eb_process: lik missing");
    const int64_t n = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(eb_process_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, z, z_ld, dtype, params, B, H, W, C, mode,
                       symbols, z_hat, z_hat_ld, lik, lik_ld);
    DSVC_CHECK_LAUNCH("eb_process_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_eb_dequantize(void* stream, const int32_t* symbols, const float* params, int32_t dtype,
                                  int32_t B, int32_t H, int32_t W, int32_t C, void* z_hat, int32_t z_hat_ld) {
    DSVC_REQUIRE(symbols && params && z_hat && dt_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0, "eb_dequantize: bad argument");
    const int64_t n = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(eb_dequantize_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, symbols, params, dtype, B, H, W, C, z_hat, z_hat_ld);
    DSVC_CHECK_LAUNCH("eb_dequantize_kernel");
    return DSVC_OK;
}

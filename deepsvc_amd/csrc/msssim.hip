// msssim.hip -- one level of MS-SSIM (pytorch-msssim 0.2.1 semantics, used by video_model.py:107,114,132 and
// test_video.py:311,359-363) and the 2x2 average pooling between levels.
//
// Per level and (batch, channel): mean over the VALID 11x11 Gaussian-window positions of
//     cs   = (2*s12 + C2) / (s1 + s2 + C2)
//     ssim = (2*m1*m2 + C1) / (m1^2 + m2^2 + C1) * cs
// with m = G*x, s1 = G*x^2 - m1^2, s2 = G*y^2 - m2^2, s12 = G*xy - m1*m2, G = separable 11-tap Gaussian (sigma 1.5),
// C1 = (0.01 L)^2, C2 = (0.03 L)^2.  HBM-bound (two 3-channel frames read once per level); one workgroup owns a 32x8 tile
// of window positions, stages the (32+10)x(8+10) patch of both images in LDS, filters horizontally then vertically out
// of LDS, and leaves one deterministic partial sum per workgroup (no atomics); a second tiny kernel adds them in fp64.
#include "common.h"

namespace {

constexpr int TW = 32, THH = 8, WIN = 11, PW = TW + WIN - 1, PH = THH + WIN - 1;

struct SsimArgs {
    const void* x; const void* y;
    int x_ld, y_ld, x_dtype, y_dtype;
    int B, H, W, C;
    int tiles_x, tiles_y;
    float c1, c2;
    float g[WIN];
};

__global__ void __launch_bounds__(256) ssim_partial_kernel(const SsimArgs a, float* partial) {
    __shared__ float px[PH][PW + 1], py[PH][PW + 1];
    __shared__ float hq[5][PH][TW + 1];          // horizontally filtered x, y, xx, yy, xy
    __shared__ float red[2][256];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int c = bid % a.C, b = bid / a.C;
    const int Ho = a.H - WIN + 1, Wo = a.W - WIN + 1;
    const int oy0 = ty * THH, ox0 = tx * TW;
    for (int i = tid; i < PH * PW; i += 256) {
        const int r = i / PW, q = i - r * PW;
        const int iy = oy0 + r, ix = ox0 + q;
        float vx = 0.f, vy = 0.f;
        if (iy < a.H && ix < a.W) {
            const int64_t pix = ((int64_t)b * a.H + iy) * a.W + ix;
            vx = load_elem(a.x, a.x_dtype, pix * a.x_ld + c);
            vy = load_elem(a.y, a.y_dtype, pix * a.y_ld + c);
        }
        px[r][q] = vx; py[r][q] = vy;
    }
    __syncthreads();
    for (int i = tid; i < PH * TW; i += 256) {
        const int r = i / TW, q = i - r * TW;
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float vx = px[r][q + k], vy = py[r][q + k], g = a.g[k];
            sx += g * vx; sy += g * vy; sxx += g * vx * vx; syy += g * vy * vy; sxy += g * vx * vy;
        }
        hq[0][r][q] = sx; hq[1][r][q] = sy; hq[2][r][q] = sxx; hq[3][r][q] = syy; hq[4][r][q] = sxy;
    }
    __syncthreads();
    float s_ssim = 0.f, s_cs = 0.f;
    {
        const int r = tid / TW, q = tid - r * TW;      // 256 threads == 8 x 32 window positions
        if (oy0 + r < Ho && ox0 + q < Wo) {
            float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float g = a.g[k];
                m1 += g * hq[0][r + k][q]; m2 += g * hq[1][r + k][q];
                e11 += g * hq[2][r + k][q]; e22 += g * hq[3][r + k][q]; e12 += g * hq[4][r + k][q];
            }
            const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
            const float s1 = e11 - m11, s2 = e22 - m22, s12 = e12 - m12;
            const float cs = (2.f * s12 + a.c2) / (s1 + s2 + a.c2);
            s_cs = cs;
            s_ssim = (2.f * m12 + a.c1) / (m11 + m22 + a.c1) * cs;
        }
    }
    red[0][tid] = s_ssim; red[1][tid] = s_cs;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {                 // fixed tree: deterministic
        if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
        __syncthreads();
    }
    if (tid == 0) { partial[2 * (size_t)blockIdx.x] = red[0][0]; partial[2 * (size_t)blockIdx.x + 1] = red[1][0]; }
}

// one wave per (b, c): sums its tiles' partials in fp64 and divides by the number of window positions
__global__ void ssim_final_kernel(const float* partial, int tiles, double inv_n, float* out_ssim, float* out_cs) {
    const int bc = blockIdx.x, lane = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int i = lane; i < tiles; i += 64) {
        s0 += partial[2 * ((size_t)bc * tiles + i)];
        s1 += partial[2 * ((size_t)bc * tiles + i) + 1];
    }
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); }
    if (lane == 0) { out_ssim[bc] = (float)(s0 * inv_n); out_cs[bc] = (float)(s1 * inv_n); }
}

// F.avg_pool2d(x, kernel_size=2, padding=(H%2, W%2)) with count_include_pad=True: window starts at 2*o - pad, missing
// elements count as zeros, always divided by 4
__global__ void avgpool2_pad_kernel(const void* x, int x_ld, int x_dtype, float* out, int B, int H, int W, int C, int Ho, int Wo,
                                    int pad_y, int pad_x) {
    const int64_t n = (int64_t)B * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float s = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int iy = 2 * oy - pad_y + dy, ix = 2 * ox - pad_x + dx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) s += load_elem(x, x_dtype, (((int64_t)b * H + iy) * W + ix) * x_ld + c);
            }
        out[i] = 0.25f * s;
    }
}

}  // namespace

extern "C" int64_t dsvc_ssim_workspace_floats(int32_t B, int32_t H, int32_t W, int32_t C) {
    if (H < WIN || W < WIN) return 0;
    const int64_t tiles = (int64_t)((W - WIN + 1 + TW - 1) / TW) * ((H - WIN + 1 + THH - 1) / THH);
    return 2 * tiles * B * C;
}

extern "C" int dsvc_ssim_level(void* stream, const dsvc_operand* x, const dsvc_operand* y, int32_t B, int32_t H, int32_t W,
                               int32_t C, float data_range, float* workspace, float* out_ssim, float* out_cs) {
    DSVC_REQUIRE(x && y && workspace && out_ssim && out_cs, "ssim_level: null argument");
    DSVC_REQUIRE(H >= WIN && W >= WIN && B > 0 && C > 0, "ssim_level: image %dx%d smaller than the 11x11 window", H, W);
    SsimArgs a;
    a.x = x->ptr; a.y = y->ptr; a.x_ld = x->ld; a.y_ld = y->ld; a.x_dtype = x->dtype; a.y_dtype = y->dtype;
    a.B = B; a.H = H; a.W = W; a.C = C;
    a.tiles_x = (W - WIN + 1 + TW - 1) / TW; a.tiles_y = (H - WIN + 1 + THH - 1) / THH;
    a.c1 = (0.01f * data_range) * (0.01f * data_range); a.c2 = (0.03f * data_range) * (0.03f * data_range);
    {   // _fspecial_gauss_1d(11, 1.5): exp(-(i-5)^2 / (2 sigma^2)) normalised, computed in fp32 like torch does
        float g[WIN], sum = 0.f;
        for (int i = 0; i < WIN; ++i) { const float d = (float)(i - WIN / 2); g[i] = expf(-(d * d) / (2.f * 1.5f * 1.5f)); sum += g[i]; }
        for (int i = 0; i < WIN; ++i) a.g[i] = g[i] / sum;
    }
    const int tiles = a.tiles_x * a.tiles_y;
    const int64_t nblocks = (int64_t)tiles * B * C;
    DSVC_REQUIRE(nblocks < (1ll << 31), "ssim_level: grid too large");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ssim_partial_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, a, workspace);
    DSVC_CHECK_LAUNCH("ssim_partial_kernel");
    const double inv_n = 1.0 / ((double)(H - WIN + 1) * (double)(W - WIN + 1));
    hipLaunchKernelGGL(ssim_final_kernel, dim3(B * C), dim3(64), 0, st, (const float*)workspace, tiles, inv_n, out_ssim, out_cs);
    DSVC_CHECK_LAUNCH("ssim_final_kernel");
    return DSVC_OK;
}

extern "C" int dsvc_avgpool2_pad(void* stream, const dsvc_operand* x, float* out, int32_t B, int32_t H, int32_t W, int32_t C) {
    DSVC_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0, "avgpool2_pad: bad argument");
    const int pad_y = H & 1, pad_x = W & 1;
    const int Ho = (H + 2 * pad_y - 2) / 2 + 1, Wo = (W + 2 * pad_x - 2) / 2 + 1;
    const int64_t n = (int64_t)B * Ho * Wo * C;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(avgpool2_pad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x->ptr, x->ld, x->dtype, out, B, H, W, C,
                       Ho, Wo, pad_y, pad_x);
    DSVC_CHECK_LAUNCH("avgpool2_pad_kernel");
    return DSVC_OK;
}

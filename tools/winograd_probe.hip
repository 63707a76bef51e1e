// tools/winograd_probe.hip -- what would the MFMA phase of the dominant conv (3x3 s1 64 -> 64, f16 operands, fp32 accumulate,
// csrc/conv3x3_pp.hip) cost as a Winograd convolution on a gfx950 CU?  VERDICT round 3 item 3 asked for a measurement instead of a sizing.
//
// This is an INSTRUCTION-MIX probe, not a convolution: every mode runs, per wave and "tile step", the LDS reads, vector-ALU transforms and
// MFMAs that one wave of the two-group kernel would issue for its 2 x 32 output pixels x 64 output channels over all 64 input channels,
// with the data dependences of the real thing (fragment reads -> input transform -> MFMA operands -> accumulators -> output transform),
// on random f16 data resident in LDS (80-byte row pitch, as in the product kernel).  Nothing is loaded from or stored to HBM inside the
// timed loop, so the numbers are the M-phase cost alone: cycles per tile step (s_memtime), the clock the chip holds under that mix
// (s_memtime / s_memrealtime, DVFS give-back item 6 of MI355X_MICROARCH.md) and the resulting rate in direct-convolution FLOP.
//
//   mode 0  direct           144 MFMA (32x32x16) + 144 ds_read_b128 per tile step                    (= conv3x3_pp's two M phases)
//   mode 1  F(2,3) along x    96 MFMA + 96 weight reads + 48 pixel reads + 192 v_pk_add_f16 (input transform B^T d) + 128 v_add_f32
//                            (output transform A^T m); 4 positions x 2 N-tiles of accumulators = 128 registers; transformed weights
//                            3 rows x 4 positions x 64 x 64 = 98 KB dense (fits beside two patches only without the 80-byte padding)
//   mode 2  F(2x2,3x3)        64 MFMA + 64 weight reads + 128 pixel reads + 512 v_pk_add_f16 + 512 v_add_f32; a wave owns 32 tiles x 32
//                            output channels and walks the 16 positions row by row (4 live position accumulators + 4 output
//                            accumulators = 128 registers: all 16 at once would be 256); transformed weights 131 KB dense: they do NOT
//                            fit beside the patches, the probe just reads them from a 82 KB window
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/winograd_probe tools/winograd_probe.hip && /tmp/winograd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;

constexpr int ROWB = 80;                 // LDS row pitch (64 data bytes + 16 pad: conflict-free ds_read_b128 over 32 consecutive rows)
constexpr int W_ROWS = 16 * 64;          // weight window: 16 (position / tap) x 64 output-channel rows = 82 KB
constexpr int P_ROWS = 2 * 12 * 40;      // two input-chunk patches of 12 x 40 pixel rows = 77 KB  (total 159 KB)
constexpr int LDS_BYTES = (W_ROWS + P_ROWS) * ROWB;

__device__ __forceinline__ h16x8 ldsv(const char* p) { return *(const h16x8*)p; }

template <int MODE, int WPS>
__global__ void __launch_bounds__(256 * WPS, WPS) probe(const uint4* __restrict__ rnd, float* __restrict__ sink_out, unsigned long long* __restrict__ stamps, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 16; i += blockDim.x) ((uint4*)smem)[i] = rnd[i];
    __syncthreads();
    const char* lw = smem;                                   // weights
    const char* lp = smem + W_ROWS * ROWB;                   // patches: [chunk][row 12][col 40]
    const int r = lane & 31, h = lane >> 5;
    const int wrow = (wave & 3) * 2;                         // this wave's first tile row inside the patch
    f32x16 sink = {};
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int st = 0; st < steps; ++st) {
        asm volatile("" ::: "memory");                       // LDS is constant here: without this every fragment read is hoisted out of the loop (and spilled)
        if constexpr (MODE == 0) {
            // ---- direct: 2 chunks x 9 taps x 2 k-halves x (2 rows x 2 N-tiles) ----
            f32x16 acc[2][2] = {};
#pragma unroll 1
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        h16x8 af[2], bf[2];
#pragma unroll
                        for (int n = 0; n < 2; ++n) bf[n] = ldsv(lw + ((t * 64 + n * 32 + r) * ROWB) + h * 32 + ks * 16);
#pragma unroll
                        for (int m = 0; m < 2; ++m) af[m] = ldsv(lp + ((ch * 12 + wrow + m + t / 3) * 40 + r + t % 3) * ROWB + h * 32 + ks * 16);
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[n], af[m], acc[m][n], 0, 0, 0);
                        if (WPS == 2) __builtin_amdgcn_sched_barrier(0);    // two waves per SIMD cover each other's LDS latency; keeps the scheduler from hoisting every read (spills)
                    }
            sink += acc[0][0] + acc[0][1] + acc[1][0] + acc[1][1];
        } else if constexpr (MODE == 1) {
            // ---- F(2,3) along x: lane tile = (row r>>4, column pair r&15); even / odd pixel columns in separate planes of 20 rows ----
            f32x16 acc[4][2] = {};
            const int trow = r >> 4, tcol = r & 15;
#pragma unroll 1
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        h16x8 d[4];
#pragma unroll
                        for (int dx = 0; dx < 4; ++dx)
                            d[dx] = ldsv(lp + ((ch * 12 + wrow + trow + ty) * 40 + (dx & 1) * 20 + tcol + (dx >> 1)) * ROWB + h * 32 + ks * 16);
                        const h16x8 v[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};        // B^T d: 16 v_pk_add_f16
#pragma unroll
                        for (int pos = 0; pos < 4; ++pos)
#pragma unroll
                            for (int n = 0; n < 2; ++n) {
                                const h16x8 wf = ldsv(lw + (((ty * 4 + pos) * 64 + n * 32 + r) * ROWB) + h * 32 + ks * 16);
                                acc[pos][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, v[pos], acc[pos][n], 0, 0, 0);
                            }
                        if (WPS == 2) __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
            for (int n = 0; n < 2; ++n) {                       // A^T m: two outputs per tile and channel
                const f32x16 y0 = acc[0][n] + acc[1][n] + acc[2][n], y1 = acc[1][n] - acc[2][n] - acc[3][n];
                sink += y0 * 1.0001f + y1;
            }
        } else {
            // ---- F(2x2,3x3): wave = 32 tiles (2 tile rows x 16) x 32 output channels; positions walked row by row ----
            const int trow = r >> 4, tcol = r & 15, nt = wave & 1;
            f32x16 y[2][2] = {};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x16 mrow[4] = {};
                constexpr int RA[4] = {0, 1, 2, 1}, RB[4] = {2, 2, 1, 3};
#pragma unroll 1
                for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        h16x8 w4[4];
#pragma unroll
                        for (int dx = 0; dx < 4; ++dx) {         // row combination of B^T d B: two pixel rows per transformed row
                            const h16x8 a = ldsv(lp + ((ch * 12 + ((wave >> 1) & 1) * 4 + 2 * trow + RA[i]) * 40 + (dx & 1) * 20 + tcol + (dx >> 1)) * ROWB + h * 32 + ks * 16);
                            const h16x8 b = ldsv(lp + ((ch * 12 + ((wave >> 1) & 1) * 4 + 2 * trow + RB[i]) * 40 + (dx & 1) * 20 + tcol + (dx >> 1)) * ROWB + h * 32 + ks * 16);
                            w4[dx] = (i == 1) ? a + b : a - b;
                        }
                        const h16x8 v[4] = {w4[0] - w4[2], w4[1] + w4[2], w4[2] - w4[1], w4[1] - w4[3]};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const h16x8 wf = ldsv(lw + (((i * 4 + j) * 64 + nt * 32 + r) * ROWB) + h * 32 + ks * 16);
                            mrow[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, v[j], mrow[j], 0, 0, 0);
                        }
                        if (WPS == 2) __builtin_amdgcn_sched_barrier(0);
                    }
                const f32x16 q0 = mrow[0] + mrow[1] + mrow[2], q1 = mrow[1] - mrow[2] - mrow[3];       // A^T over j
                if (i == 0) { y[0][0] += q0; y[0][1] += q1; }
                else if (i == 1) { y[0][0] += q0; y[0][1] += q1; y[1][0] += q0; y[1][1] += q1; }
                else if (i == 2) { y[0][0] += q0; y[0][1] += q1; y[1][0] -= q0; y[1][1] -= q1; }
                else { y[1][0] -= q0; y[1][1] -= q1; }
                if (WPS == 2) __builtin_amdgcn_sched_barrier(0);
            }
            sink += y[0][0] + y[0][1] * 1.0001f + y[1][0] * 0.9999f + y[1][1];
        }
    }
    if (tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += sink[i];
    sink_out[blockIdx.x * blockDim.x + tid] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int WPS>
int run(const char* name, const uint4* rnd, float* sink, unsigned long long* stamps, int steps, double out_per_wave_step) {
    auto k = probe<MODE, WPS>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const int grid = 256, block = 256 * WPS;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), LDS_BYTES, 0, rnd, sink, stamps, steps);
    CK(hipDeviceSynchronize());
    // >= 2 s of back-to-back launches first, so that the clock has settled (DVFS give-back item 6)
    const int warm = 40, timed = 10;
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), LDS_BYTES, 0, rnd, sink, stamps, steps);
    CK(hipEventRecord(e0));
    for (int i = 0; i < timed; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), LDS_BYTES, 0, rnd, sink, stamps, steps);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hs(2 * grid);
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (int b = 0; b < grid; ++b) { cyc.push_back((double)hs[2 * b] / steps); clk.push_back((double)hs[2 * b] / (double)hs[2 * b + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double us = ms * 1e3 / timed;
    // direct-convolution FLOP that the outputs of one step stand for: outputs x 2 x 64 input channels x 9 taps
    const double flop = (double)grid * 4 * WPS * steps * out_per_wave_step * 2.0 * 64 * 9;
    printf("%-34s %d wave(s)/SIMD: %8.0f clocks per wave and tile step (median), in-kernel clock %5.0f MHz, %8.1f us per launch, %6.0f direct-conv TFLOP/s-equivalent\n",
           name, WPS, cyc[grid / 2], clk[grid / 2], us, flop / (us * 1e-6) / 1e12);
    return 0;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 600;
    uint4* rnd; float* sink; unsigned long long* stamps;
    std::vector<unsigned short> h(LDS_BYTES / 2);
    unsigned s = 12345u;
    for (auto& v : h) {                                       // random f16 in (-1, 1): sign, exponent 2^-8 .. 2^-1, random mantissa
        s = s * 1664525u + 1013904223u;
        const unsigned e = 7 + ((s >> 20) & 7), m = (s >> 8) & 0x3FF, sg = (s >> 31) & 1;
        v = (unsigned short)((sg << 15) | (e << 10) | m);
    }
    CK(hipMalloc(&rnd, LDS_BYTES)); CK(hipMalloc(&sink, 256 * 512 * 4)); CK(hipMalloc(&stamps, 2 * 256 * 8));
    CK(hipMemcpy(rnd, h.data(), LDS_BYTES, hipMemcpyHostToDevice));
    printf("# tools/winograd_probe.hip: M-phase instruction-mix probe, %d tile steps per launch, 256 workgroups (one per CU), random f16 data in LDS\n", steps);
    printf("# one tile step = what ONE wave issues for 4096 outputs (64 pixels x 64 channels, or 128 x 32 in mode 2) over 64 input channels x 3x3 taps\n");
    for (int rep = 0; rep < 2; ++rep) {
        if (run<0, 1>("direct (144 MFMA)", rnd, sink, stamps, steps, 4096)) return 1;
        if (run<1, 1>("F(2,3) along x (96 MFMA)", rnd, sink, stamps, steps, 4096)) return 1;
        if (run<2, 1>("F(2x2,3x3) (64 MFMA)", rnd, sink, stamps, steps, 4096)) return 1;
        if (run<0, 2>("direct (144 MFMA)", rnd, sink, stamps, steps, 4096)) return 1;
        if (run<1, 2>("F(2,3) along x (96 MFMA)", rnd, sink, stamps, steps, 4096)) return 1;
        if (run<2, 2>("F(2x2,3x3) (64 MFMA)", rnd, sink, stamps, steps, 4096)) return 1;
    }
    return 0;
}

// Shared host/device helpers of libdsvc (gfx950 only).
#pragma once
#ifndef DSVC_HOST_ONLY
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <stdio.h>
#include "../../include/dsvc.h"

#ifndef DSVC_HOST_ONLY
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// The 16-bit storage / MFMA operand type of the fast mode is IEEE binary16 (11-bit significand): with bfloat16 (8 bits) the
// decoded frames drifted 0.05-0.1 dB from fp32 mode over a GOP (measured) -- outside the 0.05 dB target.
typedef _Float16 h16;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;

#endif  // DSVC_HOST_ONLY
// ---- error plumbing (api.cpp) ----------------------------------------------------------------
void dsvc_set_error(const char* fmt, ...);
#define DSVC_REQUIRE(cond, ...)                         \
    do {                                                \
        if (!(cond)) {                                  \
            dsvc_set_error(__VA_ARGS__);                \
            return DSVC_E_ARG;                          \
        }                                               \
    } while (0)
#ifndef DSVC_HOST_ONLY
#define DSVC_CHECK_LAUNCH(what)                                                          \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) {                                                          \
            dsvc_set_error("%s: %s", what, hipGetErrorString(e_));                       \
            return DSVC_E_LAUNCH;                                                        \
        }                                                                                \
    } while (0)

#endif
// ---- kernel-selection switches of dsvc_conv2d (dsvc_set_option, host_misc.cpp) ------------------
struct DsvcOptions {
    int specialised, pp, ws, conv1x1, ws_waves4, ws_direct_store, tall_min_tiles, reg_epilogue, patch_prefetch, pp16, epi_f32, shuffle_vec, head, conv1x1_rows, small_pf, pps, pps_f32, conv1x1_f32, head_f32, skip_zero_k, pp_qc, pp_mix, pps_oneseg, headtap, cin8, rr7;
};
extern DsvcOptions g_dsvc_opt;

// ---- launch counters (dsvc_get_counter, host_misc.cpp): which conv kernel family took a launch ----------------
enum { DSVC_CNT_GENERIC = 0, DSVC_CNT_PP, DSVC_CNT_WS, DSVC_CNT_CONV1X1, DSVC_CNT_PP_POST, DSVC_CNT_HEAD, DSVC_CNT_PPS, DSVC_CNT_HEADTAP, DSVC_CNT_CIN8, DSVC_CNT_RR7, DSVC_CNT_N };
extern long long g_dsvc_counters[DSVC_CNT_N];
static inline void dsvc_count(int which) { __atomic_fetch_add(&g_dsvc_counters[which], 1, __ATOMIC_RELAXED); }

#ifndef DSVC_HOST_ONLY
// ---- f16 stores saturate ----------------------------------------------------------------------
// MODE.FP16_OVFL (hwreg MODE bit 23): an overflowed f16 conversion result is clamped to +-65504 instead of becoming +-inf
// (true infinities and NaNs pass through).  Every kernel that can store binary16 sets it first: a trained checkpoint whose
// activations exceed the f16 range then degrades gracefully -- encoder and decoder saturate identically, so streams stay
// decodable -- instead of poisoning the latents with inf / NaN (tests/test_gpu_ops.py::test_f16_stores_saturate).
#define DSVC_F16_SAT() __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1)

// ---- element access with runtime storage type ------------------------------------------------
__device__ __forceinline__ float h16_bits_to_f32(unsigned short b) {
    return (float)__builtin_bit_cast(h16, b);
}
__device__ __forceinline__ unsigned short f32_to_h16_bits(float f) {
    return __builtin_bit_cast(unsigned short, (h16)f);  // v_cvt_f16_f32: RNE; overflow saturates under DSVC_F16_SAT(), NaN stays NaN
}
// the two halves of a packed dword as floats / two floats -> one packed dword (low half = a; one v_cvt_pk_f16_f32, RNE)
__device__ __forceinline__ float h16_lo(unsigned u) { return (float)__builtin_bit_cast(h16x2, u).x; }
__device__ __forceinline__ float h16_hi(unsigned u) { return (float)__builtin_bit_cast(h16x2, u).y; }
__device__ __forceinline__ unsigned pk_h16(float a, float b) {
    h16x2 p;
    p.x = (h16)a;
    p.y = (h16)b;
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float load_elem(const void* p, int dtype, int64_t i) {
    return dtype == DSVC_F32 ? ((const float*)p)[i] : h16_bits_to_f32(((const unsigned short*)p)[i]);
}
__device__ __forceinline__ void store_elem(void* p, int dtype, int64_t i, float v) {
    if (dtype == DSVC_F32) ((float*)p)[i] = v;
    else ((unsigned short*)p)[i] = f32_to_h16_bits(v);
}
#endif  // DSVC_HOST_ONLY
static inline int dsvc_esize(int dtype) { return dtype == DSVC_F32 ? 4 : 2; }
#ifndef DSVC_HOST_ONLY

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
        case DSVC_ACT_RELU: return v > 0.f ? v : 0.f;
        case DSVC_ACT_LRELU: return v > 0.f ? v : v * slope;
        case DSVC_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        case DSVC_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case DSVC_ACT_HALF_TANH: return 0.5f * tanhf(v);
        default: return v;
    }
}
#endif  // DSVC_HOST_ONLY

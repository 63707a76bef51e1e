// host_misc.cpp -- host-side helpers of libdsvc: error plumbing, integer CDF construction,
// weight repacking for the MFMA conv kernel, transposed-conv -> sub-pixel conv rewrite.
#include <cstdarg>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.h"

static thread_local char g_err[512] = "";

void dsvc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dsvc_last_error(void) { return g_err; }
extern "C" int dsvc_version(void) { return DSVC_VERSION; }

static int env_flag(const char* name) { return getenv(name) != nullptr; }
DsvcOptions g_dsvc_opt = {
    /*specialised*/ !env_flag("DSVC_NO_WS"), /*pp*/ !env_flag("DSVC_NO_PP"), /*ws*/ 1, /*conv1x1*/ 1,
    /*ws_waves4*/ env_flag("DSVC_WS_WAVES4"), /*ws_direct_store*/ env_flag("DSVC_WS_DIRECT"),
    /*tall_min_tiles*/ getenv("DSVC_TALL_MIN_TILES") ? atoi(getenv("DSVC_TALL_MIN_TILES")) : 1024,
    /*reg_epilogue*/ !env_flag("DSVC_NO_REGEPI"), /*patch_prefetch*/ !env_flag("DSVC_NO_PATCHPF"), /*pp16*/ !env_flag("DSVC_NO_PP16"), /*epi_f32*/ !env_flag("DSVC_NO_EPI_F32"),
    /*shuffle_vec*/ !env_flag("DSVC_NO_SHUFFLE_VEC"), /*head*/ getenv("DSVC_HEAD") ? atoi(getenv("DSVC_HEAD")) : !env_flag("DSVC_NO_HEAD"), /*conv1x1_rows*/ !env_flag("DSVC_NO_1X1_ROWS"), /*small_pf*/ !env_flag("DSVC_NO_SMALL_PF"), /*pps*/ !env_flag("DSVC_NO_PPS"), /*pps_f32*/ !env_flag("DSVC_NO_PPS_F32"), /*conv1x1_f32*/ !env_flag("DSVC_NO_1X1_F32"), /*head_f32*/ !env_flag("DSVC_NO_HEAD_F32"), /*skip_zero_k*/ !env_flag("DSVC_NO_SKIP_ZERO_K"), /*pp_qc*/ env_flag("DSVC_PP_QC"), /*pp_mix*/ !env_flag("DSVC_NO_PP_MIX"), /*pps_oneseg*/ !env_flag("DSVC_NO_PPS_ONESEG"), /*headtap*/ !env_flag("DSVC_NO_HEADTAP"), /*cin8*/ !env_flag("DSVC_NO_CIN8"), /*rr7*/ !env_flag("DSVC_NO_RR7")};

extern "C" int dsvc_set_option(const char* key, int value) {
    struct { const char* k; int* v; } tab[] = {
        {"specialised", &g_dsvc_opt.specialised}, {"pp", &g_dsvc_opt.pp}, {"ws", &g_dsvc_opt.ws},
        {"conv1x1", &g_dsvc_opt.conv1x1}, 
        {"ws_waves4", &g_dsvc_opt.ws_waves4}, {"ws_direct_store", &g_dsvc_opt.ws_direct_store},
        {"tall_min_tiles", &g_dsvc_opt.tall_min_tiles}, {"reg_epilogue", &g_dsvc_opt.reg_epilogue},
        {"patch_prefetch", &g_dsvc_opt.patch_prefetch}, {"pp16", &g_dsvc_opt.pp16}, {"epi_f32", &g_dsvc_opt.epi_f32},
        {"shuffle_vec", &g_dsvc_opt.shuffle_vec}, {"head", &g_dsvc_opt.head}, {"conv1x1_rows", &g_dsvc_opt.conv1x1_rows}, {"small_pf", &g_dsvc_opt.small_pf}, {"pps", &g_dsvc_opt.pps}, {"pps_f32", &g_dsvc_opt.pps_f32}, {"conv1x1_f32", &g_dsvc_opt.conv1x1_f32}, {"head_f32", &g_dsvc_opt.head_f32}, {"skip_zero_k", &g_dsvc_opt.skip_zero_k}, {"pp_qc", &g_dsvc_opt.pp_qc}, {"pp_mix", &g_dsvc_opt.pp_mix}, {"pps_oneseg", &g_dsvc_opt.pps_oneseg}, {"headtap", &g_dsvc_opt.headtap}, {"cin8", &g_dsvc_opt.cin8}, {"rr7", &g_dsvc_opt.rr7}};
    for (auto& e : tab)
        if (key && strcmp(key, e.k) == 0) { const int old = *e.v; *e.v = value; return old; }
    dsvc_set_error("dsvc_set_option: unknown key '%s'", key ? key : "(null)");
    return DSVC_E_ARG;
}

long long g_dsvc_counters[DSVC_CNT_N] = {0};
extern long long g_dsvc_lut_overflow;      // rans_host.cpp
extern "C" int64_t dsvc_get_counter(const char* key) {
    if (key && strcmp(key, "lut_registry_overflow") == 0) return (int64_t)__atomic_load_n(&g_dsvc_lut_overflow, __ATOMIC_RELAXED);
    struct { const char* k; int i; } tab[] = {{"conv_generic", DSVC_CNT_GENERIC}, {"conv3x3_pp", DSVC_CNT_PP}, {"conv3x3_ws", DSVC_CNT_WS},
                                              {"conv1x1", DSVC_CNT_CONV1X1}, {"conv3x3_pp_fused_tail", DSVC_CNT_PP_POST}, {"conv_head", DSVC_CNT_HEAD}, {"conv3x3_pps", DSVC_CNT_PPS}, {"conv_headtap", DSVC_CNT_HEADTAP}, {"conv_cin8", DSVC_CNT_CIN8}, {"conv7x7_rr", DSVC_CNT_RR7}};
    for (auto& e : tab)
        if (key && strcmp(key, e.k) == 0) return (int64_t)__atomic_load_n(&g_dsvc_counters[e.i], __ATOMIC_RELAXED);
    dsvc_set_error("dsvc_get_counter: unknown key '%s'", key ? key : "(null)");
    return DSVC_E_ARG;
}

#ifndef DSVC_HOST_ONLY      // (DSVC_HOST_ONLY: the sanitizer build of the host functions, tests/native/ -- no HIP runtime, no device entry points)
extern "C" int dsvc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
#endif

// compressai _CXX.pmf_to_quantized_cdf (ported there from ryg_rans) -- SURVEY.md Appendix A.4;
// reached from the reference through update(): image_model.py:319-324, video_model.py:179-182.
extern "C" int dsvc_pmf_to_quantized_cdf(const float* pmf, int32_t n, int32_t precision, int32_t* cdf_out) {
    DSVC_REQUIRE(pmf && cdf_out && n > 0 && precision > 0 && precision <= 16, "pmf_to_quantized_cdf: bad argument");
    for (int i = 0; i < n; ++i)
        if (!(pmf[i] >= 0.f) || !std::isfinite(pmf[i])) { dsvc_set_error("pmf_to_quantized_cdf: invalid pmf[%d]", i); return DSVC_E_PMF; }
    std::vector<uint32_t> c((size_t)n + 1);
    c[0] = 0;
    uint32_t total = 0;
    for (int i = 0; i < n; ++i) { c[i + 1] = (uint32_t)std::round(pmf[i] * (float)(1 << precision)); total += c[i + 1]; }
    if (total == 0) { dsvc_set_error("pmf_to_quantized_cdf: pmf sums to zero"); return DSVC_E_PMF; }
    uint32_t run = 0;
    for (int i = 0; i <= n; ++i) { run += (uint32_t)((((uint64_t)1 << precision) * c[i]) / total); c[i] = run; }
    c[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (c[i] != c[i + 1]) continue;
        uint32_t best_freq = ~0u; int best = -1;   // steal from the rarest symbol that can spare one count
        for (int j = 0; j < n; ++j) {
            const uint32_t f = c[j + 1] - c[j];
            if (f > 1 && f < best_freq) { best_freq = f; best = j; }
        }
        if (best < 0) { dsvc_set_error("pmf_to_quantized_cdf: cannot make bin %d non-empty", i); return DSVC_E_PMF; }
        if (best < i) { for (int j = best + 1; j <= i; ++j) c[j]--; }
        else { for (int j = i + 1; j <= best; ++j) c[j]++; }
    }
    for (int i = 0; i <= n; ++i) cdf_out[i] = (int32_t)c[i];
    return DSVC_OK;
}

static inline uint16_t f32_to_h16_host(float f) {  // IEEE binary16, round to nearest even (same rounding as v_cvt_f16_f32)
    const _Float16 h = (_Float16)f;
    uint16_t u; std::memcpy(&u, &h, 2);
    return u;
}

extern "C" int64_t dsvc_conv_packed_bytes(int32_t Cout, int32_t nseg, const int32_t* segC, int32_t kh, int32_t kw, int32_t compute) {
    if (Cout <= 0 || nseg < 1 || nseg > 3 || !segC || kh < 1 || kw < 1) return DSVC_E_ARG;
    const int kce = compute == DSVC_F32 ? 16 : 32;
    int64_t nch = 0;
    for (int s = 0; s < nseg; ++s) { if (segC[s] <= 0) return DSVC_E_ARG; nch += (segC[s] + kce - 1) / kce; }
    const int64_t Np = (Cout + 31) / 32 * 32;
    return nch * kh * kw * Np * 64;
}

// OIHW fp32 -> [chunk][tap][Np][64 B] of the compute type; chunks never straddle input segments.
extern "C" int dsvc_conv_pack_weights(const float* w, int32_t Cout, int32_t nseg, const int32_t* segC,
                                      int32_t kh, int32_t kw, int32_t compute, void* packed) {
    DSVC_REQUIRE(w && packed && segC, "conv_pack_weights: null argument");
    DSVC_REQUIRE(Cout > 0 && nseg >= 1 && nseg <= 3 && kh >= 1 && kw >= 1, "conv_pack_weights: bad shape");
    DSVC_REQUIRE(compute == DSVC_F32 || compute == DSVC_F16, "conv_pack_weights: bad compute type");
    const int kce = compute == DSVC_F32 ? 16 : 32;
    const int64_t Np = (Cout + 31) / 32 * 32;
    const int taps = kh * kw;
    int Cin = 0;
    for (int s = 0; s < nseg; ++s) { DSVC_REQUIRE(segC[s] > 0, "conv_pack_weights: empty segment"); Cin += segC[s]; }
    const int64_t total = dsvc_conv_packed_bytes(Cout, nseg, segC, kh, kw, compute);
    std::memset(packed, 0, (size_t)total);
    int64_t chunk = 0;
    int cbase = 0;
    for (int s = 0; s < nseg; ++s) {
        const int nch = (segC[s] + kce - 1) / kce;
        for (int j = 0; j < nch; ++j, ++chunk) {
            for (int t = 0; t < taps; ++t) {
                for (int n = 0; n < Cout; ++n) {
                    char* row = (char*)packed + ((chunk * taps + t) * Np + n) * 64;
                    for (int e = 0; e < kce; ++e) {
                        const int cl = j * kce + e;
                        if (cl >= segC[s]) break;
                        const float v = w[((int64_t)n * Cin + (cbase + cl)) * taps + t];
                        if (compute == DSVC_F32) ((float*)row)[e] = v;
                        else ((uint16_t*)row)[e] = f32_to_h16_host(v);
                    }
                }
            }
        }
        cbase += segC[s];
    }
    return DSVC_OK;
}

// ConvTranspose2d(k5,s2,p2,op1): out[2Y+py][2X+px] = sum_{dy,dx in {-1,0,1}} in[Y+dy][X+dx] * Wt[py+2-2dy][px+2-2dx]
// (taps outside 0..4 vanish) -> a 3x3 pad-1 conv producing 4 phase channels per output channel.
extern "C" int dsvc_deconv5x5s2_to_conv3x3(const float* wt, const float* bias, int32_t Cin, int32_t Cout,
                                           float* wo, float* bias4) {
    DSVC_REQUIRE(wt && wo && Cin > 0 && Cout > 0, "deconv_to_conv: bad argument");
    std::memset(wo, 0, sizeof(float) * (size_t)4 * Cout * Cin * 9);
    for (int co = 0; co < Cout; ++co)
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                const int o = co * 4 + py * 2 + px;
                if (bias4) bias4[o] = bias ? bias[co] : 0.f;
                for (int dy = -1; dy <= 1; ++dy) {
                    const int ky = py + 2 - 2 * dy;
                    if (ky < 0 || ky > 4) continue;
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int kx = px + 2 - 2 * dx;
                        if (kx < 0 || kx > 4) continue;
                        for (int ci = 0; ci < Cin; ++ci)
                            wo[(((int64_t)o * Cin + ci) * 3 + (dy + 1)) * 3 + (dx + 1)] =
                                wt[(((int64_t)ci * Cout + co) * 5 + ky) * 5 + kx];
                    }
                }
            }
    return DSVC_OK;
}


#ifndef DSVC_HOST_ONLY
// ---- one scan step = a recorded launch sequence, issued without returning to the host language (include/dsvc.h) ----------------
extern "C" int dsvc_entropy_scan_step(void* stream, const dsvc_scan_op* ops, int32_t n_ops) {
    DSVC_REQUIRE(ops != nullptr && n_ops >= 0, "entropy_scan_step: null op list");
    for (int32_t i = 0; i < n_ops; ++i) {
        const dsvc_scan_op& o = ops[i];
        int rc = DSVC_E_ARG;
        switch (o.kind) {
            case DSVC_SCAN_CONV: rc = dsvc_conv2d(stream, &o.u.conv); break;
            case DSVC_SCAN_GC_QUANTIZE: {
                const dsvc_scan_gc& g = o.u.gc;
                rc = dsvc_gc_quantize(stream, g.y, g.y_ld, g.mu, g.mu_ld, g.scale, g.scale_ld, g.dtype, g.table, g.levels, g.B, g.H, g.W, g.C,
                                      g.symbols, g.indexes, g.y_hat, g.y_hat_ld);
                break;
            }
            case DSVC_SCAN_GC_DEQUANTIZE: {
                const dsvc_scan_gc& g = o.u.gc;
                rc = dsvc_gc_dequantize(stream, g.symbols, g.mu, g.mu_ld, g.dtype, g.B, g.H, g.W, g.C, g.y_hat, g.y_hat_ld);
                break;
            }
            case DSVC_SCAN_EB_PROCESS: {
                const dsvc_scan_eb& e = o.u.eb;
                rc = dsvc_eb_process(stream, e.z, e.z_ld, e.dtype, e.params, e.B, e.H, e.W, e.C, e.mode, e.symbols, e.z_hat, e.z_hat_ld, e.lik, e.lik_ld);
                break;
            }
            case DSVC_SCAN_EB_DEQUANTIZE: {
                const dsvc_scan_eb& e = o.u.eb;
                rc = dsvc_eb_dequantize(stream, e.symbols, e.params, e.dtype, e.B, e.H, e.W, e.C, e.z_hat, e.z_hat_ld);
                break;
            }
            case DSVC_SCAN_ELTWISE: {
                const dsvc_scan_ew& w = o.u.ew;
                auto opt = [](const dsvc_operand& x) { return x.ptr ? &x : nullptr; };
                rc = dsvc_eltwise(stream, w.op, w.n_pix, w.C, opt(w.a), opt(w.b), opt(w.c), opt(w.d), opt(w.e), w.out, w.out_ld, w.out_dtype);
                break;
            }
            case DSVC_SCAN_WARP: {
                const dsvc_scan_call& c = o.u.call;
                rc = dsvc_warp_bilinear_border(stream, c.p[0], c.i[0], c.i[1], (const float*)c.p[1], c.i[2], (void*)c.p[2], c.i[3], c.i[4], c.i[5], c.i[6], c.i[7], c.i[8]);
                break;
            }
            case DSVC_SCAN_UPSAMPLE2X: {
                const dsvc_scan_call& c = o.u.call;
                rc = dsvc_upsample2x_bilinear(stream, (const float*)c.p[0], c.i[0], (float*)c.p[1], c.i[1], c.i[2], c.i[3], c.i[4], c.i[5], c.f[0]);
                break;
            }
            case DSVC_SCAN_POOL2X2: {
                const dsvc_scan_call& c = o.u.call;
                rc = dsvc_pool2x2(stream, c.i[0], c.p[0], c.i[1], (void*)c.p[1], c.i[2], c.i[3], c.i[4], c.i[5], c.i[6], c.i[7]);
                break;
            }
            case DSVC_SCAN_PIXEL_SHUFFLE: {
                const dsvc_scan_call& c = o.u.call;
                rc = dsvc_pixel_shuffle_nchw(stream, (const float*)c.p[0], (void*)c.p[1], c.i[0], c.i[1], c.i[2], c.i[3], c.i[4], c.i[5], c.i[6]);
                break;
            }
            case DSVC_SCAN_NCHW_TO_NHWC: {
                const dsvc_scan_call& c = o.u.call;
                rc = dsvc_nchw_to_nhwc(stream, (const float*)c.p[0], (void*)c.p[1], c.i[0], c.i[1], c.i[2], c.i[3], c.i[4], c.i[5]);
                break;
            }
            case DSVC_SCAN_NHWC_TO_NCHW: {
                const dsvc_scan_call& c = o.u.call;
                rc = dsvc_nhwc_to_nchw(stream, c.p[0], c.i[0], c.i[1], (float*)c.p[1], c.i[2], c.i[3], c.i[4], c.i[5]);
                break;
            }
            case DSVC_SCAN_SE_GATE: {
                const dsvc_scan_call& c = o.u.call;
                rc = dsvc_se_gate(stream, c.p[0], c.i[0], c.i[1], c.i[2], c.i[3], c.i[4], c.i[5], (const float*)c.p[1], (const float*)c.p[2], c.i[6],
                                  (float*)c.p[3], (float*)c.p[4]);
                break;
            }
            default:
                dsvc_set_error("entropy_scan_step: op %d has unknown kind %d", i, o.kind);
                return DSVC_E_ARG;
        }
        if (rc != DSVC_OK) {
            char inner[400];
            snprintf(inner, sizeof(inner), "%s", dsvc_last_error());
            dsvc_set_error("entropy_scan_step: op %d of %d (kind %d) failed: %s", i, n_ops, o.kind, inner);
            return rc;
        }
    }
    return DSVC_OK;
}

extern "C" int dsvc_pframe_segment_run(void* stream, const dsvc_scan_op* ops, int32_t n_ops) {
    return dsvc_entropy_scan_step(stream, ops, n_ops);       // same interpreter: a frame segment is a longer op list
}
#endif  // DSVC_HOST_ONLY

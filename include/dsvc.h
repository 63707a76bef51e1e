/* dsvc.h -- C ABI of the MI355X-native DeepSVC P-frame hot path (libdsvc.so).
 *
 * Boundary contract:
 *   - plain pointers and sizes only; no torch / C++ types cross this interface;
 *   - every device pointer is caller-owned HBM (PyTorch owns it in the shipped host layer);
 *     the library allocates nothing the caller must free except rANS coder handles;
 *   - every function returns 0 on success or a negative DSVC_E_* code; no exception crosses
 *     the boundary; dsvc_last_error() returns a thread-local message;
 *   - device functions are stream-ordered on `stream` (a hipStream_t passed as void*) and do
 *     not synchronise; host functions (rANS, CDF, packing) are synchronous CPU code.
 *
 * Activations are NHWC ("pixel-major"): element (b,y,x,c) of a tensor lives at
 *   ptr[((b*H + y)*W + x)*ld + c],  ld >= C  (ld > C addresses a channel slice of a wider
 * buffer: this is how torch.cat / chunk of the reference become zero-copy here).
 *
 * Each entry point names the reference interface it replaces (file:line in LHB116/DeepSVC).
 */
#ifndef DSVC_H
#define DSVC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DSVC_VERSION 3   /* 3: frame-level program API (dsvc_pframe_program_*, dsvc_pframe_compress / _decompress), dsvc_cdf_forget_lut; 2: binary16 (not bfloat16) is dtype code 1; scan ops, refine_aggregate, counters; f16 stores saturate */

/* error codes */
#define DSVC_OK 0
#define DSVC_E_ARG (-1)      /* bad argument (shape, alignment, enum) */
#define DSVC_E_LAUNCH (-2)   /* HIP launch / runtime error */
#define DSVC_E_RANGE (-3)    /* output buffer too small / stream exhausted */
#define DSVC_E_PMF (-4)      /* invalid pmf (negative, NaN, zero total) */
#define DSVC_E_NODEV (-5)    /* no HIP device */

/* element types of activation storage / MFMA compute */
#define DSVC_F32 0
#define DSVC_F16 1      /* IEEE binary16 storage / MFMA operands (fp32 accumulation) */

/* fused activations */
#define DSVC_ACT_NONE 0
#define DSVC_ACT_RELU 1      /* nn.ReLU                       modules.py:125,180,198 */
#define DSVC_ACT_LRELU 2     /* nn.LeakyReLU(slope)           modules.py:179,196,238 */
#define DSVC_ACT_GELU 3      /* nn.GELU() exact erf           image_model.py:86-134  */
#define DSVC_ACT_SIGMOID 4   /* nn.Sigmoid                    modules.py:333,340,416 */
#define DSVC_ACT_HALF_TANH 5 /* 0.5*tanh(x)                   image_model.py:187,246,294 */

const char* dsvc_last_error(void);
int dsvc_version(void);
/* number of HIP devices visible (0 when none / no driver); never throws */
int dsvc_device_count(void);
/* Kernel-selection switches of dsvc_conv2d, for A/B measurements and tests from one process (no reference
 * counterpart: the reference has one cuDNN path).  Keys: "specialised" (1: conv1x1 / conv3x3_ws / conv3x3_pp /
 * may take a launch, 0: generic kernel only), "pp", "ws", "conv1x1" (each 0/1),
 * "ws_waves4", "ws_direct_store", "tall_min_tiles" (the generic kernel uses 16x32-pixel tiles for f16
 * stride-1 convs with 64 output channels per group from this many tiles on; a huge value turns them off), "reg_epilogue" (1: the
 * generic kernel's LDS-free epilogue where it applies), "patch_prefetch" (1: ... with the next chunk's input patch prefetched),
 * "pp16" (the two-group 3x3 kernel's MFMA shape: 0 = 32x32x16, 2 = 16x16x32, 1 = per variant whichever measured faster),
 * "epi_f32" (1: the vector epilogue issues only fp32 loads / stores when output and residuals are fp32), "shuffle_vec" (1: PixelShuffle(2)
 * outputs are stored as 4-channel vectors of one output pixel; 0: element by element), "head" (narrow f16 output heads -- Cout <= 4,
 * 3x3 over 32 / 64 / 64+64 channels or 7x7 over 16 -- of large images as packed dot products on the vector ALU: 1 = the shapes where that
 * measured faster than the MFMA kernels, 2 = every covered shape, 0 = none), "conv1x1_rows" (1: the pointwise kernel reads and writes
 * whole pixel rows and transposes to the MFMA operand layout through LDS; 0: lanes address 32-byte pieces of 32 rows), "small_pf" (1: the
 * generic kernel's 4x32 / 8x32 x 32-channel variants prefetch the next channel chunk's input patch: small images, f16 storage), "pps" (1: f16 3x3
 * stride-1 convs over 3+ 32-channel input chunks with 64 output channels on >= 1024 tiles take the two-group kernel with weights streamed per
 * chunk, csrc/conv3x3_pps.hip -- also with a pre-activation or one f16 residual, and 128 -> 128 channels as two 64-column passes inside one
 * launch; 2: also 96 / 128 output channels over 2+ chunks, measured no faster; 0: the general kernel), "pps_f32" (1: ... and so do fp32 3x3 stride-1 convs with 64 output channels over 3+ 16-channel
 * chunks -- the parity mode's 64 -> 64 family, any pre-activation / activation / fp32 residuals; bit-identical to the general kernel), "conv1x1_f32"
 * (1: fp32 1x1 convs over 64 / 96 / 128 channels of images of 16 K+ pixels take the streaming kernel csrc/conv1x1_f32.hip; bit-identical to the general kernel), "head_f32" (1: fp32 3x3 / 7x7 stride-1 convs with <= 4 output channels over whole 16-channel chunks of
 * images of 256+ tiles run as the MFMA's own fma chain on the vector ALU, csrc/conv_head_f32.hip; bit-identical to the general kernel), "skip_zero_k" (1: fp32 launches with 2- / 3-channel segments skip the MFMAs
 * whose k slots hold only padding; bit-identical), "headtap" (1: f16 3x3 stride-1 convs with <= 3 output channels over one or two segments of 32 or 64
 * channels on 256+ tiles run as a per-input-pixel GEMM over (tap, output) columns on the matrix pipe followed by a 9-point gather-sum,
 * csrc/conv_headtap.hip: tried before "head"; 0: "head" and the MFMA kernels decide), "cin8" (1: f16 3x3 -> 32 / 64 and 7x7 -> 32 stride-1 convs over ONE
 * segment of exactly 8 channels with contiguous 16-byte pixels, f16 output, no residual, activation none / ReLU / LeakyReLU, on 64+ tiles pack the
 * x-taps of a kernel row into the MFMA's K axis, csrc/conv_cin8.hip; 0: the other kernels pad each tap to a 32-channel chunk), "rr7" (1: f16 7x7
 * stride-1 convs 32 -> 16 / 32 / 64 and 64 -> 32 over one segment, f16 output, no residual, activation none / ReLU / LeakyReLU, on 128+ tiles of
 * 16x32 pixels: each wave owns four output rows and reuses a pixel fragment across the kernel rows in registers, weights straight from global
 * memory, csrc/conv7x7_rr.hip -- 32 -> 16 on the 16x16x32 MFMA shape, without padded weight rows; 2: 32 -> 16 on the 32-row tile like the others (A/B); 0: the general kernel).  Defaults come from
 * the environment at load time (DSVC_NO_WS, DSVC_NO_PP, DSVC_WS_WAVES4, DSVC_WS_DIRECT,
 * DSVC_TALL_MIN_TILES, DSVC_NO_REGEPI, DSVC_NO_PATCHPF, DSVC_NO_PP16, DSVC_NO_EPI_F32, DSVC_NO_SHUFFLE_VEC, DSVC_NO_HEAD / DSVC_HEAD, DSVC_NO_1X1_ROWS, DSVC_NO_SMALL_PF, DSVC_NO_PPS, DSVC_NO_PPS_F32, DSVC_NO_1X1_F32, DSVC_NO_HEAD_F32, DSVC_NO_SKIP_ZERO_K, DSVC_NO_HEADTAP, DSVC_NO_CIN8, DSVC_NO_RR7).
 * Returns the previous value, or DSVC_E_ARG for an unknown key.  Results are identical whichever kernel runs in
 * fp32 mode; in f16 mode they agree within the f16 tolerance of the tests. */
int dsvc_set_option(const char* key, int value);
/* Diagnostic: how many dsvc_conv2d launches each kernel family has taken since the library was loaded (process-wide).
 * Keys: "conv_generic", "conv3x3_pp", "conv3x3_pp_fused_tail", "conv3x3_ws", "conv1x1", "conv_head", "conv3x3_pps", "conv_headtap", "conv_cin8", "conv7x7_rr"; "lut_registry_overflow" (see dsvc_cdf_build_lut).  Tests use it to prove that the
 * large-image kernels ran inside a frame that is checked against the reference.  Returns DSVC_E_ARG for an unknown key. */
int64_t dsvc_get_counter(const char* key);
/* Diagnostic: time every dsvc_conv2d launch of ONE shape (single-segment k x k conv, `stride`, C -> Cout channels on an H x W input) with
 * HIP events recorded on the launch stream, wherever the launch comes from (direct call, scan step, frame segment).
 * dsvc_profile_begin arms it (dropping earlier records); dsvc_profile_end disarms it, waits for the events and writes up to `cap`
 * durations in ms (and the number of residual operands of each launch); returns the number of records (ms_out == NULL: count only,
 * records kept).  Process-wide; used by bench.py's roofline leg. */
int dsvc_profile_begin(int32_t kh, int32_t stride, int32_t C, int32_t Cout, int32_t H, int32_t W);
int64_t dsvc_profile_end(float* ms_out, int32_t* nres_out, int64_t cap);

/* ------------------------------------------------------------------------------------------
 * Convolution family (one MFMA implicit-GEMM kernel, fused epilogue).
 * Replaces nn.Conv2d / nn.ConvTranspose2d calls of modules.py:127-131,176-178,200-204,237-242,
 * 253,307-311,322-340,364,382-418 and image_model.py:25-27,53-146 (compressai conv/deconv/
 * conv3x3/subpel_conv3x3), together with the pointwise op that follows each of them.
 * ------------------------------------------------------------------------------------------ */
typedef struct dsvc_seg {
    const void* ptr; /* channel 0 of pixel (0,0) of image 0 of this input segment */
    int32_t C;       /* channels taken from this segment */
    int32_t ld;      /* elements between consecutive pixels */
    int32_t dtype;   /* DSVC_F32 / DSVC_F16 */
} dsvc_seg;

typedef struct dsvc_conv_desc {
    int32_t B, H, W;      /* input batch / height / width (shared by all segments) */
    int32_t nseg;         /* 1..3: input = channel-concatenation of the segments (torch.cat) */
    dsvc_seg in[3];
    int32_t kh, kw, stride, pad;
    int32_t Cout;         /* conv output channels (before depth-to-space) */
    int32_t compute;      /* DSVC_F32: v_mfma_f32_32x32x2_f32; DSVC_F16: v_mfma_f32_32x32x16_f16 */
    const void* weight;   /* device buffer produced by dsvc_conv_pack_weights for this compute type */
    const float* bias;    /* device [Cout] fp32, or NULL */
    int32_t pre_act;      /* activation applied to the INPUT while staging (ResBlock1 first_layer) */
    int32_t act;          /* activation applied to conv+bias */
    float slope;          /* LeakyReLU slope for pre_act/act */
    /* out = act(conv+bias) + res1*res1_scale[b][c] + res2   (each optional) */
    const void* res1; int32_t res1_ld, res1_dtype;
    const float* res1_scale; /* device [B][Cout] fp32 or NULL (SELayer gate, modules.py:226-230) */
    const void* res2; int32_t res2_ld, res2_dtype;
    int32_t shuffle;      /* 1, or r=2: nn.PixelShuffle(2) folded into the store (Cout % 4 == 0) */
    void* out; int32_t out_ld, out_dtype;
    /* s2d = 1: the inputs are read through a SPACE-TO-DEPTH view: B, H, W describe the view grid, every view pixel (Y, X)
     * is the 2x2 block of pixels (2Y+py, 2X+px) of a (2H x 2W) NHWC image with row pitch in[s].ld, and view channel
     * g*4*V + (2*py+px)*V + c is channel g*V + c of that pixel (V = 8 for f16 compute, 4 for fp32: one 16-byte vector).
     * in[s].C counts VIEW channels (4x the image's).  Requires kh = kw = 3, stride 1, pad 1, input storage == compute type,
     * 16-byte aligned rows.  This is how conv5x5/s2 (compressai conv(), image_model.py:53-66,344-358,376-382) runs: the
     * 5x5 stride-2 kernel is exactly a 3x3 stride-1 kernel over this view (weights remapped once on the host), which
     * avoids both the 25-tap weight re-staging per tile and the padding of 8-channel segments to 32-channel chunks. */
    int32_t s2d;
    /* Optional pointwise conv fused behind the activation (ResBottleneckBlock's conv2 -> LeakyReLU -> conv3 -> + x,
     * modules.py:173-189 == image_model.py:22-38):
     *     out = post_weight * act(conv + bias) + post_bias + res1
     * post_weight: packed weights (dsvc_conv_pack_weights) of a 1x1 conv Cout -> Cout, post_bias: device [Cout] fp32 or NULL.
     * The intermediate act(conv + bias) is rounded to the compute type exactly as if it had been stored and re-read.
     * Supported where the two-group 3x3 kernel applies (f16, 64 -> 64 channels, >= 1024 tiles, f16 NHWC in/out/res1,
     * no res2 / gate / shuffle); dsvc_conv2d fails with DSVC_E_ARG otherwise -- callers fuse only then. */
    const void* post_weight; const float* post_bias;
    /* batch_inner > 0: the B batch items are INDEPENDENT convs of identical shape in one launch (the slice scan runs the
     * mean and the scale stack of a slice -- image_model.py:168-176,227-237 -- side by side).  Item b uses the packed
     * weights at weight + b * weight_bstride bytes and the bias at bias + b * bias_bstride floats, and reads input segment s at
     *     in[s].ptr + (b % batch_inner) * in_bstride[s][0] + (b / batch_inner) * in_bstride[s][1]      (bytes; 0 = shared by the items)
     * instead of the dense batch layout.  Outputs and residuals keep the dense [B, H, W, ld] layout.  Needs the vector
     * staging path (16-byte aligned rows); excludes s2d, post_weight and res1_scale.  0 = an ordinary batch. */
    int32_t batch_inner;
    int64_t weight_bstride, bias_bstride;
    int64_t in_bstride[3][2];
    /* batch_inner > 0 only: out_bstride > 0 places item b's output (and residual res1, with res1_bstride) at out + b * out_bstride
     * bytes as ONE image [H, W, ld] -- e.g. consecutive channel slices of one buffer: the latent-residual-prediction stacks of a scan
     * level write y_hat slice after slice in place (image_model.py:185-188,244-247).  0 = the dense [B, H, W, ld] layout. */
    int64_t out_bstride, res1_bstride;
} dsvc_conv_desc;

/* bytes of the packed weight buffer for a conv with these input segments */
int64_t dsvc_conv_packed_bytes(int32_t Cout, int32_t nseg, const int32_t* segC, int32_t kh, int32_t kw, int32_t compute);
/* HOST: OIHW fp32 weights -> packed [chunk][tap][Np][64B] of the compute type (host memory) */
int dsvc_conv_pack_weights(const float* w_oihw, int32_t Cout, int32_t nseg, const int32_t* segC,
                           int32_t kh, int32_t kw, int32_t compute, void* packed_host);
/* HOST: ConvTranspose2d(k=5,s=2,p=2,op=1) weight [Cin][Cout][5][5] (image_model.py:69-81) ->
 * equivalent 3x3 stride-1 pad-1 conv weight [4*Cout][Cin][3][3] whose output, depth-to-space'd by 2
 * (channel co*4 + py*2 + px), equals the transposed conv.  bias4 (if given) receives bias repeated. */
int dsvc_deconv5x5s2_to_conv3x3(const float* w_iohw, const float* bias, int32_t Cin, int32_t Cout,
                                float* w_oihw_out, float* bias4_out);
int dsvc_conv2d(void* stream, const dsvc_conv_desc* d);

/* ------------------------------------------------------------------------------------------
 * Resampling / layout kernels (HBM-bound)
 * ------------------------------------------------------------------------------------------ */
/* torch_warp (modules.py:25-62): out(b,y,x,:) = bilinear(in, x+flow[..,0], y+flow[..,1]), border clamp.
 * flow is fp32 NHWC with 2 channels. */
int dsvc_warp_bilinear_border(void* stream, const void* in, int32_t in_ld, int32_t in_dtype,
                              const float* flow, int32_t flow_ld, void* out, int32_t out_ld, int32_t out_dtype,
                              int32_t B, int32_t H, int32_t W, int32_t C);
/* bilinearupsacling (modules.py:107-112) followed by "* scale" (modules.py:163), fp32 */
int dsvc_upsample2x_bilinear(void* stream, const float* in, int32_t in_ld, float* out, int32_t out_ld,
                             int32_t B, int32_t H, int32_t W, int32_t C, float scale);
/* F.avg_pool2d(2,2) (modules.py:156-157) / nn.MaxPool2d(2,2) (modules.py:260): mode 0 = avg, 1 = max */
int dsvc_pool2x2(void* stream, int32_t mode, const void* in, int32_t in_ld, void* out, int32_t out_ld,
                 int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C);
/* nn.PixelShuffle(r) on an NCHW fp32 source producing NHWC (InterLayerPrediction.d2s, modules.py:391) */
int dsvc_pixel_shuffle_nchw(void* stream, const float* in_nchw, void* out, int32_t out_ld, int32_t out_dtype,
                            int32_t B, int32_t Cout, int32_t Hin, int32_t Win, int32_t r);
/* layout conversion at the API boundary (reference tensors are NCHW fp32) */
int dsvc_nchw_to_nhwc(void* stream, const float* in, void* out, int32_t out_ld, int32_t out_dtype,
                      int32_t B, int32_t C, int32_t H, int32_t W);
int dsvc_nhwc_to_nchw(void* stream, const void* in, int32_t in_ld, int32_t in_dtype, float* out,
                      int32_t B, int32_t C, int32_t H, int32_t W);

/* ------------------------------------------------------------------------------------------
 * Pointwise combinations (HBM-bound).  Operands are (ptr, ld, dtype) NHWC slices over n_pix pixels.
 * ------------------------------------------------------------------------------------------ */
#define DSVC_EW_COPY 0    /* out = a                                  (torch.cat fall-back)        */
#define DSVC_EW_ADD 1     /* out = a + b                              video_model.py:62,164        */
#define DSVC_EW_SUB 2     /* out = a - b                              video_model.py:50,148        */
#define DSVC_EW_BLEND2 3  /* out = w*a + (1-w)*b                      modules.py:436               */
#define DSVC_EW_BLEND3 4  /* out = w*a + v*b + (1-w-v)*c  (w=d, v=e)  modules.py:353               */
#define DSVC_EW_CLAMP01 5 /* out = clamp(a,0,1)                       video_model.py:167           */
typedef struct dsvc_operand { const void* ptr; int32_t ld; int32_t dtype; } dsvc_operand;
int dsvc_eltwise(void* stream, int32_t op, int64_t n_pix, int32_t C, const dsvc_operand* a,
                 const dsvc_operand* b, const dsvc_operand* c, const dsvc_operand* d, const dsvc_operand* e,
                 void* out, int32_t out_ld, int32_t out_dtype);

/* SELayer (modules.py:217-230): gate[b][c] = sigmoid(W2 relu(W1 mean_hw(x))).  partial is a caller
 * workspace of at least dsvc_se_workspace_floats(B,C) floats.  Deterministic (no atomics). */
int64_t dsvc_se_workspace_floats(int32_t B, int32_t C);
int dsvc_se_gate(void* stream, const void* x, int32_t ld, int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C,
                 const float* w1 /*[C/r][C]*/, const float* w2 /*[C][C/r]*/, int32_t hidden,
                 float* partial, float* gate /*[B][C]*/);

/* RefineNet1's temporal aggregation (semantic_layer.py:1166-1184), fp32 NHWC, one image x against L
 * support images: per pixel n(v) = v/||v||_2 over C channels, a_l = mean_c (mean != 0) or sum_c of n(rx)*n(rs_l),
 * w = softmax_l(a), out = x + sum_l x*w_l, then ReLU when relu != 0 (semantic_layer.py:1082).  rs: [L][n_pix][rs_ld].
 * C in {64, 128, 192, 256}, L <= 16.  Deterministic. */
int dsvc_refine_aggregate(void* stream, const float* x, int32_t x_ld, const float* rx, int32_t rx_ld, const float* rs, int32_t rs_ld,
                          int32_t L, int64_t n_pix, int32_t C, int32_t mean, int32_t relu, float* out, int32_t out_ld);

/* deterministic reductions into out[0] (fp64 accumulate in the final stage); partial: workspace of
 * dsvc_reduce_workspace_floats() floats.
 * mode 0: sum(a)   1: sum(log(a))   2: sum((a-b)^2)   (video_model.py:38-42,46,53-56,68) */
int64_t dsvc_reduce_workspace_floats(void);
int dsvc_reduce(void* stream, int32_t mode, int64_t n_pix, int32_t C, const dsvc_operand* a, const dsvc_operand* b,
                float* partial, double* out);

/* One level of MS-SSIM with pytorch-msssim 0.2.1 semantics (third-party, absent from the reference tree; call sites
 * video_model.py:107,114,132 and test_video.py:311,359-363: ms_ssim(x, y, data_range=1.0)):
 * out_ssim[b*C+c], out_cs[b*C+c] = mean over the VALID positions of the separable 11-tap Gaussian window (sigma 1.5)
 * of the ssim / contrast-structure maps, K = (0.01, 0.03).  x, y: NHWC slices of either storage type, H, W >= 11.
 * workspace: dsvc_ssim_workspace_floats(B,H,W,C) floats.  Deterministic (no atomics, fp64 final sum).
 * dsvc_avgpool2_pad: F.avg_pool2d(x, 2, padding=(H%2, W%2)) between the levels (zeros count, divide by 4) into a dense
 * fp32 NHWC tensor of size B x ((H+2(H%2)-2)/2+1) x ((W+2(W%2)-2)/2+1) x C. */
int64_t dsvc_ssim_workspace_floats(int32_t B, int32_t H, int32_t W, int32_t C);
int dsvc_ssim_level(void* stream, const dsvc_operand* x, const dsvc_operand* y, int32_t B, int32_t H, int32_t W, int32_t C,
                    float data_range, float* workspace, float* out_ssim, float* out_cs);
int dsvc_avgpool2_pad(void* stream, const dsvc_operand* x, float* out, int32_t B, int32_t H, int32_t W, int32_t C);

/* ------------------------------------------------------------------------------------------
 * Entropy-model arithmetic on device (compressai 1.2.1 semantics)
 * Symbol / index planes are int32 in the reference's NCHW flattening order [C][h][w]
 * (image_model.py:241-242: y_q_slice.reshape(-1)).
 * ------------------------------------------------------------------------------------------ */
/* GaussianConditional.quantize(y,"symbols",mu) + build_indexes(scale) + y_hat = q + mu
 * (image_model.py:237-239).  y/mu/scale/y_hat are NHWC slices; table = 64 fp32 scale levels (device).
 * symbols/indexes/y_hat may each be NULL to skip. */
int dsvc_gc_quantize(void* stream, const void* y, int32_t y_ld, const void* mu, int32_t mu_ld,
                     const void* scale, int32_t scale_ld, int32_t dtype, const float* table, int32_t levels,
                     int32_t B, int32_t H, int32_t W, int32_t C,
                     int32_t* symbols, int32_t* indexes, void* y_hat, int32_t y_hat_ld);
/* GaussianConditional.dequantize (image_model.py:290): y_hat = float(symbols) + mu */
int dsvc_gc_dequantize(void* stream, const int32_t* symbols, const void* mu, int32_t mu_ld, int32_t dtype,
                       int32_t B, int32_t H, int32_t W, int32_t C, void* y_hat, int32_t y_hat_ld);
/* GaussianConditional.forward eval (image_model.py:181): y_hat = round(y-mu)+mu,
 * lik = max(Phi((.5-|v|)/s) - Phi((-.5-|v|)/s), 1e-9), s = max(scale, 0.11); lik fp32 NHWC */
int dsvc_gc_likelihood(void* stream, const void* y, int32_t y_ld, const void* mu, int32_t mu_ld,
                       const void* scale, int32_t scale_ld, int32_t dtype, int64_t n_pix, int32_t C,
                       void* y_hat, int32_t y_hat_ld, float* lik, int32_t lik_ld);
/* EntropyBottleneck (image_model.py:155-162,206-207,260).  params: device fp32 blob per channel
 * [C][59] = matrices(3,9,9,9,3) biases(3,3,3,3,1) factors(3,3,3,3) median(1), already softplus'd /
 * tanh'd on the host.  mode bit0: write symbols (int32 planar [C][h][w]); bit1: write z_hat;
 * bit2: write likelihood (fp32 NHWC). */
#define DSVC_EB_PARAMS_PER_CH 59
int dsvc_eb_process(void* stream, const void* z, int32_t z_ld, int32_t dtype, const float* params,
                    int32_t B, int32_t H, int32_t W, int32_t C, int32_t mode,
                    int32_t* symbols, void* z_hat, int32_t z_hat_ld, float* lik, int32_t lik_ld);
/* z_hat = float(symbols) + median  (EntropyBottleneck.decompress -> dequantize) */
int dsvc_eb_dequantize(void* stream, const int32_t* symbols, const float* params, int32_t dtype,
                       int32_t B, int32_t H, int32_t W, int32_t C, void* z_hat, int32_t z_hat_ld);

/* One step of the channel-conditional entropy scan as ONE call (dsvc_entropy_scan_step; reference loops
 * image_model.py:226-251 (encoder) and :276-297 (decoder), whose every iteration returns to Python -- and to the host with
 * .tolist() -- between its ~11 small convs).  A step is the fixed launch sequence of one dependency level of the scan over a
 * codec's static workspace: the parameter nets (cc_mean || cc_scale of every slice of the level as batch items of one conv per
 * layer), build_indexes (+ quantise on the encoder side / dequantise on the decoder side) and the latent-residual-prediction
 * stacks -- or the whole encoder scan, hyper-analysis included, since the encoder needs no host round trip at all.
 * The caller describes the sequence ONCE as an array of ops (pointers into buffers it keeps alive) and replays it per frame;
 * the library issues the launches back to back on `stream` (stream-ordered, no host synchronisation, capturable into a HIP
 * graph).  Results are those of issuing the same ops through dsvc_conv2d / dsvc_gc_* / dsvc_eb_* one by one.
 * On failure returns the failing op's error code; dsvc_last_error() names the op index. */
#define DSVC_SCAN_CONV 0            /* u.conv: dsvc_conv2d */
#define DSVC_SCAN_GC_QUANTIZE 1     /* u.gc:   dsvc_gc_quantize (symbols / indexes / y_hat each optional) */
#define DSVC_SCAN_GC_DEQUANTIZE 2   /* u.gc:   dsvc_gc_dequantize (symbols, mu -> y_hat) */
#define DSVC_SCAN_EB_PROCESS 3      /* u.eb:   dsvc_eb_process */
#define DSVC_SCAN_EB_DEQUANTIZE 4   /* u.eb:   dsvc_eb_dequantize */
#define DSVC_SCAN_ELTWISE 5         /* u.ew:   dsvc_eltwise (f16 mode packs small fp32 support slices for the conv's fast staging path) */
typedef struct dsvc_scan_gc {
    const void* y; const void* mu; const void* scale;
    int32_t y_ld, mu_ld, scale_ld, dtype;
    const float* table; int32_t levels;
    int32_t B, H, W, C;
    int32_t* symbols; int32_t* indexes;
    void* y_hat; int32_t y_hat_ld;
} dsvc_scan_gc;
typedef struct dsvc_scan_eb {
    const void* z; int32_t z_ld, dtype;
    const float* params;
    int32_t B, H, W, C, mode;
    int32_t* symbols;
    void* z_hat; int32_t z_hat_ld;
    float* lik; int32_t lik_ld;
} dsvc_scan_eb;
typedef struct dsvc_scan_ew {
    int32_t op, C; int64_t n_pix;
    dsvc_operand a, b, c, d, e;      /* an absent operand has ptr == NULL */
    void* out; int32_t out_ld, out_dtype;
} dsvc_scan_ew;
/* op kinds beyond the entropy scan: what a whole P-frame segment needs besides convs (frame level, below) */
#define DSVC_SCAN_WARP 6            /* u.call: dsvc_warp_bilinear_border  p = {in, flow, out}     i = {in_ld, in_dtype, flow_ld, out_ld, out_dtype, B, H, W, C} */
#define DSVC_SCAN_UPSAMPLE2X 7      /* u.call: dsvc_upsample2x_bilinear   p = {in, out}           i = {in_ld, out_ld, B, H, W, C}   f = {scale} */
#define DSVC_SCAN_POOL2X2 8         /* u.call: dsvc_pool2x2               p = {in, out}           i = {mode, in_ld, out_ld, dtype, B, H, W, C} */
#define DSVC_SCAN_PIXEL_SHUFFLE 9   /* u.call: dsvc_pixel_shuffle_nchw    p = {in_nchw, out}      i = {out_ld, out_dtype, B, Cout, Hin, Win, r} */
#define DSVC_SCAN_NCHW_TO_NHWC 10   /* u.call: dsvc_nchw_to_nhwc          p = {in, out}           i = {out_ld, out_dtype, B, C, H, W} */
#define DSVC_SCAN_NHWC_TO_NCHW 11   /* u.call: dsvc_nhwc_to_nchw          p = {in, out}           i = {in_ld, in_dtype, B, C, H, W} */
#define DSVC_SCAN_SE_GATE 12        /* u.call: dsvc_se_gate               p = {x, w1, w2, partial, gate}  i = {ld, dtype, B, H, W, C, hidden} */
typedef struct dsvc_scan_call {     /* argument block of the pointwise / layout ops above */
    const void* p[6];
    int32_t i[10];
    float f[2];
} dsvc_scan_call;
typedef struct dsvc_scan_op {
    int32_t kind;           /* DSVC_SCAN_* */
    int32_t reserved;
    union { dsvc_conv_desc conv; dsvc_scan_gc gc; dsvc_scan_eb eb; dsvc_scan_ew ew; dsvc_scan_call call; } u;
} dsvc_scan_op;
int dsvc_entropy_scan_step(void* stream, const dsvc_scan_op* ops, int32_t n_ops);

/* Frame level (SURVEY.md section 8b "frame level (later)": dsvc_pframe_compress / dsvc_pframe_decompress).
 * The reference's unit of work is one call -- DeepSVC.compress (video_model.py:137-151) / DeepSVC.decompress (:153-167) -- behind
 * which PyTorch issues ~400 kernels one by one.  Here a P-frame is a handful of host-synchronisation-free SEGMENTS (encoder:
 * motion estimation + motion analysis + scan | synthesis + prediction + feature extraction + residual analysis + scan; decoder:
 * motion synthesis + prediction | residual synthesis | reconstruction; the range coder runs on the host between them, by format),
 * each recorded ONCE as an array of dsvc_scan_op over (a) the caller's frame tensors, (b) one workspace arena in which the
 * segment's temporaries are packed by lifetime, (c) the model's packed weights; per frame the host layer patches the pointers of
 * the frame tensors into the array and makes ONE call per segment.  The library issues the segment's launches back to back on
 * `stream` (stream-ordered, no host synchronisation); results are those of the same ops issued one by one. */
int dsvc_pframe_segment_run(void* stream, const dsvc_scan_op* ops, int32_t n_ops);

/* The same boundary for a caller WITHOUT the Python layer: one call per P-frame, as the reference's harness makes them
 * (test_video.py:342 `mv_out_enc, res_out_enc = p_model.compress(ref_frame, x_cur, sm[0], feature)` and :349
 * `feature, x_hat, warped, predict = p_model.decompress(ref_frame, mv_out_enc, res_out_enc, sm[0], feature)`;
 * video_model.py:137-151,153-167).  A PROGRAM BLOB -- written once by deepsvc_amd/export.py::export_pframe_program from a model whose
 * weights were loaded and whose tables were built (update()) -- carries the recorded segments of both frame variants (first P-frame of a
 * GOP: feature == NULL, modules.py:425-428; steady state) and of the decoder-scan levels with relocation lists, the packed weights, the
 * scan-workspace and arena sizes and the integer CDF tables.  dsvc_pframe_program_load allocates everything on `device` (hipMalloc) and
 * resolves the relocations; the handle is then good for any number of frames of that size and precision.  Not re-entrant: one handle per
 * stream.  All frame tensors are DEVICE pointers in the reference's layouts:
 *   ref, cur     float32 [1,3,H,W]   (NCHW, H x W = the padded frame size the blob was exported for: dsvc_pframe_program_info)
 *   sm_fea       float32 [1,256,H/4,W/4]   (NCHW; the semantic layer's feature, test_video.py:336-338)
 *   feature      the carried state of the previous frame exactly as dsvc_pframe_decompress wrote it (NHWC, [1,H,W,64] in the blob's storage
 *                type: info.feature_bytes), or NULL for the first P-frame after a key frame (test_video.py:296-297,332)
 *   recon, warped, predict   float32 [1,3,H,W], clamped to [0,1] (video_model.py:167); warped / predict may be NULL
 * Byte strings (host memory): index 0 = motion y, 1 = motion z, 2 = residual y, 3 = residual z -- the reference's
 * mv_out_enc["strings"][0][0], [1][0], res_out_enc["strings"][0][0], [1][0] (compressai 1.2.1 rANS format); capacities from
 * info.string_cap (worst case).  The calls return when the strings are complete (compress) / when every launch has been queued on
 * `stream` (decompress: the decoded tensors are stream-ordered). */
typedef struct dsvc_pframe_program dsvc_pframe_program;
typedef struct dsvc_pframe_info {
    int32_t H, W;               /* padded frame size */
    int32_t precision;          /* DSVC_F32 / DSVC_F16: the arithmetic the blob was recorded in */
    int32_t reserved;
    int64_t feature_bytes;      /* size of the carried feature state */
    int64_t string_cap[4];      /* worst-case size of each byte string */
} dsvc_pframe_info;
int dsvc_pframe_program_load(const void* blob, int64_t nbytes, int32_t device, dsvc_pframe_program** out);
void dsvc_pframe_program_destroy(dsvc_pframe_program* program);
int dsvc_pframe_program_info(const dsvc_pframe_program* program, dsvc_pframe_info* info);
int dsvc_pframe_compress(dsvc_pframe_program* program, void* stream, const float* ref, const float* cur, const float* sm_fea, const void* feature,
                         uint8_t* const out[4], const int64_t cap[4], int64_t nbytes[4]);
int dsvc_pframe_decompress(dsvc_pframe_program* program, void* stream, const float* ref, const uint8_t* const strings[4], const int64_t nbytes[4],
                           const float* sm_fea, const void* feature, void* feature_out, float* recon, float* warped, float* predict);

/* ------------------------------------------------------------------------------------------
 * HOST: integer CDF construction and the range coder (compressai 1.2.1 _CXX / ans; Appendix A.4/A.5)
 * ------------------------------------------------------------------------------------------ */
/* _CXX.pmf_to_quantized_cdf: pmf float[n] -> cdf int32[n+1] */
int dsvc_pmf_to_quantized_cdf(const float* pmf, int32_t n, int32_t precision, int32_t* cdf);

typedef struct dsvc_cdf_table {
    const int32_t* cdf;     /* [rows][stride] */
    int32_t rows, stride;
    const int32_t* cdf_length; /* [rows] */
    const int32_t* offset;     /* [rows] */
    const uint16_t* lut;       /* optional [rows][256] decode accelerator: MUST come from dsvc_cdf_build_lut (which validates
                                  every row: starts at 0, strictly increasing, ends at 65536), or NULL -- then the coder
                                  validates the rows itself on every call */
} dsvc_cdf_table;
/* lut_out[row*256 + b] = largest symbol s with cdf[row][s] <= (b << 8): the decoder starts its search there.
 * Validates every row and registers (lut_out -> fingerprint of the table) so that coder calls carrying this LUT skip the per-call
 * validation.  Contract: the table is not modified while the LUT is in use (rebuild the LUT after any change; a fingerprint of the
 * geometry and of nine entries per row catches recycled buffers, not adversarial in-place edits).  A LUT the library did not build
 * for the table it arrives with is ignored (binary search + per-call validation).  The registry holds 256 LUTs; beyond that
 * dsvc_get_counter("lut_registry_overflow") counts tables that fell back to per-call validation. */
int dsvc_cdf_build_lut(const dsvc_cdf_table* t, uint16_t* lut_out);
/* Drop a LUT from the registry before its memory is freed (returns 1 if it was registered, 0 otherwise). */
int dsvc_cdf_forget_lut(const uint16_t* lut);

/* BufferedRansEncoder (image_model.py:221,253-254) */
int dsvc_rans_encoder_create(void** enc);
void dsvc_rans_encoder_destroy(void* enc);
int dsvc_rans_encode_with_indexes(void* enc, const int32_t* symbols, const int32_t* indexes, int64_t n,
                                  const dsvc_cdf_table* t);
/* flush(): returns the byte count; if out==NULL only queries; encoder is reset after a successful copy */
int64_t dsvc_rans_flush(void* enc, uint8_t* out, int64_t cap);
/* RansDecoder (image_model.py:273-274,288); the stream bytes are copied */
int dsvc_rans_decoder_create(void** dec, const uint8_t* data, int64_t nbytes);
void dsvc_rans_decoder_destroy(void* dec);
int dsvc_rans_decode_stream(void* dec, const int32_t* indexes, int64_t n, const dsvc_cdf_table* t, int32_t* out);

/* ------------------------------------------------------------------------------------------
 * DEVICE: multi-stream range coder -- a labelled, NON-reference-compatible coding of the same symbols with the same tables (SURVEY.md
 * section 8f-2; csrc/rans_ms.hip).  The reference's format (one rANS state per string, image_model.py:253-254,273-274,288) forces a
 * host round trip and a synchronisation at every dependency level of the decoder scan; here a level's symbols are dealt round-robin to
 * n_streams independent 32-bit rANS streams (16-bit words), one GPU thread each, so a level decodes in ONE stream-ordered launch.
 * `dev_table`: a dsvc_cdf_table whose four arrays (cdf, cdf_length, offset, lut -- the lut as dsvc_cdf_build_lut makes it) are DEVICE
 * copies of a validated host table.  Plane order / index meaning as dsvc_rans_encode_with_indexes.
 *   encode: stream t writes its words downwards from the end of slots[t * slot_words ..], reports its final state and word count;
 *           slot_words >= dsvc_rans_ms_slot_words(ceil(n / n_streams)).  The caller concatenates the streams' words (host or device).
 *   decode: states / word_offsets[n_streams + 1] (in 16-bit words from `payload`) per stream -> symbols_out[n].
 * deepsvc_amd/entropy_models.py defines the string layout built around these (header, per-level states / lengths, payload). */
int64_t dsvc_rans_ms_slot_words(int32_t symbols_per_stream);
int dsvc_rans_ms_encode(void* stream, const int32_t* symbols, const int32_t* indexes, int64_t n, int32_t n_streams, const dsvc_cdf_table* dev_table,
                        uint16_t* slots, int32_t slot_words, uint32_t* states, uint16_t* lens);
int dsvc_rans_ms_decode(void* stream, const int32_t* indexes, int64_t n, int32_t n_streams, const dsvc_cdf_table* dev_table,
                        const uint32_t* states, const uint32_t* word_offsets, const uint16_t* payload, int64_t payload_words, int32_t* symbols_out);

#ifdef __cplusplus
}
#endif
#endif /* DSVC_H */

/*
 * ssmi.h - C ABI of libssmi.so: hand-written HIP (gfx950 / CDNA4) kernels for the training hot path of
 * jnissin/semantic-segmentation (ENet-style encoder-decoders, custom pooling layers, class-weighted
 * pixelwise cross-entropy, confusion-matrix metrics, Adam/SGD).
 *
 * The reference has no native/FFI boundary: its operator API is Python/Keras and the arithmetic lives in
 * TensorFlow kernels.  Each entry point below therefore cites the reference call site (path relative to the
 * reference checkout) whose TensorFlow/Keras op it replaces.
 *
 * Conventions
 *   - every function returns an ssmi_status (0 = OK, negative = error); ssmi_last_error() gives the text of
 *     the calling thread's last failure.
 *   - all tensor arguments are DEVICE pointers owned by the caller, dense NHWC unless a leading-dimension
 *     argument says otherwise, float32 unless the name says otherwise.
 *   - every call is asynchronous and ordered on the hipStream_t passed as `stream` (void*, may be NULL for
 *     the default stream).  No entry point allocates or frees device memory; scratch space is passed in as
 *     `ws` and sized by the matching *_workspace_bytes() query.
 *   - the library holds no global mutable state except the per-thread error string.
 */
#ifndef SSMI_H_
#define SSMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t ssmi_status;
enum {
  SSMI_OK = 0,
  SSMI_E_INVALID_ARG = -1,
  SSMI_E_SHAPE = -2,
  SSMI_E_UNSUPPORTED = -3,
  SSMI_E_HIP = -4,
  SSMI_E_WORKSPACE = -5
};

/* Element type of ACTIVATION tensors (layer inputs / outputs and their gradients).  Parameters, BatchNormalization
 * statistics, optimizer state, partial sums and all arithmetic are float32 whatever the activation type; pooling
 * indices are int32 and confusion matrices int64.  SSMI_BF16 halves the bytes every HBM-bound kernel of the step moves
 * (north-star configuration "fp32 and bf16 training"; the reference itself is float32 only, src/settings.py).
 * Entry points whose descriptor carries a `dtype` field take `void*` activation pointers of that type; the others have
 * a twin with the suffix _bf16 whose activation pointers are `uint16_t*` (bfloat16 bit patterns). */
enum { SSMI_F32 = 0, SSMI_BF16 = 1 };

const char* ssmi_last_error(void);
/* Name of the kernel variant the calling thread's last convolution entry point dispatched, as it appears in a
 * rocprofv3 kernel trace (e.g. "conv_gemm_lds<true, 4, 1, true>"); profiling aid, no reference counterpart. */
const char* ssmi_last_kernel(void);
/* Per-launch timing of the convolution kernels (forward and data gradient), for roofline reports; no reference
 * counterpart.  Between ssmi_profile_begin() and ssmi_profile_end() every such launch is bracketed by HIP events on
 * its own stream (do not use under graph capture); _end synchronises on them and returns the number of records;
 * _get returns a record: kernel variant name, duration in ms, algorithmic bytes and flops of that launch. */
ssmi_status ssmi_profile_begin(void);
ssmi_status ssmi_profile_end(int32_t* count);
ssmi_status ssmi_profile_get(int32_t index, char* name, int32_t name_len, float* ms, double* bytes, double* flops);
/* What an event bracket adds to any launch: median time between two events recorded around an EMPTY kernel on the
 * profiled stream, measured by the last ssmi_profile_end().  _get returns raw durations; subtract this to compare with a
 * kernel trace (rocprofv3 reports the kernel's own begin/end timestamps). */
float ssmi_profile_bracket_overhead_ms(void);
int32_t ssmi_version(void);
/* A one-thread kernel that stores the device's constant-rate wall clock (ticks; ssmi_device_timestamp_khz() ticks per
 * millisecond, 0 if unknown) into *out when the stream reaches it.  Unlike an event it can be captured into a HIP graph
 * and read after every replay: the only way to time points INSIDE a captured step (where the main chain ends, where a
 * segment starts).  Profiling aid, no reference counterpart. */
ssmi_status ssmi_device_timestamp(int64_t* out, void* stream);
int32_t ssmi_device_timestamp_khz(void);

/* ------------------------------------------------------------------------------------------------------
 * Convolutions.  Kernel layout HWIO = [kh][kw][cin][cout] (Keras Conv2D).  Cross-correlation.
 * The descriptor always describes the FORWARD convolution x[n,h,w,cin] -> y[n,ho,wo,cout]:
 *   y[n,oy,ox,co] = bias[co] + sum_{ky,kx,ci} x[n, oy*stride + ky*dil - pad_t, ox*stride + kx*dil - pad_l, ci] * w[ky,kx,ci,co]
 * with out-of-range x read as zero.  pad_t/pad_l are the TF "SAME" pad_before values (0 for VALID);
 * ssmi_same_pad() computes them.  x_ld / y_ld are the element strides between consecutive pixels of x / y
 * (>= cin / cout) so that a convolution can read or write a channel slice of a wider tensor (the concat of
 * the initial block, src/models.py:1063).  Activation tensors (x, y, their gradients, add, skip, bn_x) are `void*`: their
 * element type is d->dtype; kernels, biases and kernel gradients are always float32.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct ssmi_conv_desc {
  int32_t n, h, w, cin;
  int32_t ho, wo, cout;
  int32_t kh, kw;
  int32_t stride;
  int32_t dil;
  int32_t pad_t, pad_l;
  int32_t x_ld, y_ld;
  int32_t dtype;   /* SSMI_F32 (0) or SSMI_BF16: element type of x, y, their gradients, `add`, `skip` and `bn_x` */
  /* Optional (NULL = none), SSMI_BF16 only: bfloat16 copies of this layer's kernel written by ssmi_conv_pack_weights_bf16,
   *   w_fwd_pack[tap][co][ci] and w_dgrad_pack[kh*kw-1-tap][ci][co] = bf16(w[tap][ci][co]).
   * With them the forward convolution / the stride-1 data gradient of layers with cin (resp. cout) a multiple of 16 and more
   * than 16 output (resp. input) channels run on the bfloat16 matrix instructions (v_mfma_f32_32x32x16_bf16): products of
   * bfloat16 activations and bfloat16-ROUNDED weights, float32 accumulation.  So do the stride-1 layers with at most 16 output
   * (resp. input) channels whose reduction is 1x1 over 16 / 64 / 128 or 3x3 over 16 channels (v_mfma_f32_16x16x16_bf16).
   * ssmi_conv2d_pack_usage() tells which packs a layer would use.  The packs must be rewritten whenever w changes. */
  const void* w_fwd_pack;
  const void* w_dgrad_pack;
} ssmi_conv_desc;

/* Per-channel statistics and parameters of a BatchNormalization as the kernels consume them (float32 device vectors). */
typedef struct ssmi_bn_params {
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
} ssmi_bn_params;

/* TF SAME padding rule: out = ceil(in/stride); total = max((out-1)*stride + (k-1)*dil + 1 - in, 0);
 * before = total/2.  Host-only helper. */
ssmi_status ssmi_same_pad(int32_t in, int32_t k, int32_t stride, int32_t dil, int32_t* out, int32_t* pad_before);

/* Keras Conv2D forward (src/models.py:1056-1060, 1090-1096, 1105-1133, 1142-1147).  bias may be NULL. */
ssmi_status ssmi_conv2d_fwd(const ssmi_conv_desc* d, const void* x, const float* w, const float* bias, void* y,
                            void* stream);
/* Inference-phase fusion: convolution + BatchNormalization with fixed statistics + activation in one launch
 * (src/models.py: every Conv2D -> BatchNormalization -> PReLU / LeakyReLU chain when the learning phase is 0):
 *   y = act(gamma * ((conv(x) + bias - mean) * invstd) + beta),  act in {SSMI_ACT_NONE, SSMI_ACT_PRELU (alpha per channel),
 *   SSMI_ACT_LEAKY (slope)}; mean / invstd come from ssmi_bn_infer_stats(_arena).  Same arithmetic as ssmi_bn_act_fwd. */
ssmi_status ssmi_conv2d_fwd_bn_act_infer(const ssmi_conv_desc* d, const void* x, const float* w, const float* bias,
                                         const float* mean, const float* invstd, const float* gamma, const float* beta,
                                         int32_t act, const float* alpha, float slope, void* y, void* stream);
/* The same with a skip tensor added between the normalisation and the activation:
 *   y = act(gamma * ((conv(x) + bias - mean) * invstd) + beta + skip[pixel * skip_ld + channel])
 * i.e. the last projection of a regular bottleneck, its BatchNormalization, the Add with the block input and the final
 * PReLU (src/models.py:1142-1174) in one launch when the learning phase is 0 (SpatialDropout2D is the identity then).
 * skip has the output's pixel grid, skip_ld >= cout floats per pixel; skip == NULL gives ssmi_conv2d_fwd_bn_act_infer.
 * Same arithmetic, in the same order, as ssmi_conv2d_fwd -> ssmi_merge_fwd(other_mode 0, drop NULL). */
ssmi_status ssmi_conv2d_fwd_bn_add_act_infer(const ssmi_conv_desc* d, const void* x, const float* w, const float* bias,
                                             const float* mean, const float* invstd, const float* gamma, const float* beta,
                                             const void* skip, int32_t skip_ld, int32_t act, const float* alpha,
                                             float slope, void* y, void* stream);

/* Convolution whose output feeds a training-phase BatchNormalization (every conv -> BN pair of src/models.py, e.g.
 * :1096-1100): same y as ssmi_conv2d_fwd plus the statistics ssmi_bn_train_stats would compute from y (save_mean,
 * save_invstd, moving-average update), with the per-channel sums taken in the convolution kernel's epilogue instead of
 * a second pass over y.  ws: ssmi_conv2d_fwd_bn_stats_workspace_bytes(d). */
size_t ssmi_conv2d_fwd_bn_stats_workspace_bytes(const ssmi_conv_desc* d);
ssmi_status ssmi_conv2d_fwd_bn_stats(const ssmi_conv_desc* d, const void* x, const float* w, const float* bias, void* y,
                                     float eps, float momentum, int32_t unbiased_moving, float* moving_mean,
                                     float* moving_var, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes,
                                     void* stream);

/* Normalise on load: the 1x1 expansion of a bottleneck (src/models.py:1138-1147: BatchNormalization -> PReLU -> Conv2D 1x1)
 * with the BatchNormalization + PReLU applied to the operand inside the convolution kernels, so that the normalised tensor is
 * never stored: x_raw is the RAW input of that BatchNormalization (the previous convolution's output), `pre` its final
 * statistics and parameters (training phase: save_mean / save_invstd), pre_alpha the PReLU slopes (NULL: no activation).
 *   ssmi_conv2d_fwd_bn_stats_prebn(d, x_raw, ...)  == ssmi_conv2d_fwd_bn_stats(d, ssmi_bn_act_fwd(x_raw), ...)
 *   ssmi_conv2d_bwd_weight_prebn(d, x_raw, ...)    == ssmi_conv2d_bwd_weight(d, ssmi_bn_act_fwd(x_raw), ...)
 * bit for bit (the same arithmetic per element; bfloat16 storage rounds the operand as the stored tensor would have been).
 * 1x1 stride-1 layers over 8..64 input channels (whole blocks of 8; 16 for bfloat16) with more than 32 output channels whose
 * weight gradient takes one tile per wave: ssmi_conv2d_prebn_supported says whether a layer qualifies; others return
 * SSMI_E_UNSUPPORTED.  The data gradient of the layer is unchanged (it produces
 * the gradient of the normalised tensor, which ssmi_conv2d_bwd_data_bn_bwd pushes through the BatchNormalization). */
int32_t ssmi_conv2d_prebn_supported(const ssmi_conv_desc* d, int32_t with_bias);
ssmi_status ssmi_conv2d_fwd_bn_stats_prebn(const ssmi_conv_desc* d, const void* x_raw, const ssmi_bn_params* pre,
                                           const float* pre_alpha, const float* w, const float* bias, void* y, float eps,
                                           float momentum, int32_t unbiased_moving, float* moving_mean, float* moving_var,
                                           float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, void* stream);
ssmi_status ssmi_conv2d_bwd_weight_prebn(const ssmi_conv_desc* d, const void* x_raw, const ssmi_bn_params* pre,
                                         const float* pre_alpha, const void* dy, float* dw, float* dbias, void* ws,
                                         size_t ws_bytes, void* stream);

/* The gradient a bottleneck's merge backward hands to its 1x1 expansion, left unformed: ssmi_merge_bwd_deferred stops
 * after the reduction (dsum + the three coefficient vectors k1, m1, m2 in `coef`, 3 * c floats), and a consumer forms
 *   dmain = k1 * (dsum * drop - m1 - ((main - mean) * invstd) * m2)          (the arithmetic of ssmi_merge_bwd)
 * from these operands itself.  drop: [n, c] table or NULL; hw: pixels per image. */
typedef struct ssmi_premerge {
  const void* dsum;
  const void* main;
  const float* mean;
  const float* invstd;
  const float* coef;
  const float* drop;
  int64_t hw;
} ssmi_premerge;
/* ssmi_conv2d_bwd_data_bn_bwd_rot for the expansion convolution of a bottleneck (src/models.py:1142-1147) whose output
 * gradient is given in that unformed state: the data-gradient kernel applies the merge backward to its operand, so the
 * main chain of the backward pass does not wait for (or read back) the stored tensor.  Results are bit-identical to
 * ssmi_merge_bwd followed by ssmi_conv2d_bwd_data_bn_bwd_rot.  float32; ssmi_conv2d_bwd_data_premerge_supported(d) says
 * whether the layer qualifies (1x1, stride 1, forward-form data gradient, at most 32 input and 128 output channels). */
/* Weight gradients finish with a reduction over per-workgroup partial sums, a 5 us launch per layer.  Between
 * ssmi_wg_reduce_defer_begin() and ssmi_wg_reduce_defer_flush() the weight-gradient entry points called BY THIS HOST THREAD
 * (ssmi_conv2d_bwd_weight*, ssmi_conv2d_transpose_bwd_weight, ssmi_expand_weight_reduce) leave that reduction out and remember
 * it; flush launches all remembered ones as one kernel per 48 layers on `stream` (bit-identical results) and, with end != 0,
 * stops the deferral.  Until the flush the gradients are not written and every deferred layer's workspace must stay untouched:
 * give each layer its own.  Layers with 16384 or more gradient elements are reduced at once as before. */
ssmi_status ssmi_wg_reduce_defer_begin(void);
ssmi_status ssmi_wg_reduce_defer_flush(int32_t end, void* stream);

/* ssmi_conv2d_bwd_data_bn_bwd_rot stopped after the reduction: da = the plain data gradient (behind the BatchNormalization +
 * activation), dgamma / dbeta / dalpha written, coef = k1 | m1 | m2 (3 * cin floats) for a consumer that runs the apply pass
 *   dx_bn = k1 * (da * act'(z) - m1 - xhat * m2)
 * itself (ssmi_expand_next_apply).  _supported: the layer takes the fused reduction (stride 1, cin % 4 == 0, ...). */
int32_t ssmi_conv2d_bwd_data_bn_bwd_deferred_supported(const ssmi_conv_desc* d, int32_t ld_x);
ssmi_status ssmi_conv2d_bwd_data_bn_bwd_rot_deferred(const ssmi_conv_desc* d, const void* dy, const float* w, const float* w_rot,
                                                     void* da, const void* bn_x, int32_t ld_x, const float* mean, const float* invstd,
                                                     const float* gamma, const float* beta, int32_t act, const float* alpha, float slope,
                                                     float* dgamma, float* dbeta, float* dalpha, float* coef, void* ws, size_t ws_bytes,
                                                     void* stream);
int32_t ssmi_conv2d_bwd_data_premerge_supported(const ssmi_conv_desc* d);
ssmi_status ssmi_conv2d_bwd_data_bn_bwd_premerge(const ssmi_conv_desc* d, const ssmi_premerge* pm, const float* w,
                                                 const float* w_rot, void* da, const void* bn_x, int32_t ld_x,
                                                 const float* mean, const float* invstd, const float* gamma,
                                                 const float* beta, int32_t act, const float* alpha, float slope,
                                                 void* dx_bn, float* dgamma, float* dbeta, float* dalpha, void* ws,
                                                 size_t ws_bytes, void* stream);

/* Gradient w.r.t. x (tf.gradients through Conv2D).  dx[n,h,w,cin] (pixel stride x_ld) is fully overwritten:
 * dx = conv_bwd_data(dy, w) + (add ? add : 0), where `add` (nullable) has the layout of dx.  The `add` operand
 * carries the gradient of the block's skip branch (src/models.py:1171). */
ssmi_status ssmi_conv2d_bwd_data(const ssmi_conv_desc* d, const void* dy, const float* w, const void* add, void* dx,
                                 void* stream);

/* Data gradient with a workspace: wide stride-1 layers (both channel counts >= 64) run as a forward convolution over dy
 * with the rotated, transposed kernel matrix, which is built in `ws` by a small launch in front (src/models.py:828-879, the
 * 3x3 Conv2D stacks of U-Net / SegNet, whose gradients Keras derives).  ssmi_conv2d_bwd_data_workspace_bytes() may be 0:
 * then ws is not touched and the call equals ssmi_conv2d_bwd_data. */
/* Pre-rotated kernel matrices: ssmi_conv_rotate_weights writes, for every table entry, the data-gradient form
 *   w_rot[kh-1-ky][kw-1-kx][co][ci] = w[ky][kx][ci][co]
 * of the convolution kernel at theta + offset to theta_rot + offset (one launch for a whole parameter arena; `table` is
 * DEVICE memory).  The _rot entry points take that matrix beside the original and run the data gradient of stride-1 layers
 * as a forward convolution over dy wherever ssmi_conv2d_bwd_data_uses_rotated() says so (w_rot may be NULL: then they equal
 * the plain entry points).  No counterpart in the reference: Keras derives these gradients (src/models.py:1069-1319). */
typedef struct ssmi_rot_entry {
  int64_t offset;            /* in floats, from theta / theta_rot */
  int32_t kh, kw, cin, cout;
} ssmi_rot_entry;
/* bfloat16 kernel packs for a whole parameter arena (one launch): for every table entry (device memory) writes
 * fwd_pack / dgrad_pack (bfloat16 arrays of 2 * n elements for a theta of n floats, either may be NULL; the pack of the entry
 * at theta + offset starts at element 2 * offset, i.e. at the same BYTE offset as in theta) as described at ssmi_conv_desc.
 * ssmi_conv2d_pack_usage(desc): bit 0 = the forward pass would use w_fwd_pack, bit 1 = the data gradient w_dgrad_pack. */
ssmi_status ssmi_conv_pack_weights_bf16(const ssmi_rot_entry* table, int32_t count, const float* theta, uint16_t* fwd_pack,
                                        uint16_t* dgrad_pack, void* stream);
int32_t ssmi_conv2d_pack_usage(const ssmi_conv_desc* d);
int32_t ssmi_conv2d_bwd_data_uses_rotated(const ssmi_conv_desc* d);
ssmi_status ssmi_conv_rotate_weights(const ssmi_rot_entry* table, int32_t count, const float* theta, float* theta_rot,
                                     void* stream);
ssmi_status ssmi_conv2d_bwd_data_rot(const ssmi_conv_desc* d, const void* dy, const float* w, const float* w_rot, const void* add,
                                     void* dx, void* stream);
size_t ssmi_conv2d_bwd_data_workspace_bytes(const ssmi_conv_desc* d);
ssmi_status ssmi_conv2d_bwd_data_ws(const ssmi_conv_desc* d, const void* dy, const float* w, const void* add, void* dx,
                                    void* ws, size_t ws_bytes, void* stream);
/* Data gradient whose result is the gradient of a BatchNormalization(+activation, +SpatialDropout2D) OUTPUT, i.e. every
 * "BN -> act -> conv" chain walked backwards (src/models.py:1096-1118 and the convolution blocks :846-876).  Same `da`
 * as ssmi_conv2d_bwd_data (desc = the convolution; da has the x-like shape [n,h,w,cin], pixel stride x_ld) followed by
 * ssmi_bn_act_drop_bwd on it (bn_x = the raw input of that BN, pixel stride ld_x): dx_bn, dgamma, dbeta, dalpha.  For
 * stride-1 convolutions the per-channel sums are taken in the data-gradient kernel's epilogue, which saves one pass over
 * da and bn_x; other shapes run the two plain passes.  ws: ssmi_conv2d_bwd_data_bn_bwd_workspace_bytes(d). */
size_t ssmi_conv2d_bwd_data_bn_bwd_workspace_bytes(const ssmi_conv_desc* d);
ssmi_status ssmi_conv2d_bwd_data_bn_bwd(const ssmi_conv_desc* d, const void* dy, const float* w, const void* add,
                                        void* da, const void* bn_x, int32_t ld_x, const float* mean,
                                        const float* invstd, const float* gamma, const float* beta, int32_t act,
                                        const float* alpha, float slope, const float* drop, int64_t hw, void* dx_bn,
                                        float* dgamma, float* dbeta, float* dalpha, void* ws, size_t ws_bytes,
                                        void* stream);
/* the same with a pre-rotated kernel matrix (see ssmi_conv_rotate_weights) */
ssmi_status ssmi_conv2d_bwd_data_bn_bwd_rot(const ssmi_conv_desc* d, const void* dy, const float* w, const float* w_rot, const void* add,
                                        void* da, const void* bn_x, int32_t ld_x, const float* mean,
                                        const float* invstd, const float* gamma, const float* beta, int32_t act,
                                        const float* alpha, float slope, const float* drop, int64_t hw, void* dx_bn,
                                        float* dgamma, float* dbeta, float* dalpha, void* ws, size_t ws_bytes,
                                        void* stream);

/* Gradient w.r.t. the kernel (and bias when dbias != NULL).  dw/dbias are overwritten.  Deterministic
 * two-stage reduction through `ws`. */
size_t ssmi_conv2d_bwd_weight_workspace_bytes(const ssmi_conv_desc* d);
ssmi_status ssmi_conv2d_bwd_weight(const ssmi_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                                   void* ws, size_t ws_bytes, void* stream);

/* Keras Conv2DTranspose(padding='same', strides=s) (src/models.py:1244-1250, 1311-1317, 1556-1562, 1620-1626) =
 * gradient of the SAME forward convolution described by `d`:
 *   tconv input  xt[n, ho, wo, cout],  tconv output yt[n, h, w, cin] (h = ho*stride, w = wo*stride),
 *   kernel (kh, kw, tconv_out = cin, tconv_in = cout)  -- the Keras Conv2DTranspose kernel layout, which is the
 *   HWIO layout of the forward convolution `d`.  bias has cin entries (nullable). */
ssmi_status ssmi_conv2d_transpose_fwd(const ssmi_conv_desc* d, const void* xt, const float* w, const float* bias,
                                      void* yt, void* stream);
ssmi_status ssmi_conv2d_transpose_bwd_data(const ssmi_conv_desc* d, const void* dyt, const float* w, void* dxt,
                                           void* stream);
size_t ssmi_conv2d_transpose_bwd_weight_workspace_bytes(const ssmi_conv_desc* d);
ssmi_status ssmi_conv2d_transpose_bwd_weight(const ssmi_conv_desc* d, const void* xt, const void* dyt, float* dw,
                                             float* dbias, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * BatchNormalization(axis=-1, momentum, epsilon) + activation (src/models.py:1064-1065, 1100-1101, ...).
 * act: 0 = none, 1 = PReLU(shared_axes=[1,2]) with per-channel alpha, 2 = LeakyReLU(slope).
 * ---------------------------------------------------------------------------------------------------- */
enum { SSMI_ACT_NONE = 0, SSMI_ACT_PRELU = 1, SSMI_ACT_LEAKY = 2 };

/* Training-phase statistics of x[npix, c] (pixel stride ld): save_mean, save_invstd = 1/sqrt(biased_var + eps),
 * and the Keras moving-average update  moving <- momentum*moving + (1-momentum)*batch  (momentum is the weight of
 * the OLD value; the reference uses 0.1).  unbiased_moving != 0 feeds the Bessel-corrected variance into the moving
 * update (tf.nn.fused_batch_norm behaviour).  moving_* may be NULL (no update). */
size_t ssmi_bn_workspace_bytes(int64_t npix, int32_t c);
ssmi_status ssmi_bn_train_stats(const float* x, int64_t npix, int32_t c, int32_t ld, float eps, float momentum,
                                int32_t unbiased_moving, float* moving_mean, float* moving_var, float* save_mean,
                                float* save_invstd, void* ws, size_t ws_bytes, void* stream);
/* Inference phase: save_mean = moving_mean, save_invstd = 1/sqrt(moving_var + eps). */
ssmi_status ssmi_bn_infer_stats(const float* moving_mean, const float* moving_var, int32_t c, float eps,
                                float* save_mean, float* save_invstd, void* stream);
/* The same for every BatchNormalization layer of a model in one launch: `state` is an arena holding the layers'
 * [moving_mean | moving_variance] vectors, is_var[i] != 0 marks the variance elements, out[i] = state[i] for means and
 * 1 / sqrt(state[i] + eps) for variances (the inference-phase forward of a whole network needs one such launch). */
ssmi_status ssmi_bn_infer_stats_arena(const float* state, const uint8_t* is_var, int64_t count, float eps, float* out,
                                      void* stream);
/* y = act(gamma * (x - mean) * invstd + beta).  x and y dense [npix, c] (x pixel stride ld_x). */
ssmi_status ssmi_bn_act_fwd(const float* x, int32_t ld_x, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, int32_t act, const float* alpha, float slope, float* y, int64_t npix,
                            int32_t c, void* stream);
/* Backward of the above in the training phase (statistics are functions of x):
 *   dx (pixel stride ld_x), dgamma, dbeta, dalpha (PReLU only) are overwritten. */
ssmi_status ssmi_bn_act_bwd(const float* dy, const float* x, int32_t ld_x, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, int32_t act, const float* alpha, float slope,
                            float* dx, float* dgamma, float* dbeta, float* dalpha, int64_t npix, int32_t c, void* ws,
                            size_t ws_bytes, void* stream);

/* The convolution block of the U-Net / SegNet builders (src/models.py:828-878): BatchNormalization ->
 * [LeakyReLU] -> SpatialDropout2D as ONE pass.  Same as ssmi_bn_act_fwd / _bwd with two additions:
 *   drop  [n, c] table of 0 or 1/(1-rate) per (image, channel), nullable; hw = pixels per image (npix % hw == 0);
 *   ld_y / ld_dy  pixel stride of the output / incoming gradient, so the result can land in a channel slice of a
 *   concatenation buffer (src/models.py:927).
 *   y = drop * act(bn(x));  backward: dz = dy * drop * act'(.), then the BN backward of ssmi_bn_act_bwd. */
ssmi_status ssmi_bn_act_drop_fwd(const float* x, int32_t ld_x, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int32_t act, const float* alpha, float slope,
                                 const float* drop, int64_t hw, float* y, int32_t ld_y, int64_t npix, int32_t c,
                                 void* stream);
ssmi_status ssmi_bn_act_drop_bwd(const float* dy, int32_t ld_dy, const float* x, int32_t ld_x, const float* mean,
                                 const float* invstd, const float* gamma, const float* beta, int32_t act,
                                 const float* alpha, float slope, const float* drop, int64_t hw, float* dx,
                                 float* dgamma, float* dbeta, float* dalpha, int64_t npix, int32_t c, void* ws,
                                 size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Bottleneck merge (src/models.py:1150-1172, 1265-1296, 1594-1607): one fused pass for
 *   out = PReLU_alpha( dropscale[n,c] * BN3(main) + other' )
 * other_mode selects other':
 *   0  identity           other[n,h,w,c]
 *   1  channel zero-pad   other[n,h,w,c_other], zeros for channels >= c_other      (MaxPooling2D + ZeroPadding2D branch)
 *   2  nearest 2x upsample of BN_o(other[n,h/2,w/2,c])                             (UpSampling2D branch)
 *   3  MaxUnpooling2D of BN_o(other[n,h/2,w/2,c]) with idx[n,h/2,w/2,c]            (window-local int32 indices)
 * drop (nullable) is the SpatialDropout2D keep/scale table [n,c] (0 or 1/(1-rate)).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct ssmi_merge_desc {
  int32_t n, h, w, c;
  int32_t other_mode;
  int32_t c_other;
  int32_t dtype;   /* SSMI_F32 (0) or SSMI_BF16: element type of main, other, out and their gradients */
} ssmi_merge_desc;

ssmi_status ssmi_merge_fwd(const ssmi_merge_desc* d, const void* main, const ssmi_bn_params* bn3, const float* drop,
                           const void* other, const ssmi_bn_params* bn_other, const int32_t* idx, const float* alpha,
                           void* out, void* stream);
/* Backward.  Outputs (all overwritten):
 *   dmain   [n,h,w,c]   gradient w.r.t. the raw `main` tensor (through dropout and BN3, training phase)
 *   dsum    [n,h,w,c]   gradient w.r.t. the pre-activation sum (= gradient of the identity / padded skip branch)
 *   dother_bn [n,h/2,w/2,c] (modes 2,3 only) gradient w.r.t. BN_o's OUTPUT (2x2 sum / gather of dsum); the caller
 *                       pushes it through ssmi_bn_act_bwd(act = none) to reach the other-branch convolution
 *   dgamma3, dbeta3, dalpha  parameter gradients */
size_t ssmi_merge_workspace_bytes(const ssmi_merge_desc* d);
ssmi_status ssmi_merge_bwd(const ssmi_merge_desc* d, const void* dout, const void* main, const ssmi_bn_params* bn3,
                           const float* drop, const void* other, const ssmi_bn_params* bn_other, const int32_t* idx,
                           const float* alpha, void* dmain, void* dsum, void* dother_bn, float* dgamma3,
                           float* dbeta3, float* dalpha, void* ws, size_t ws_bytes, void* stream);

/* ssmi_merge_bwd in two halves: _deferred runs the reduction (dsum, dother_bn, parameter gradients) and leaves the
 * coefficients of the last pass in `coef` (3 * c floats, caller-owned, must live until the consumers have run);
 * _apply is that last pass (dmain), for consumers that want the tensor stored after all.  Together they are ssmi_merge_bwd. */
ssmi_status ssmi_merge_bwd_deferred(const ssmi_merge_desc* d, const void* dout, const void* main, const ssmi_bn_params* bn3,
                                    const float* drop, const void* other, const ssmi_bn_params* bn_other, const int32_t* idx,
                                    const float* alpha, void* dsum, void* dother_bn, float* dgamma3, float* dbeta3,
                                    float* dalpha, float* coef, void* ws, size_t ws_bytes, void* stream);
ssmi_status ssmi_merge_bwd_apply(const ssmi_merge_desc* d, const void* dsum, const void* main, const ssmi_bn_params* bn3,
                                 const float* drop, const float* coef, void* dmain, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Pooling family.
 * ---------------------------------------------------------------------------------------------------- */
/* tf.nn.max_pool_with_argmax(ksize 2x2, strides 2, SAME) as wrapped by MaxPoolingWithArgmax2D
 * (src/layers/pooling.py:43-49) and, with idx == NULL, Keras MaxPooling2D 2x2 (src/models.py:1062,1159; identical
 * values for even h,w).  Window scanned h then w, strict '>' so the first maximum wins; index =
 * (iy*w + ix)*c + ch inside the image; windows are clipped at the border.
 * y has pixel stride y_ld (>= c).  idx_i32 and idx_f32 (the reference returns float32 indices) are nullable,
 * dense [n,ho,wo,c]. */
ssmi_status ssmi_maxpool_argmax_fwd(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, float* y, int32_t y_ld,
                                    int32_t* idx_i32, float* idx_f32, void* stream);
/* Gradient routing to the argmax position (gather form over the input grid, no atomics):
 *   dx[n,iy,ix,ch] (+)= (idx[n,iy/2,ix/2,ch] == (iy*w+ix)*c+ch) ? dy[n,iy/2,ix/2,ch] : 0
 * dy has pixel stride dy_ld.  accumulate != 0 adds into dx. */
ssmi_status ssmi_maxpool_argmax_bwd(const float* dy, int32_t dy_ld, const int32_t* idx, int32_t n, int32_t h, int32_t w,
                                    int32_t c, float* dx, int32_t accumulate, void* stream);
/* MaxUnpooling2D (src/layers/pooling.py:71-103): y[n, idx] += x, zeros elsewhere; x,idx [n,h,w,c] -> y [n,2h,2w,c].
 * General tf.scatter_nd semantics (any in-range index, duplicates sum) : memset + atomic scatter. */
ssmi_status ssmi_maxunpool_fwd(const float* x, const int32_t* idx, int32_t n, int32_t h, int32_t w, int32_t c, float* y,
                               void* stream);
/* Same result for window-local indices (those produced by ssmi_maxpool_argmax_fwd): pure gather over the output,
 * no memset, no atomics. */
ssmi_status ssmi_maxunpool_local_fwd(const float* x, const int32_t* idx, int32_t n, int32_t h, int32_t w, int32_t c,
                                     float* y, void* stream);
/* Gradient of scatter_nd w.r.t. the values: dx[n,h,w,c] = dy[n, idx[n,h,w,c]]. */
ssmi_status ssmi_maxunpool_bwd(const float* dy, const int32_t* idx, int32_t n, int32_t h, int32_t w, int32_t c,
                               float* dx, void* stream);
/* float32 <-> int32 index views (the reference moves indices as float32, src/layers/pooling.py:48,78). */
ssmi_status ssmi_idx_f32_to_i32(const float* src, int32_t* dst, int64_t count, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Loss + metrics.
 * ---------------------------------------------------------------------------------------------------- */
/* _segmentation_sparse_weighted_pixelwise_crossentropy_loss (src/losses.py:243-265) fused with its gradient, the
 * prediction argmax (src/metrics.py:259) and the streaming confusion matrix (src/metrics.py:89-130, 355-378):
 *   loss   = (1/npix) * sum_p w[p] * (logsumexp(z[p,:]) - z[p, label[p]])
 *   dlogits[p,k] = grad_scale * w[p]/npix * (softmax(z[p,:])[k] - [k == label[p]])        (dlogits nullable)
 *   cm[label[p], argmax_k z[p,k]] += 1 for p < npix_metrics with label[p] not in ignore_mask   (cm int64, nullable, accumulated)
 *   npix_metrics may exceed npix (semi-supervised batches, src/generators.py:1764-1766: the metrics see every row, the
 *   classification cost only the labeled ones): pixels in [npix, npix_metrics) update cm / pred_out only, their dlogits
 *   rows are left untouched, and logits / labels must cover npix_metrics pixels.
 * loss_out is a device float, overwritten.  ignore_mask bit k set <=> class k ignored (num_classes <= 64).
 * Labels outside [0, num_classes) are an error in TensorFlow; here they contribute nothing and set *bad_labels
 * (device int32, nullable) to the count seen.  Deterministic loss reduction through ws. */
size_t ssmi_softmax_xent_workspace_bytes(int64_t npix);
ssmi_status ssmi_softmax_xent_fwd_bwd(const float* logits, const int32_t* labels, const float* weights, int64_t npix,
                                      int32_t num_classes, float grad_scale, float* dlogits, float* loss_out,
                                      int64_t* cm, int64_t npix_metrics, uint64_t ignore_mask, int32_t* pred_out,
                                      int32_t* bad_labels, void* ws, size_t ws_bytes, void* stream);
/* The network's last layer and the loss in one pass: the `logits` layer of the ENets,
 * Conv2DTranspose(num_classes, (2, 2), strides=(2, 2), padding='same') (src/models.py:1318, 1627), followed by
 * ssmi_softmax_xent_fwd_bwd on its output, without the logits tensor travelling through memory:
 *   logits[n, 2i+ky, 2j+kx, co] = bias[co] + sum_ci x[n, i, j, ci] * kernel[ky, kx, co, ci]      (Keras layout (kh, kw, out, in))
 * x: [n, h, w, cin] of `dtype` with row stride x_ld elements; labels / weights / the optional outputs cover the
 * n * 2h * 2w output pixels; logits_out (nullable) receives the logits the loss was computed from, dlogits (nullable)
 * the gradient for ssmi_conv2d_transpose_bwd_data / _bwd_weight of the same layer; every other argument as in
 * ssmi_softmax_xent_fwd_bwd (workspace size: ssmi_softmax_xent_workspace_bytes).  Built for cin = 16 and 24 classes
 * (ssmi_logits_tconv2x2_xent_supported); other shapes return SSMI_E_UNSUPPORTED and take the two-call route. */
int32_t ssmi_logits_tconv2x2_xent_supported(int32_t cin, int32_t num_classes);
ssmi_status ssmi_logits_tconv2x2_xent_fwd_bwd(int32_t dtype, const void* x, int32_t n, int32_t h, int32_t w, int32_t cin,
                                              int32_t x_ld, const float* kernel, const float* bias, int32_t num_classes,
                                              const int32_t* labels, const float* weights, int64_t npix, float grad_scale,
                                              void* logits_out, void* dlogits, float* loss_out, int64_t* cm,
                                              int64_t npix_metrics, uint64_t ignore_mask, int32_t* pred_out,
                                              int32_t* bad_labels, void* ws, size_t ws_bytes, void* stream);
/* ssmi_softmax_xent_fwd_bwd for callers that cannot allocate or carry a workspace (code under HIP-graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, results and
 * arithmetic (bit-identical) as ssmi_softmax_xent_fwd_bwd, but the per-block partial sums of the loss go through a
 * 16 KiB device array that is part of the library's code object (one copy per GPU, held until the process ends) instead
 * of `ws`.  The call allocates nothing; the first call on a device looks the array's address up once (hipGetSymbolAddress,
 * a runtime call that is not stream work): that first call must happen outside stream capture, or the capture must have
 * been begun with ssmi_graph_begin_capture, which does the lookup for the current device before recording starts; a
 * first call under any other capture returns SSMI_E_UNSUPPORTED and records nothing.  At most
 * ssmi_softmax_xent_static_max_pixels() pixels (max(npix, npix_metrics)); larger inputs are rejected with
 * SSMI_E_UNSUPPORTED.  All calls on one GPU share that GPU's array: calls that may overlap on one GPU (different
 * streams) race on it and must be serialised by the caller.  Nothing in the array has to be zero beforehand. */
int64_t ssmi_softmax_xent_static_max_pixels(void);
ssmi_status ssmi_softmax_xent_fwd_bwd_static(const float* logits, const int32_t* labels, const float* weights, int64_t npix,
                                             int32_t num_classes, float grad_scale, float* dlogits, float* loss_out,
                                             int64_t* cm, int64_t npix_metrics, uint64_t ignore_mask, int32_t* pred_out,
                                             int32_t* bad_labels, void* stream);
/* Host-side derivations from a confusion matrix (rows = labels): tf.metrics.accuracy / mean_per_class_accuracy /
 * mean_iou as used by src/metrics.py:276-352, and the reference's own numpy definitions
 * (src/scripts/calculate_metrics_from_cfms.py:12-53) which drop ignored classes before averaging.
 * out[0]=accuracy, out[1]=tf mean_per_class_accuracy, out[2]=tf mean_iou (classes with non-zero denominator),
 * out[3]=tf mean_iou over all classes, out[4]=script mpca, out[5]=script miou. */
ssmi_status ssmi_metrics_from_cm(const int64_t* cm_host, int32_t num_classes, uint64_t ignore_mask, double* out6);

/* Mean-teacher consistency cost and its gradient (src/losses.py:195-215 called from
 * segmentation_mean_teacher_lambda_loss, src/losses.py:377-436):
 *   cost = coef * mean_pixels( mean_classes( (softmax(teacher_logits) - softmax(logits))^2 ) )
 * over all `npix` pixels (labeled and unlabeled rows).  `coef_dev` is a 1-element DEVICE float so that a captured
 * graph can be replayed while the coefficient follows its schedule.  The teacher logits take no gradient.
 * dlogits (nullable): rows of the first `npix_accumulate` pixels are added to (they already hold the classification
 * gradient written by ssmi_softmax_xent_fwd_bwd over the labeled rows), the remaining rows are overwritten.
 * loss_io (nullable): receives cost, or loss_io + cost when accumulate_loss != 0.  cost_out (nullable) receives cost.
 * Deterministic: per-block partial sums in `ws`, reduced in double by a single block. */
size_t ssmi_mt_consistency_workspace_bytes(int64_t npix);
ssmi_status ssmi_mt_consistency_fwd_bwd(const float* logits, const float* teacher_logits, int64_t npix, int32_t num_classes,
                                        const float* coef_dev, float grad_scale, int64_t npix_accumulate, float* dlogits,
                                        float* loss_io, int32_t accumulate_loss, float* cost_out, void* ws, size_t ws_bytes,
                                        void* stream);

/* Superpixel cost of the unlabeled rows and its gradient (src/losses.py:108-192, called from
 * segmentation_superpixel_lambda_loss :439-492 and segmentation_mean_teacher_superpixel_lambda_loss :495-572):
 *   S = softmax(logits);  Gx, Gy = depthwise 3x3 cross-correlation of every class map with the Sobel mask
 *   [[1,2,1],[0,0,0],[-1,-2,-1]] and its transpose (SAME zero padding inside each image);
 *   Gx, Gy = max(., epsilon)  (the reference clips from BELOW only: negative responses become epsilon and pass no gradient);
 *   cost = coef * mean_pixels( sqrt(sum_classes(Gx^2 + Gy^2)) * min(max(mask, 0), 1) ).
 * logits / mask / dlogits point at the FIRST UNLABELED row (n rows of h*w pixels); coef_dev is a 1-element device float.
 * dlogits (nullable) is overwritten (accumulate_grad = 0) or added to (1: the mean-teacher term already wrote these rows).
 * loss_io / accumulate_loss / cost_out as in ssmi_mt_consistency_fwd_bwd.  The workspace holds three [n*h*w, classes]
 * float planes (softmax and the two stencil gradients) plus the per-block partial sums; the reference's
 * settings.EPSILON is 1e-6. */
size_t ssmi_superpixel_workspace_bytes(int64_t npix, int32_t num_classes);
ssmi_status ssmi_superpixel_cost_fwd_bwd(const float* logits, const int32_t* mask, int32_t n, int32_t h, int32_t w,
                                         int32_t num_classes, const float* coef_dev, float grad_scale, float epsilon,
                                         float* dlogits, int32_t accumulate_grad, float* loss_io, int32_t accumulate_loss,
                                         float* cost_out, void* ws, size_t ws_bytes, void* stream);

/* Classification cost of the encoder-only models (src/losses.py:579-588 called from
 * classification_categorical_crossentropy_lambda_loss :636-652): y_true [n, classes] soft / multi-hot labels, weights
 * [n, classes]:  loss = -mean_n sum_c y * log(clip(softmax(logits), epsilon, 1)) * w   (settings.EPSILON = 1e-6);
 * dlogits (nullable) = grad_scale * dloss/dlogits; the clip passes no gradient below epsilon. */
ssmi_status ssmi_classification_xent_fwd_bwd(const float* logits, const float* y_true, const float* weights, int32_t n,
                                             int32_t num_classes, float epsilon, float grad_scale, float* dlogits,
                                             float* loss_out, void* stream);

/* Teacher weights as an exponential moving average of the student's (src/trainers.py:1692-1706):
 *   teacher[i] = fl32(fl32(a) * teacher[i]) + fl32(fl32(1 - a) * student[i])   (no fused multiply-add, like numpy)
 * applied to a whole arena; `1 - a` is formed in double before rounding, as the reference's Python float does. */
ssmi_status ssmi_ema_update(float* teacher, const float* student, double a, int64_t count, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Optimizers over a flat parameter arena (src/keras_extensions/extended_optimizers.py:35-67, 117-162).
 * `step` points at a 16-byte device block: an int64 holding Keras' `iterations` (value before the update) followed
 * by 8 bytes of scratch in which the library keeps the step's effective learning rate.  The call increments the
 * counter on the device, so a captured graph can be replayed.  lr_scale (nullable) is a per-ELEMENT table implementing lr_scalers.
 * grad_scale multiplies every gradient first (1/world_size after a sum all-reduce).
 * ---------------------------------------------------------------------------------------------------- */
ssmi_status ssmi_adam_step(float* params, const float* grads, float* m, float* v, float* vhat, int64_t count,
                           int64_t* step, float lr, float beta_1, float beta_2, float epsilon, float decay,
                           float grad_scale, const float* lr_scale, void* stream);
ssmi_status ssmi_sgd_step(float* params, const float* grads, float* moments, int64_t count, int64_t* step, float lr,
                          float momentum, float decay, int32_t nesterov, float grad_scale, const float* lr_scale,
                          void* stream);

/* The same updates with the hyper-parameters read from a DEVICE block of 8 floats instead of call arguments, so that a
 * step captured into a HIP graph follows what the reference's callbacks do to the optimizer between steps
 * (StepwiseLearningRateScheduler sets lr and beta_2 per step, src/callbacks/stepwise_learning_rate_scheduler.py:50-64;
 * ReduceLROnPlateau, src/trainers.py:1025-1043): the host rewrites the block, the graph is replayed unchanged.
 *   Adam: hyper = {lr, beta_1, beta_2, epsilon, decay, grad_scale, unused, unused}
 *   SGD : hyper = {lr, momentum, decay, nesterov (0 or 1), grad_scale, unused, unused, unused} */
ssmi_status ssmi_adam_step_hp(float* params, const float* grads, float* m, float* v, float* vhat, int64_t count,
                              int64_t* step, const float* hyper, const float* lr_scale, void* stream);
ssmi_status ssmi_sgd_step_hp(float* params, const float* grads, float* moments, int64_t count, int64_t* step,
                             const float* hyper, const float* lr_scale, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * SpatialDropout2D keep/scale table (src/models.py:1151,1267): mask[i] = (u_i >= rate) ? 1/(1-rate) : 0 with
 * u_i from Philox4x32-10 keyed by (seed, *step_counter, stream_id, i).  step_counter is a device int64 (read
 * only; the optimizer kernel advances it), so replayed graphs draw fresh masks every step.
 * ---------------------------------------------------------------------------------------------------- */
ssmi_status ssmi_dropout_mask(float* mask, int64_t count, float rate, uint64_t seed, const int64_t* step_counter,
                              uint32_t stream_id, void* stream);

/* Utility kernels. */
ssmi_status ssmi_fill_f32(float* dst, float value, int64_t count, void* stream);
/* Keras UpSampling2D((2,2)) (src/models.py:920,1287,1713): x[n,h,w,c] -> y[n,2h,2w,c]; y (forward) and dy (backward) may be
 * channel slices of a wider tensor (pixel stride y_ld / dy_ld), which is how `concatenate([up, skip])` (src/models.py:927)
 * is assembled without a separate pass over the upsampled half. */
ssmi_status ssmi_upsample_nn2x_fwd(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, float* y, int32_t y_ld,
                                   void* stream);
ssmi_status ssmi_upsample_nn2x_bwd(const float* dy, int32_t dy_ld, int32_t n, int32_t h, int32_t w, int32_t c, float* dx,
                                   void* stream);
/* dst[p, 0:c] (accumulate ? += : =) src[p, 0:c] for npix pixels with independent pixel strides: the skip half of a
 * concatenate (src/models.py:927) and the slices of its gradient. */
ssmi_status ssmi_copy_ld(const float* src, int32_t src_ld, float* dst, int32_t dst_ld, int64_t npix, int32_t c,
                         int32_t accumulate, void* stream);
/* SpatialDropout2D as its own layer (src/models.py:876): y[n,p,c] = x[n,p,c] * table[n,c]; the same call is its gradient. */
ssmi_status ssmi_scale_nc(const float* x, const float* table, float* y, int32_t n, int64_t hw, int32_t c, void* stream);
/* AveragePooling2D with the pool size of the whole map + Flatten, and GlobalAveragePooling2D (the heads of the
 * encoder-only models, src/models.py:985-993, 1040-1048, 1347-1354, 1667): y[n, c] = mean over hw pixels of x[n, p, c];
 * backward dx[n, p, c] = dy[n, c] / hw. */
ssmi_status ssmi_global_avgpool_fwd(const float* x, int32_t n, int64_t hw, int32_t c, float* y, void* stream);
ssmi_status ssmi_global_avgpool_bwd(const float* dy, int32_t n, int64_t hw, int32_t c, float* dx, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * bfloat16 twins of the entry points above whose activation tensors are not described by a descriptor.  Same
 * arguments and semantics as the function without the suffix (cited there); activation tensors are bfloat16 bit patterns
 * (uint16_t), everything else (statistics, parameters, tables, indices, partial sums) keeps its type.  Arithmetic is
 * float32; results are rounded to nearest even when stored.  Pooling compares the bfloat16 values it reads, so indices
 * are exact for the tensor as stored.
 * ---------------------------------------------------------------------------------------------------- */
ssmi_status ssmi_bn_train_stats_bf16(const uint16_t* x, int64_t npix, int32_t c, int32_t ld, float eps, float momentum,
                                     int32_t unbiased_moving, float* moving_mean, float* moving_var, float* save_mean,
                                     float* save_invstd, void* ws, size_t ws_bytes, void* stream);
ssmi_status ssmi_bn_act_fwd_bf16(const uint16_t* x, int32_t ld_x, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, int32_t act, const float* alpha, float slope, uint16_t* y, int64_t npix,
                                 int32_t c, void* stream);
ssmi_status ssmi_bn_act_bwd_bf16(const uint16_t* dy, const uint16_t* x, int32_t ld_x, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int32_t act, const float* alpha, float slope,
                                 uint16_t* dx, float* dgamma, float* dbeta, float* dalpha, int64_t npix, int32_t c, void* ws,
                                 size_t ws_bytes, void* stream);
ssmi_status ssmi_bn_act_drop_fwd_bf16(const uint16_t* x, int32_t ld_x, const float* mean, const float* invstd,
                                      const float* gamma, const float* beta, int32_t act, const float* alpha, float slope,
                                      const float* drop, int64_t hw, uint16_t* y, int32_t ld_y, int64_t npix, int32_t c,
                                      void* stream);
ssmi_status ssmi_bn_act_drop_bwd_bf16(const uint16_t* dy, int32_t ld_dy, const uint16_t* x, int32_t ld_x, const float* mean,
                                      const float* invstd, const float* gamma, const float* beta, int32_t act,
                                      const float* alpha, float slope, const float* drop, int64_t hw, uint16_t* dx,
                                      float* dgamma, float* dbeta, float* dalpha, int64_t npix, int32_t c, void* ws,
                                      size_t ws_bytes, void* stream);
ssmi_status ssmi_maxpool_argmax_fwd_bf16(const uint16_t* x, int32_t n, int32_t h, int32_t w, int32_t c, uint16_t* y,
                                         int32_t y_ld, int32_t* idx_i32, float* idx_f32, void* stream);
ssmi_status ssmi_maxpool_argmax_bwd_bf16(const uint16_t* dy, int32_t dy_ld, const int32_t* idx, int32_t n, int32_t h, int32_t w,
                                         int32_t c, uint16_t* dx, int32_t accumulate, void* stream);
ssmi_status ssmi_upsample_nn2x_fwd_bf16(const uint16_t* x, int32_t n, int32_t h, int32_t w, int32_t c, uint16_t* y, int32_t y_ld,
                                        void* stream);
ssmi_status ssmi_upsample_nn2x_bwd_bf16(const uint16_t* dy, int32_t dy_ld, int32_t n, int32_t h, int32_t w, int32_t c,
                                        uint16_t* dx, void* stream);
ssmi_status ssmi_copy_ld_bf16(const uint16_t* src, int32_t src_ld, uint16_t* dst, int32_t dst_ld, int64_t npix, int32_t c,
                              int32_t accumulate, void* stream);
ssmi_status ssmi_scale_nc_bf16(const uint16_t* x, const float* table, uint16_t* y, int32_t n, int64_t hw, int32_t c, void* stream);
ssmi_status ssmi_softmax_xent_fwd_bwd_bf16(const uint16_t* logits, const int32_t* labels, const float* weights, int64_t npix,
                                           int32_t num_classes, float grad_scale, uint16_t* dlogits, float* loss_out,
                                           int64_t* cm, int64_t npix_metrics, uint64_t ignore_mask, int32_t* pred_out,
                                           int32_t* bad_labels, void* ws, size_t ws_bytes, void* stream);
/* float32 <-> bfloat16 conversion of a dense tensor (round to nearest even): the images entering a bfloat16 network and the
 * logits leaving it for float32 consumers (src/generators.py:732-768 produces float32 batches; predict returns float32). */
ssmi_status ssmi_cast_f32_to_bf16(const float* src, uint16_t* dst, int64_t count, void* stream);
ssmi_status ssmi_cast_bf16_to_f32(const uint16_t* src, float* dst, int64_t count, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Input contract on the device (src/generators.py:711-770, 1787-1825): from the uint8 crops of a batch to the tensors
 * the training step consumes.
 *   images uint8 [n,h,w,3] -> X float32 = ((px / 255 - 0.5) * 2 - mean3) / std3   (mean3 / std3 nullable, already in
 *                             the normalised [-1,1] range; same float32 operation order as the reference)
 *   masks  uint8 [n,h,w] (red channel of the material mask; nullable: images only)
 *          -> Y int32: rows < num_labeled: r_to_class[r] (int32[256], -1 = colour without a class);
 *                      other rows: the raw value (src/generators.py:1806-1810)
 *          -> W float32: rows < num_labeled: class_weights[Y]; other rows: 1
 *   *bad (device int32, nullable) is incremented by the number of labeled pixels whose colour has no class (the
 *   reference raises KeyError there); such pixels get class 0 and weight 0.
 * ---------------------------------------------------------------------------------------------------- */
ssmi_status ssmi_preprocess_batch(const uint8_t* images, const uint8_t* masks, const int32_t* r_to_class,
                                  const float* class_weights, int32_t num_classes, int32_t num_labeled,
                                  const float* mean3, const float* std3, int32_t n, int32_t h, int32_t w, float* X,
                                  int32_t* Y, float* W, int32_t* bad, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Full-image inference around the network (src/predict.py:86-125 PredictionImage, :465-527 main).
 *   ssmi_predict_prepare   : uint8 image [h,w,3] -> X float32 [hp,wp,3]: the normalised image (same arithmetic as
 *                            ssmi_preprocess_batch) at rows v_pad_before.., columns h_pad_before..; every other pixel =
 *                            cval3 (device float[3], nullable = 0).  The reference pads after normalising, with the data
 *                            set's per-channel mean as fill (src/predict.py:115-124, src/utils/image_utils.py:1400-1434).
 *   ssmi_predict_accumulate: crops [v_pad_before : +h, h_pad_before : +w] out of a prediction [hp,wp,classes]
 *                            (src/predict.py:484-490), multiplies class 0 by bg_weight (:524) and writes (accumulate = 0)
 *                            or adds (accumulate = 1) it to acc [h,w,classes]; the sum runs in the order numpy's mean over
 *                            the stacked predictions uses (:527).
 *   ssmi_predict_topk      : mean_out (nullable) = acc / count; topk (nullable, int32 [k, npix]) = the k best classes per
 *                            pixel by descending mean (src/utils/prediction_utils.py:95-99); equal values rank the lower
 *                            class first.
 * ---------------------------------------------------------------------------------------------------- */
ssmi_status ssmi_predict_prepare(const uint8_t* image, const float* mean3, const float* std3, int32_t h, int32_t w,
                                 int32_t v_pad_before, int32_t h_pad_before, int32_t hp, int32_t wp, const float* cval3,
                                 float* X, void* stream);
ssmi_status ssmi_predict_accumulate(const float* pred, int32_t hp, int32_t wp, int32_t num_classes, int32_t v_pad_before,
                                    int32_t h_pad_before, int32_t h, int32_t w, float bg_weight, int32_t accumulate, float* acc,
                                    void* stream);
ssmi_status ssmi_predict_topk(const float* acc, int64_t npix, int32_t num_classes, int32_t count, int32_t k, float* mean_out,
                              int32_t* topk, void* stream);


/* ------------------------------------------------------------------------------------------------------
 * HIP graph capture of a whole training step (replaces the one tf.Session.run per step of
 * ExtendedModel.train_on_batch, src/keras_extensions/extended_model.py:1002).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct ssmi_graph ssmi_graph;
ssmi_status ssmi_graph_begin_capture(void* stream);
ssmi_status ssmi_graph_end_capture(void* stream, ssmi_graph** out);
ssmi_status ssmi_graph_launch(ssmi_graph* g, void* stream);
ssmi_status ssmi_graph_destroy(ssmi_graph* g);

/* ----------------------------------------------------------------------------------------------------
 * Collectives of the data-parallel step (src/trainers.py trains on ONE device; this is the exchange step of the one-process-
 * per-GPU trainer: gradient buckets summed over the ranks, int64 confusion matrices summed for sharded evaluation,
 * parameters broadcast from rank 0).  A thin layer over RCCL, looked up with dlopen at the first call (the process's own
 * copy is reused).  Rank 0 calls ssmi_comm_unique_id and hands the 128 bytes to the other ranks by any channel; every rank
 * then calls ssmi_comm_init with its device current.  The collectives are in place and asynchronous on `stream`.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct ssmi_comm ssmi_comm;
ssmi_status ssmi_comm_unique_id(char id[128]);
ssmi_status ssmi_comm_init(const char id[128], int32_t rank, int32_t nranks, ssmi_comm** out);
ssmi_status ssmi_comm_destroy(ssmi_comm* comm);
int32_t ssmi_comm_size(const ssmi_comm* comm);
int32_t ssmi_comm_rank(const ssmi_comm* comm);
ssmi_status ssmi_comm_allreduce_sum_f32(ssmi_comm* comm, float* buf, int64_t count, void* stream);
ssmi_status ssmi_comm_allreduce_sum_i64(ssmi_comm* comm, int64_t* buf, int64_t count, void* stream);
ssmi_status ssmi_comm_broadcast_f32(ssmi_comm* comm, float* buf, int64_t count, int32_t root, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * The 1x1 expansion of a regular bottleneck without its output tensor (src/models.py:1142-1172: Conv2D 1x1 ->
 * BatchNormalization -> SpatialDropout2D -> Add with the block input -> PReLU; training phase).
 * The three calls replace  ssmi_conv2d_fwd_bn_stats_prebn + ssmi_merge_fwd  and  ssmi_merge_bwd  (other_mode 0 / 1) for a
 * block whose expansion maps 16 -> 64 or 32 -> 128 channels: each recomputes y3 = PReLU(BN(x_raw)) * w from the raw
 * input of the BatchNormalization in front of the expansion (pre, pre_alpha: its final statistics, parameters and PReLU
 * slopes, as for ssmi_conv2d_fwd_bn_stats_prebn) instead of reading a stored y3, with the merge kernels' arithmetic per
 * element.  w: the expansion's kernel [cin][cout] (no bias).  ws: ssmi_expand_workspace_bytes(d).  skip: [n,h,w,c_skip], the
 * first c_skip channels of the skip branch (c_skip = cout: identity skip, other_mode 0; fewer: the max-pooled, channel-padded
 * skip of a downsampling block, other_mode 1 - the caller pools and routes dsum back through the pooling as before).
 *   ssmi_expand_bn_stats   training-phase statistics of the BatchNormalization behind the expansion (save_mean /
 *                          save_invstd, moving-average update); stores no activation
 *   ssmi_expand_merge_fwd  out = PReLU_alpha(drop * BN3(y3) + skip); y3_out (nullable): also store y3 (tests, debugging)
 *   ssmi_expand_merge_fwd_through  the same carried through the NEXT block's 1x1 reduction (src/models.py:1090-1100; kernel
 *                          w_next [cout][cin], no bias): y1_next [n,h,w,cin] = out * w_next and the training-phase statistics
 *                          of the BatchNormalization behind it (save_mean1 / save_invstd1, moving averages when given) - the
 *                          output tile is multiplied where it was formed; replaces the next block's
 *                          ssmi_conv2d_fwd_bn_stats, whose read of `out` it saves
 *   ssmi_expand_merge_fwd_through_infer  the inference-phase form (every BatchNormalization with fixed statistics, given as
 *                          mean / invstd like the others): a1_next [n,h,w,cin] = PReLU1'(BN1'(out * w_next)), the ACTIVATED input of
 *                          the next block's second convolution; replaces ssmi_conv2d_fwd_bn_add_act_infer of this block + the next
 *                          block's ssmi_conv2d_fwd_bn_act_infer.  (ssmi_expand_merge_fwd itself serves inference as it is.)
 *   ssmi_expand_merge_bwd  dsum = gradient of the pre-activation sum (= gradient of the skip branch), dmain = gradient of
 *                          y3, dgamma3 / dbeta3 / dalpha overwritten - as ssmi_merge_bwd
 *   ssmi_expand_merge_bwd_chained  the same with dout FORMED per tile instead of read: the block that follows a regular
 *                          bottleneck starts with a 1x1 reduction of that block's output (src/models.py:1090-1096), so
 *                          dout = dy_next * w_next^T + dsum_next (dy_next [n,h,w,cin]: the gradient at that reduction's output,
 *                          w_next [cout][cin]: its kernel, dsum_next [n,h,w,cout]: the following block's skip gradient) - the
 *                          transposed product has the expansion's shape, and dout is neither written nor read.  Replaces the
 *                          following block's ssmi_conv2d_bwd_data(add = dsum_next) + this block's ssmi_expand_merge_bwd.
 *   ssmi_expand_merge_bwd_through  either of the two (dout, or dy_next + w_next + dsum_next; the others NULL) carried through
 *                          the expansion's own data gradient and the backward of the BatchNormalization + PReLU in front of it:
 *                          dy2 [n,h,w,cin] = gradient at x_raw, dgamma2 / dbeta2 / dalpha2 (NULL without pre_alpha) overwritten.
 *                          The apply pass multiplies its dmain tile with w^T in place and takes BN2's column sums there; dmain
 *                          is still stored (the weight gradient of the expansion reads it).  Replaces ssmi_expand_merge_bwd +
 *                          ssmi_conv2d_bwd_data_bn_bwd of the expansion.
 *                          (dy2 NULL with wpart1_next or next_apply given: the plain apply pass, dmain required.)
 *                          next_apply (nullable, chained form only): see ssmi_expand_next_apply.
 *                          wpart3 / wpart1_next (each nullable, ssmi_expand_weight_partial_bytes(d) bytes, 16-byte aligned):
 *                          the weight gradients of the two 1x1 layers whose operands the passes hold in LDS anyway, as
 *                          per-workgroup partial matrices - wpart3: of the expansion itself (dW3 = a2^T * dmain; dmain may then
 *                          be NULL and is not stored), wpart1_next (needs dy_next): of the following block's reduction
 *                          (dW1' = out^T * dy_next; out, this block's output, is recomputed by the reducing pass).
 *                          ssmi_expand_weight_reduce(d, wpart, dw) sums a partial buffer in a fixed order into the layer's
 *                          gradient ([cin][cout] for wpart3, [cout][cin] for wpart1_next; any stream ordered behind the call).
 *                          Replace ssmi_conv2d_bwd_weight of these layers.
 * Outputs equal the stored route's up to the order of the float32 sums inside y3 (and inside dout, da2, dW). */
/* ssmi_expand_merge_bwd_through, chained form: the backward of the BatchNormalization + PReLU behind the next block's
 * reduction has stopped after its reduction (ssmi_conv2d_bwd_data_bn_bwd_rot_deferred: coef = k1 | m1 | m2, 3 * cin floats) and
 * dy_next still holds the gradient behind them; the reducing pass runs their apply pass (ssmi_bn_act_bwd's arithmetic) on the
 * operand as it stages it and writes the result back to dy_next in place (the reduction's weight gradient reads it there). */
typedef struct ssmi_expand_next_apply {
  const void* y1;               /* [n,h,w,cin] raw output of the next block's reduction (activation type) */
  const ssmi_bn_params* bn1;    /* its BatchNormalization */
  const float* alpha;           /* PReLU slopes [cin], or NULL */
  const float* coef;            /* 3 * cin floats */
} ssmi_expand_next_apply;
typedef struct ssmi_expand_desc {
  int32_t n, h, w, cin, cout;
  int32_t dtype;   /* SSMI_F32 (0) or SSMI_BF16: element type of x_raw, skip, out, dout, dsum, dmain, y3_out */
} ssmi_expand_desc;
int32_t ssmi_expand_supported(const ssmi_expand_desc* d);
size_t ssmi_expand_workspace_bytes(const ssmi_expand_desc* d);
ssmi_status ssmi_expand_bn_stats(const ssmi_expand_desc* d, const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha,
                                 const float* w, float eps, float momentum, int32_t unbiased_moving, float* moving_mean,
                                 float* moving_var, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, void* stream);
ssmi_status ssmi_expand_merge_fwd(const ssmi_expand_desc* d, const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha,
                                  const float* w, const ssmi_bn_params* bn3, const float* drop, const void* skip, int32_t c_skip,
                                  const float* alpha, void* out, void* y3_out, void* stream);
ssmi_status ssmi_expand_merge_fwd_through(const ssmi_expand_desc* d, const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha,
                                          const float* w, const ssmi_bn_params* bn3, const float* drop, const void* skip, int32_t c_skip,
                                          const float* alpha, void* out, void* y3_out, const float* w_next, void* y1_next, float eps,
                                          float momentum, int32_t unbiased_moving, float* moving_mean1, float* moving_var1,
                                          float* save_mean1, float* save_invstd1, void* ws, size_t ws_bytes, void* stream);
ssmi_status ssmi_expand_merge_fwd_through_infer(const ssmi_expand_desc* d, const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha,
                                                const float* w, const ssmi_bn_params* bn3, const float* drop, const void* skip,
                                                int32_t c_skip, const float* alpha, void* out, const float* w_next,
                                                const ssmi_bn_params* bn1_next, const float* alpha1_next, void* a1_next, void* stream);
ssmi_status ssmi_expand_merge_bwd(const ssmi_expand_desc* d, const void* dout, const void* x_raw, const ssmi_bn_params* pre,
                                  const float* pre_alpha, const float* w, const ssmi_bn_params* bn3, const float* drop, const void* skip,
                                  int32_t c_skip, const float* alpha, void* dmain, void* dsum, float* dgamma3, float* dbeta3, float* dalpha, void* ws,
                                  size_t ws_bytes, void* stream);
ssmi_status ssmi_expand_merge_bwd_chained(const ssmi_expand_desc* d, const void* dy_next, const float* w_next, const void* dsum_next,
                                          const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha, const float* w,
                                          const ssmi_bn_params* bn3, const float* drop, const void* skip, int32_t c_skip, const float* alpha,
                                          void* dmain, void* dsum, float* dgamma3, float* dbeta3, float* dalpha, void* ws, size_t ws_bytes,
                                          void* stream);
ssmi_status ssmi_expand_merge_bwd_through(const ssmi_expand_desc* d, const void* dout, const void* dy_next, const float* w_next,
                                          const void* dsum_next, const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha,
                                          const float* w, const ssmi_bn_params* bn3, const float* drop, const void* skip, int32_t c_skip,
                                          const float* alpha, void* dmain, void* dsum, float* dgamma3, float* dbeta3, float* dalpha,
                                          void* dy2, float* dgamma2, float* dbeta2, float* dalpha2, float* wpart3, float* wpart1_next,
                                          const ssmi_expand_next_apply* next_apply, void* ws, size_t ws_bytes, void* stream);
size_t ssmi_expand_weight_partial_bytes(const ssmi_expand_desc* d);
ssmi_status ssmi_expand_weight_reduce(const ssmi_expand_desc* d, const float* wpart, float* dw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSMI_H_ */

/* libcae_hip.so -- C ABI of the MI355X-native compress/decompress hot path.
 *
 * The reference (TheJacksonLaboratory/cnn_autoencoder) is pure Python: it has no FFI of its
 * own.  Its native code on this path lives in the third-party package `compressai`
 * (C++ pybind11: `_CXX.pmf_to_quantized_cdf`, `ans.RansEncoder/RansDecoder`) and in the
 * ATen/cuDNN kernels behind `nn.Conv2d` / `nn.ConvTranspose2d` / `GDN`.  Each entry point
 * below names the reference interface (file:line under /root/reference/src) it replaces;
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, a negative cae_status otherwise;
 *     cae_last_error() returns a thread-local message for the last failure;
 *   - `*_dev` pointers are DEVICE pointers (HBM) of the current HIP device, `stream` is a
 *     hipStream_t passed as void* (NULL = default stream); all launches are asynchronous;
 *   - `*_host` pointers are host memory;
 *   - outputs are caller-allocated, except the variable-length bitstreams returned by
 *     cae_rans_encode_batch (library-allocated, release with cae_free);
 *   - a model handle may be used from several host threads (dask's threaded scheduler calls
 *     codec.encode concurrently, compress.py:121-128): entry points taking a handle
 *     serialise on a per-handle mutex.
 *   - no torch types anywhere in this ABI.
 */
#ifndef CAE_HIP_H
#define CAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cae_model cae_model_t;

enum cae_status {
    CAE_OK = 0,
    CAE_ERR_ARG = -1,      /* bad argument (the reference raises ValueError) */
    CAE_ERR_HIP = -2,      /* HIP runtime failure */
    CAE_ERR_NOMEM = -3,
    CAE_ERR_UNSUPPORTED = -4,
    CAE_ERR_CORRUPT = -5   /* bitstream ran past its end */
};

enum cae_track { CAE_ANALYSIS = 0, CAE_SYNTHESIS = 1 };
enum cae_pixfmt {
    CAE_FMT_U8_HWC = 0,  /* (n,h,w,c) uint8  -- zarr chunk layout, compress.py:101 */
    CAE_FMT_F32_NCHW = 1 /* (n,c,h,w) float  -- nn.Module call surface */
};

int cae_version(void);
const char *cae_last_error(void);
void cae_free(void *p);

/* ---- model lifecycle ------------------------------------------------------------------
 * Replaces setup_modules / load_state_dict (_autoencoders.py:458-502): the Python side reads
 * the checkpoint dict and hands every layer's tensors over in the reference's own layouts. */
int cae_model_create(int channels_org, int channels_net, int channels_bn, int compression_level,
                     int kernel_size, cae_model_t **out);
void cae_model_destroy(cae_model_t *m);

/* One unit of a track: strided conv (analysis; DownsamplingUnit model.0, _autoencoders.py:78-85;
 * weight (cout,cin,k,k)) or transposed conv (synthesis; UpsamplingUnit model.0, :204-211;
 * weight (cin,cout,k,k)), optional bias (cout), optional GDN/IGDN (model.1, :29-30) given as
 * the EFFECTIVE (re-parametrised, non-negative) beta (cout) and gamma (cout,cout). */
int cae_model_set_layer(cae_model_t *m, int track, int index, int cin, int cout,
                        const float *weight_host, const float *bias_host,
                        const float *beta_eff_host, const float *gamma_eff_host);

/* LeakyReLU / ReLU variants of a unit (DownsamplingUnit _autoencoders.py:62-76,90-92; UpsamplingUnit
 * :187-202,216-218): `act` (0 none, 1 LeakyReLU(0.01), 2 ReLU) is applied after the layer's strided
 * (transposed) convolution; when pre_weight_host is given, a stride-1 convolution cin -> cin (analysis:
 * weight (cin,cin,k,k), reflect padding; synthesis: ConvTranspose2d weight (cin,cin,k,k), padding k//2)
 * plus the same activation runs in front of it.  Call after cae_model_set_layer for the same index.
 * Both arithmetic paths (f16x3: the same split-f16 kernels with stride 1 / the activation in the store epilogue). */
int cae_model_set_layer_act(cae_model_t *m, int track, int index, int act, const float *pre_weight_host,
                            const float *pre_bias_host);

/* Arithmetic of the conv / GDN contraction: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32),
 * 1 = "f16x3": every operand split into two f16 halves, three f16 MFMAs per product, fp32
 * accumulate (22 significant bits; same 1e-4 parity bar, ~5x less matrix-pipe time).  Set before
 * cae_model_set_layer.  (No reference counterpart: the reference computes in fp32 on ATen,
 * _autoencoders.py:78-85, :204-211.)
 *
 * VALID RANGE of f16x3.  f16 has 5 exponent bits, so the split format holds finite values with
 * |v| <= 65504 at 22 significant bits (values below 2^-14 keep an absolute accuracy of 2^-25).
 *   - weights / gamma outside that range: cae_model_set_layer marks the model and every call runs on the
 *     fp32 kernels (cae_model_effective_precision reports 0);
 *   - GDN / IGDN squares: formed from y scaled per pixel by a power of two (largest |y| of the pixel's
 *     channels brought into [64,128)), beta and the result rescaled exactly: any finite pre-GDN magnitude
 *     is in range; accuracy is 2^-22 relative to the pixel's largest channel;
 *   - activations between layers, latents into the synthesis track and float inputs: a value with
 *     |v| > 65504 (or NaN) cannot be stored.  Every kernel that writes the split format then raises the
 *     call's overflow word, and the results of that call are INVALID.  Protocol: after the call, on the
 *     same host thread, t = cae_last_range_ticket(); once the stream work of the call has completed
 *     (stream / event synchronise), cae_range_check(m, t, &over); if over, repeat the call between
 *     cae_thread_force_fp32(1) / (0): it then runs on the exact-fp32 kernels (fp32 range, as the reference).
 *     The Python modules do this automatically.  1024 calls per handle are tracked. */
int cae_model_set_precision(cae_model_t *m, int precision);
int64_t cae_last_range_ticket(void);         /* thread-local; 0 = the call ran on the fp32 kernels: nothing to check */
int cae_range_check(cae_model_t *m, int64_t ticket, int *overflowed);
void cae_thread_force_fp32(int on);          /* thread-local: calls of this thread use the fp32 kernels */
int cae_model_effective_precision(cae_model_t *m, int *precision);

/* Integer tables of the factorized entropy model, as EntropyBottleneck.update() leaves them
 * (_autoencoders.py:502): quantized_cdf (channels, cdf_stride) int32, cdf_length (channels),
 * offset (channels), medians (channels) float. */
int cae_model_set_entropy(cae_model_t *m, int channels, int cdf_stride, const int32_t *quantized_cdf_host,
                          const int32_t *cdf_length_host, const int32_t *offset_host,
                          const float *medians_host);

/* ---- device hot path --------------------------------------------------------------------
 * cae_analysis replaces Analyzer.forward (_autoencoders.py:359-361) including the uint8 ->
 * float/255 conversion of ConvolutionalAutoencoder.encode (:542-545) when fmt is U8_HWC.
 * latents_dev: (n, channels_bn, ceil(h/2^L), ceil(w/2^L)) float NCHW. */
int cae_analysis(cae_model_t *m, const void *tiles_dev, int fmt, int n, int h, int w,
                 float *latents_dev, void *stream);

/* cae_synthesis replaces Synthesizer.forward (_autoencoders.py:442-455) and, for U8_HWC, the
 * x*255 -> clip -> truncating uint8 -> HWC epilogue of ConvolutionalAutoencoder.decode
 * (:576-580).  latents_dev (n, channels_bn, lh, lw) float NCHW; output (n, ., lh*2^L, lw*2^L).
 * bridges_dev: NULL, or an array of compression_level-1 device pointers that receive the
 * intermediate features fx_brg[i] (n, channels_net, lh*2^(i+1), lw*2^(i+1)) float NCHW. */
int cae_synthesis(cae_model_t *m, const float *latents_dev, int n, int lh, int lw,
                  void *out_dev, int fmt, float *const *bridges_dev, void *stream);

/* Residual units (ResidualDownsamplingUnit / ResidualUpsamplingUnit, _autoencoders.py:104-174, :230-304) and, in
 * general, the stride-1 (transposed) convolutions cin -> cin in front of a unit's strided layer.  Stage `stage`
 * (0 or 1) computes  y = post_act( act_or_gdn(conv(x_stage)) [+ unit input] ):  w (cin,cin,k,k) with bias or NULL
 * (BatchNorm folded by the caller), beta/gamma = effective GDN parameters of that stage or NULL, act / post_act as
 * in cae_model_set_layer_act.  cae_model_set_layer_act clears the stages of its layer (and creates stage 0 from
 * its pre-convolution), so call it first.  Precision 1 (f16x3): GDN / residual stages up to 128 channels run
 * on conv_s2_f16_kernel<.., S = 1, RES>; the stages of a wider unit run on the fp32 kernels between two layout
 * conversions (the strided layers stay on f16x3). */
int cae_model_set_layer_stage(cae_model_t *m, int track, int index, int stage, const float *w, const float *bias,
                              const float *beta, const float *gamma, int act, int add_residual, int post_act);

/* Multiscale colour layers (Synthesizer(multiscale_analysis=True), _autoencoders.py:417-436): a stride-1 reflect
 * convolution from the output of synthesis level `index` (< compression_level-1) to the image channels.
 * w: (cout, cin, k, k).  cae_synthesis_multiscale additionally writes colors_dev[i] (n, cout, lh*2^(i+1),
 * lw*2^(i+1)) float NCHW for every non-NULL entry: the reference's x_r[compression_level-1-i]
 * (_autoencoders.py:446-452).  Precision 1 (f16x3): colour layers to at most 32 channels; wider: precision 0. */
int cae_model_set_color_layer(cae_model_t *m, int index, int cin, int cout, const float *w, const float *bias);
int cae_synthesis_multiscale(cae_model_t *m, const float *latents_dev, int n, int lh, int lw, void *out_dev, int fmt,
                             float *const *bridges_dev, float *const *colors_dev, void *stream);

/* Fused forms of the codec path (encode: _autoencoders.py:542-551, decode: :568-580): the quantiser runs in the
 * epilogue of the last analysis layer (symbols (n, channels_bn, lh, lw) int32 instead of float latents) and the
 * dequantiser in the layout conversion in front of the first synthesis layer.  Same results as
 * cae_analysis + cae_quantize / cae_dequantize + cae_synthesis; the handle needs cae_model_set_entropy (medians). */
int cae_analysis_symbols(cae_model_t *m, const void *tiles_dev, int fmt, int n, int h, int w, int32_t *symbols_dev,
                         void *stream);
int cae_synthesis_symbols(cae_model_t *m, const int32_t *symbols_dev, int n, int lh, int lw, void *out_dev, int fmt,
                          void *stream);

/* One GDN / IGDN layer on an NCHW tensor (compressai.layers.GDN.forward; reference call site
 * _autoencoders.py:29-30).  Uses the beta/gamma of layer `index` of `track`. */
int cae_gdn_forward(cae_model_t *m, int track, int index, const float *x_dev, int n, int h, int w,
                    float *y_dev, void *stream);

/* EntropyBottleneck quantiser (compress(): symbols = round(y - median_c).int(), reference call
 * site _autoencoders.py:549-551; decompress(): y_hat = symbols + median_c, :568-571).
 * count = n * channels * hw elements in NCHW order, hw = spatial size per channel. */
int cae_quantize(cae_model_t *m, const float *latents_dev, int n, int hw, int32_t *symbols_dev, void *stream);
int cae_dequantize(cae_model_t *m, const int32_t *symbols_dev, int n, int hw, float *latents_dev, void *stream);

/* The same quantiser writing the symbols straight into PINNED HOST memory (a device-accessible host
 * pointer, e.g. hipHostMalloc / a pinned torch tensor) from at most `max_blocks` workgroups: the PCIe
 * transfer then runs beside the compute kernels of another stream on a few CUs.  (A hipMemcpyAsync D2H
 * here ran as a blit kernel that filled every CU's wave slots for the 1.8 ms the 100 MB take to cross
 * PCIe and stalled the next layer's kernels for that long; profiles/r01_experiments.md.)  The caller
 * synchronises `stream` (or an event on it) before reading the symbols on the host. */
int cae_quantize_export(cae_model_t *m, const float *latents_dev, int n, int hw, int32_t *symbols_pinned_host,
                        int max_blocks, void *stream);

/* Density network of the factorized prior (compressai EntropyBottleneck parameters `_matrix{i}`,
 * `_bias{i}`, `_factor{i}`, i = 0..n_filters; the reference builds it at _autoencoders.py:476-477 with
 * filters = [r]*K).  The caller passes EFFECTIVE values: matrices[i] = softplus(_matrix{i})
 * (channels, F[i+1], F[i]), biases[i] (channels, F[i+1]), factors[i] = tanh(_factor{i})
 * (channels, F[i+1]; i < n_filters), F = (1, filters..., 1).  likelihood_bound = 1e-9 in the
 * reference (0 disables the lower bound).  Hidden widths up to 8. */
int cae_model_set_density(cae_model_t *m, int channels, int n_filters, const int *filters,
                          const float *const *matrices, const float *const *biases,
                          const float *const *factors, float likelihood_bound);

/* Form of the likelihood p = c(y + 1/2) - c(y - 1/2) in floating point: 0 (default) = sigmoid(u) - sigmoid(l), as
 * compressai >= 1.2.x (`_likelihood` returning (likelihood, lower, upper), the version range the reference requires,
 * requirements.txt:25) is believed to compute it; 1 = the older sign trick |sigmoid(s u) - sigmoid(s l)|,
 * s = -sign(l + u).  Equal in exact arithmetic; in fp32 they differ in the last unit in the upper tail. */
int cae_model_set_likelihood_form(cae_model_t *m, int form);

/* EntropyBottleneck.__call__ in eval mode (reference call site models/tasks/_taskutils.py:97 and
 * the rate term -sum(log2 p) of models/criteria/_ratedist.py:49-54): for latents (n, channels, hw)
 *   y_hat = round(y - median_c) + median_c,
 *   likelihood = max(sigmoid(u) - sigmoid(l), bound), l,u = logits_cumulative(y_hat -/+ 0.5)
 *                (form: cae_model_set_likelihood_form),
 *   bits[i] = -sum log2(likelihood) over tile i (float64, deterministic order).
 * Any of y_hat_dev / likelihood_dev / bits_dev may be NULL.  Needs set_entropy (medians) and
 * set_density.  Calls on one handle must be stream-ordered. */
int cae_likelihood(cae_model_t *m, const float *latents_dev, int n, int hw, float *y_hat_dev,
                   float *likelihood_dev, double *bits_dev, void *stream);

/* Mean structural similarity of two (n, h, w, c) uint8 HWC batches -> ssim_dev[n] (float64): the algorithm of
 * skimage.metrics.structural_similarity(x, x_r, channel_axis=2) as the reference's metrics harness calls it
 * (test_cae.py:55-57): 7x7 uniform window, sample covariance, K1 = 0.01, K2 = 0.03, data range 255, 3-pixel border
 * cropped, mean over channels.  workspace_dev: n * ceil((h-6)/32) * ceil((w-6)/32) doubles. */
int cae_tile_ssim(const uint8_t *a_dev, const uint8_t *b_dev, int n, int h, int w, int c, double *ssim_dev,
                  double *workspace_dev, size_t workspace_elems, void *stream);

/* Mean CIE76 colour difference of two (n, pixels, 3) uint8 RGB batches -> delta_dev[n] (float64): skimage's
 * rgb2lab (sRGB, D65, 2-degree observer) + deltaE_cie76 + mean, as compute_deltaCIELAB does (test_cae.py:21-45).
 * workspace_dev: n * min(ceil(pixels/256), 128) doubles. */
int cae_tile_delta_e(const uint8_t *a_dev, const uint8_t *b_dev, int n, size_t pixels, double *delta_dev,
                     double *workspace_dev, size_t workspace_elems, void *stream);

/* Building blocks of pytorch_msssim.ms_ssim(x_r, x, data_range=255) as compute_ms_ssim calls it (test_cae.py:47-52):
 * uint8 HWC tiles -> planar float32 (n*c planes of h x w); one scale of the index on planar images: ssim_cs_dev[2p] =
 * mean ssim map, [2p+1] = mean contrast-structure map of plane p (11-tap window `window11_dev` applied separably without
 * padding, K = (0.01, 0.03), data range 255; workspace 2 * planes * ceil((h-10)/32) * ceil((w-10)/32) doubles); and the
 * 2x2 average pooling between scales (zero padding of odd sizes, out = floor((s + 2(s%2) - 2)/2) + 1 per side). */
int cae_u8hwc_to_planes(const uint8_t *tiles_dev, int n, int h, int w, int c, float *planes_dev, void *stream);
int cae_avgpool2(const float *in_dev, int planes, int h, int w, float *out_dev, void *stream);
int cae_msssim_level(const float *x_dev, const float *y_dev, int planes, int h, int w, const float *window11_dev,
                     double *ssim_cs_dev, double *workspace_dev, size_t workspace_elems, void *stream);

/* Blocking device -> pinned-host copy on the DMA engines (hsa_amd_memory_async_copy), for symbols on their way
 * to the host coder.  The caller has already waited for the kernels that produce `src_dev` (event / stream
 * synchronise); safe to call from any host thread.  hipMemcpyAsync is not used because the HIP runtime bundled
 * with PyTorch-ROCm 7.0 runs D2H copies as a blit kernel that occupies the CUs for the whole PCIe transfer. */
int cae_copy_to_host(void *dst_pinned_host, const void *src_dev, size_t bytes);

/* Per-tile sum of squared differences of two (n, elems) uint8 batches -> sse_dev[n] (float64).
 * The distortion half of the per-tile statistics record the slide driver all-gathers (the
 * reference's harness computes MSE/PSNR on the host, test_cae.py:55-68). */
int cae_tile_sse(const uint8_t *a_dev, const uint8_t *b_dev, int n, size_t elems, double *sse_dev, void *stream);

/* ---- measurement ------------------------------------------------------------------------
 * With profiling on, cae_analysis / cae_synthesis bracket every kernel they launch with HIP
 * events on the caller's stream.  cae_model_get_profile synchronises those events and ADDS the
 * elapsed milliseconds of every call since the last reset into ms[0..n_slots): slot 0 is the
 * input layout conversion, slot 1+i the fused kernel of layer i (conv/deconv + bias + GDN).
 * calls receives the number of profiled calls.  (No reference counterpart: the reference only
 * takes perf_counter wall times, test_cae.py:101-115.) */
int cae_model_set_profiling(cae_model_t *m, int enable);
int cae_model_get_profile(cae_model_t *m, int track, double *ms, int n_slots, int *calls, int reset);

/* ---- training (BASELINE config 5: bf16 convolutions, fp32 GDN) ----------------------------
 * What `loss.backward()` (train_cae_ms.py:214) differentiates on the reference: nn.Conv2d(reflect, stride 2) /
 * nn.ConvTranspose2d(stride 2, output_padding 1) (_autoencoders.py:78-85, :204-211) and compressai GDN / IGDN (:29-30).
 * Stateless launches on caller-owned device buffers.  Layout "T": channels-last [n][h][w][cp], cp = channels padded
 * to a multiple of 32 (<= 192) with zeros; *16 = bf16, *32 = fp32.  Convolutions take bf16 operands and accumulate in
 * fp32 (v_mfma_f32_32x32x16_bf16); GDN is exact fp32 (v_mfma_f32_32x32x2_f32).
 * Packed weights: MFMA B fragments of a (dim0, dim1, k, k) fp32 tensor with dimension `contract_dim` contracted:
 *   nn.Conv2d weight (cout, cin, k, k):          forward contracts dim 1, data gradient dim 0;
 *   nn.ConvTranspose2d weight (cin, cout, k, k): forward contracts dim 0, data gradient dim 1. */
size_t cae_t_packed_bytes(int contract_channels, int out_channels, int kernel_size);
int cae_t_pack_weights(const float *w_dev, int dim0, int dim1, int kernel_size, int contract_dim, void *packed_dev, void *stream);
int cae_t_from_nchw(const float *x_nchw_dev, int n, int c, int h, int w, int cp, void *out16, float *out32, void *stream);
int cae_t_to_nchw(const float *t32, int n, int c, int h, int w, int cp, float *out_nchw_dev, void *stream);
/* z = conv(x) (+bias): x16 [n][h][w][cin_p] -> z [n][ceil(h/2)][ceil(w/2)][cout_p] as fp32 and / or bf16 */
int cae_t_conv_forward(const void *x16, int n, int h, int w, int cin_p, const void *packed, int kernel_size, float *z32,
                       void *z16, int cout_p, const float *bias, void *stream);
/* data gradient of that convolution on the EXTENDED domain: gext32 [n][h+2P][w+2P][cin_p], P = k//2, the gradient with
 * respect to the reflect-PADDED input; its consumer folds the border back (cae_t_gdn_backward / cae_t_fold_to_bf16) */
int cae_t_conv_dgrad_ext(const void *gz16, int n, int oh, int ow, int cout_p, const void *packed, int kernel_size, int h,
                         int w, float *gext32, int cin_p, void *stream);
/* z = conv_transpose(x) (+bias): x16 [n][h][w][cin_p] -> z [n][2h][2w][cout_p] */
int cae_t_deconv_forward(const void *x16, int n, int h, int w, int cin_p, const void *packed, int kernel_size, float *z32,
                         void *z16, int cout_p, const float *bias, void *stream);
/* its data gradient: gz16 [n][2h][2w][cout_p] -> gx [n][h][w][cin_p] */
int cae_t_deconv_dgrad(const void *gz16, int n, int h, int w, int cout_p, const void *packed, int kernel_size, float *gx32,
                       void *gx16, int cin_p, void *stream);
/* weight gradient of either layer: gw32 [k*k][ca][cb] = sum over positions of xbig[2 pos + tap - P][a] * ysmall[pos][b].
 * conv: xbig = layer input (reflect = 1), ysmall = output gradient -> grad(cout,cin,k,k)[b][a][tap];
 * conv_transpose: xbig = output gradient (reflect = 0), ysmall = layer input -> grad(cin,cout,k,k)[b][a][tap]. */
int cae_t_wgrad(const void *xbig16, int n, int h, int w, int ca, const void *ysmall16, int oh, int ow, int cb,
                int kernel_size, int reflect, float *gw32, void *stream);
/* GDN / IGDN forward on effective parameters padded to cp (beta padding 1, gamma padding 0): y = z * rsqrt(beta +
 * gamma z^2) (inverse: sqrt). */
int cae_t_gdn_forward(const float *z32, long pixels, int cp, const float *beta, const float *gamma, int inverse, float *y32,
                      void *y16, void *stream);
/* GDN / IGDN backward.  gext32: gradient with respect to the layer output, fp32 [n][h+2 pad][w+2 pad][cp] (pad > 0: the
 * extended-domain gradient of the next convolution, folded here; pad = 0: plain).  gamma_t = gamma transposed.
 * Outputs: gz (bf16 and / or fp32) with respect to z, ggamma [cp][cp], gbeta [cp] with respect to the EFFECTIVE
 * parameters (the reparametrisation and its LowerBound gradient rule live above the ABI); gn_ws32 / gzd_ws32: scratch,
 * pixels * cp floats each. */
int cae_t_gdn_backward(const float *z32, const float *gext32, int n, int h, int w, int pad, int cp, const float *beta,
                       const float *gamma, const float *gamma_t, int inverse, float *gn_ws32, float *gzd_ws32, float *gz32,
                       void *gz16, float *ggamma, float *gbeta, void *stream);
/* LeakyReLU / ReLU units under autograd (DownsamplingUnit _autoencoders.py:62-76,90-92; UpsamplingUnit :187-202,216-218):
 * the strided layers with the activation (1 LeakyReLU(0.01), 2 ReLU) in their epilogue; the stride-1 pre-convolutions as
 * cae_t_corr_s1 -- mode 0 analysis forward (reflect), 1 its data gradient on the extended domain (h + 2P) x (w + 2P),
 * 2 synthesis forward (ConvTranspose2d stride 1, padding k//2), 3 its data gradient; the packed weights carry the
 * contraction (cae_t_pack_weights: contract_dim 1 for modes 0 and 3, 0 for modes 1 and 2) --, their weight gradient
 * cae_t_wgrad_s1 (x and y of equal size; analysis: x = input, reflect; synthesis: x = output gradient, zeros, y = input),
 * and the activation's backward cae_t_act_backward: out = g * (y > 0 ? 1 : slope) with y the activation's OUTPUT and g
 * either bf16 (g16) or the fp32 extended-domain gradient gext32, whose reflect fold is applied in place first. */
int cae_t_conv_forward_act(const void *x16, int n, int h, int w, int cin_p, const void *packed, int ks, float *z32, void *z16,
                           int cout_p, const float *bias, int act, void *stream);
int cae_t_deconv_forward_act(const void *x16, int n, int h, int w, int cin_p, const void *packed, int ks, float *z32, void *z16,
                             int cout_p, const float *bias, int act, void *stream);
int cae_t_corr_s1(const void *x16, int n, int h, int w, int ck, const void *packed, int ks, int mode, float *out32, void *out16,
                  int cn, const float *bias, int act, void *stream);
int cae_t_wgrad_s1(const void *x16, int n, int h, int w, int ca, const void *y16, int cb, int ks, int reflect, float *gw32,
                   void *stream);
int cae_t_act_backward(const void *g16, float *gext32, int pad, const void *y16, int n, int h, int w, int cp, int act,
                       void *out16, void *stream);

/* Edge layers with <= 3 image channels as pointwise GEMMs over K = (tap, channel) <= 32 (instead of padding 3 channels to
 * 32): cae_t_im2col_s2 gathers the stride-2 taps of an NCHW fp32 tensor into [n][oh][ow][32] bf16 (j = tap * c + channel;
 * reflect = 1: the first analysis layer's input (nn.Conv2d reflect, _autoencoders.py:78-85); 0: zeros outside -- the last
 * synthesis layer's output gradient, oh = h / 2); cae_t_pointwise / cae_t_wgrad_pointwise are the 1 x 1 gather-GEMM and
 * its weight gradient on channels-last bf16 tensors; cae_t_col2im_s2 sums the <= 4 products u[pos][(tap, channel)] of every
 * output pixel of ConvTranspose2d(k, 2, k//2, output_padding 1) (:204-211) into NCHW fp32 (+ bias). */
int cae_t_im2col_s2(const float *x_nchw, int n, int c, int h, int w, int oh, int ow, int ks, int reflect, void *out16,
                    void *stream);
int cae_t_col2im_s2(const float *u32, const float *bias, int n, int c, int h, int w, int ks, float *out_nchw, void *stream);
int cae_t_pointwise(const void *x16, int n, int h, int w, int ck, const void *packed, float *out32, void *out16, int cn,
                    const float *bias, int act, void *stream);
int cae_t_wgrad_pointwise(const void *x16, const void *y16, int n, int h, int w, int ca, int cb, float *gw32, void *stream);

/* Fused forms (csrc/cae_train_gdn.hpp; cp <= 128): the forward also saves the per-element factor f (y = z f: n^(-1/2),
 * IGDN n^(1/2)) in the register order the backward reads back -- f_saved holds cae_t_gdn_saved_elems(pixels, cp) floats
 * (0: shape not built) -- and the backward is ONE kernel: g_z (bf16), g_gamma, g_beta from z, f and the gradient with
 * respect to y (extended domain with padding `pad`; its reflect fold is applied to gext32 IN PLACE first), without
 * recomputing the norm. */
size_t cae_t_gdn_saved_elems(long pixels, int cp);
int cae_t_gdn_forward_save(const float *z32, long pixels, int cp, const float *beta, const float *gamma, int inverse,
                           void *y16, float *f_saved, void *stream);
int cae_t_gdn_backward_fused(const float *z32, const float *f_saved, float *gext32 /* folded in place */, int n, int h, int w, int pad, int cp,
                             const float *gamma, int inverse, void *gz16, float *ggamma, float *gbeta, void *stream);
int cae_t_fold_to_bf16(const float *gext32, int n, int h, int w, int pad, int cp, void *out16, void *stream);

/* nn.BatchNorm2d in TRAINING mode (batch statistics; the units' optional batch norm, _autoencoders.py:72-73, :87-88, and
 * its backward as torch autograd derives it) on NCHW fp32 tensors (n, c, hw).  Both directions are a pair of per-channel
 * moments and a per-channel affine map:
 *   cae_t_bn_moments   s1[c] = sum a, s2[c] = sum a b   over (n, hw), accumulated in double
 *                      (forward: a = b = x; backward: a = dy, b = x)
 *   cae_t_bn_affine    out = a A[c] + (b ? b B[c] : 0) + C[c]
 *                      (forward: y = x w rstd + (bias - mean w rstd); backward: dx = dy A + x B + C) */
int cae_t_bn_moments(const float *a, const float *b, int n, int c, long hw, double *s1, double *s2, void *stream);
int cae_t_bn_affine(const float *a, const float *b, int n, int c, long hw, const float *A, const float *B, const float *C,
                    float *out, void *stream);
int cae_t_colsum(const void *g16, long pixels, int cp, float *out, void *stream); /* bias gradient */

/* Training-mode density of the entropy bottleneck, fused (csrc/cae_density_train.hip).  Replaces the element-wise graph of
 * compressai's EntropyBottleneck.forward(training=True) / _likelihood / LowerBound under autograd
 * (_autoencoders.py:502 via models/tasks/_taskutils.py:95-108).
 * raw_params: (channels, NP) fp32 = per channel [matrices 0..K | biases 0..K | factors 0..K-1], each in its stored
 * (un-transformed) form and row-major shape; NP = cae_t_density_params(D, K) (0: shape not built; built: D = 3, K = 4).
 * forward: out = y + noise (noise may be NULL), lik = max(p(out), bound); plain = 1: sigmoid(u) - sigmoid(l), 0: sign trick.
 * backward: g_y = g_out (may be NULL) + g_lik dp/dy with the LowerBound rule; g_raw_params (channels, NP), overwritten. */
int cae_t_density_params(int filters_d, int n_filters);
int cae_t_density_forward(const float *y, const float *noise, const float *raw_params, int n, int channels, int hw, int plain,
                          float bound, float *out, float *lik, void *stream);
int cae_t_density_backward(const float *out, const float *g_lik, const float *g_out, const float *raw_params, int n,
                           int channels, int hw, int plain, float bound, float *g_y, float *g_raw_params, void *stream);
/* compressai NonNegativeParametrizer (GDN beta / gamma; layers/gdn.py via _autoencoders.py GDN units) under autograd:
 * out = max(x, bound)^2 - pedestal;  g_x = g 2 max(x, bound) where x >= bound or that value is negative (LowerBound rule). */
int cae_t_reparam_forward(const float *x, long n, float bound, float pedestal, float *out, void *stream);
int cae_t_reparam_backward(const float *x, const float *g, long n, float bound, float *gx, void *stream);

/* One clip + Adam step over all parameters of a training step (train_cae_ms.py:221-230: per optimiser
 * clip_grad_norm_(params, max_norm) then torch.optim.Adam.step(), no amsgrad) in two launches.  A group = one optimiser:
 * tensors ordered by group; per group lr, betas, eps, weight_decay, max_norm (<= 0: no clipping) and the 1-based step
 * count of THIS step (bias corrections).  All pointers are device pointers to fp32; partial_ws holds one float per
 * 2048-element chunk (sum over tensors of ceil(numel / 2048)); deterministic (no atomics). */
#define CAE_OPTIM_MAX_TENSORS 64
#define CAE_OPTIM_MAX_GROUPS 8
int cae_t_clip_adam(int ntensors, float *const *params, const float *const *grads, float *const *exp_avg,
                    float *const *exp_avg_sq, const int *numel, const int *group, int ngroups, const float *lr,
                    const float *beta1, const float *beta2, const float *eps, const float *weight_decay,
                    const float *max_norm, const int *step, float *partial_ws, size_t partial_elems, void *stream);

/* ---- host entropy coding ---------------------------------------------------------------
 * Replace compressai._CXX.pmf_to_quantized_cdf and compressai.ans.RansEncoder /
 * RansDecoder (encode_with_indexes / decode_with_indexes), reached from
 * EntropyBottleneck.update / compress / decompress (_autoencoders.py:502, :549-551, :568-571).
 * Streams are coded independently (one per tile), in parallel on `threads` host threads
 * (0 = $CAE_CODER_THREADS, else min(CPUs of this process, 16)).  Symbol order inside a stream is (c, y, x) raster; the CDF row
 * of a symbol is its channel c. */
/* CPUs this process may keep busy: affinity mask, capped by the cgroup CPU quota, divided by LOCAL_WORLD_SIZE. */
int cae_cpu_budget(void);
/* Streams one coder thread walks in lockstep (2 or 4 independent chains per loop iteration; CAE_CODER_LOCKSTEP, else 4
 * when fewer than 16 CPUs are available to this process: fewer CPU-seconds per symbol, half as many work items). */
int cae_coder_lockstep(void);
/* Size of the coder pool a cae_rans_*_batch call with `threads` = requested and n_streams streams uses. */
int cae_coder_threads(int requested, int n_streams);

int cae_pmf_to_quantized_cdf(const float *pmf_host, int n, int precision, uint32_t *cdf_host /* n+1 */);

/* symbols_host: (n_streams, channels, hw) int32.  On success out_bufs[i] (library-allocated,
 * cae_free each) holds stream i and out_lens[i] its byte length. */
int cae_rans_encode_batch(cae_model_t *m, const int32_t *symbols_host, int n_streams, int hw,
                          uint8_t **out_bufs, size_t *out_lens, int threads);
/* The same streams in ONE library-allocated buffer (cae_free): stream i = out_buf[offsets[i] .. offsets[i+1]),
 * offsets has n_streams + 1 entries.  (The batched drivers hand the streams on without per-stream copies.) */
int cae_rans_encode_packed(cae_model_t *m, const int32_t *symbols_host, int n_streams, int hw, uint8_t **out_buf,
                           size_t *offsets, int threads);
/* bufs[i]/lens[i]: stream i.  symbols_host out: (n_streams, channels, hw) int32. */
int cae_rans_decode_batch(cae_model_t *m, const uint8_t *const *bufs, const size_t *lens, int n_streams,
                          int hw, int32_t *symbols_host, int threads);

/* ---- codec front door -------------------------------------------------------------------
 * The reference's numcodecs plugin contract as C entry points: ConvolutionalAutoencoder.encode
 * (_autoencoders.py:539-555: (h,w,c) uint8 chunk -> '>QQ' header + rANS payload) and .decode (:557-584: chunk bytes ->
 * (H,W,c) uint8 tile), one chunk per call, called CONCURRENTLY from dask's thread pool on one shared codec
 * (compress.py:121-128, decompress.py:51-58).  Both calls block and are thread-safe without any caller-side lock.  The
 * range coder of a chunk runs in the calling thread; the GPU parts of the calls that are waiting at the same time are
 * launched as ONE batch (chunks of equal shape, at most max_batch) as soon as one of `inflight` device-side slots is
 * free -- no timer: a lone caller is served at once.  Results do not depend on the grouping.
 *
 * analysis / synthesis: the two track handles of the codec (cae_model_set_layer ... and cae_model_set_entropy done on
 * BOTH: the encode side codes with the analysis handle's tables, the decode side with the synthesis handle's); the
 * door does not own them, they must outlive it and must not be re-configured while calls are in flight.  The handles
 * stay usable through the other entry points on any stream (calls on a handle are ordered on the device in call
 * order).  max_batch <= 0: 32; inflight <= 0: 3.  HIP device = the calling thread's current device.
 * cae_door_destroy serves the queued calls first; no call may be running or started once it has been entered. */
typedef struct cae_door cae_door_t;
int cae_door_create(cae_model_t *analysis, cae_model_t *synthesis, int max_batch, int inflight, cae_door_t **out);
void cae_door_destroy(cae_door_t *door);
/* Codec.encode: tile_host (h,w,c) uint8 C-contiguous (any host memory).  *out: library-allocated chunk (cae_free) =
 * 16-byte big-endian (h,w) header + rANS payload, *out_len its length. */
int cae_door_encode(cae_door_t *door, const uint8_t *tile_host, int h, int w, int c, uint8_t **out, size_t *out_len);
/* Shape of the tile a chunk decodes to: H = (h >> L) << L as the reference derives the latent size (floor, :565). */
int cae_door_decode_shape(cae_door_t *door, const uint8_t *chunk_host, size_t len, int *h, int *w, int *c);
/* Codec.decode: chunk bytes -> out_host (H,W,c) uint8 (capacity in bytes >= H*W*c). */
int cae_door_decode(cae_door_t *door, const uint8_t *chunk_host, size_t len, uint8_t *out_host, size_t out_capacity);
/* Counters since creation / the last reset: batches launched, chunks served, batches repeated on the fp32 kernels,
 * then seconds summed over the calls: caller staging (copy into pinned memory / range decode), caller waiting, caller
 * coding (range encode / copy out), and the waiting time split into queue, launch, device, pull (DMA to the host) and
 * wake-up; last: the calls waiting in the queue right now. */
enum cae_door_stat {
    CAE_DOOR_STAT_BATCHES = 0, CAE_DOOR_STAT_CHUNKS, CAE_DOOR_STAT_FP32_REPEATS, CAE_DOOR_STAT_T_STAGE,
    CAE_DOOR_STAT_T_WAIT, CAE_DOOR_STAT_T_CODE, CAE_DOOR_STAT_T_QUEUE, CAE_DOOR_STAT_T_LAUNCH, CAE_DOOR_STAT_T_DEVICE,
    CAE_DOOR_STAT_T_PULL, CAE_DOOR_STAT_T_WAKE, CAE_DOOR_STAT_QUEUED, CAE_DOOR_STATS
};
int cae_door_stats(cae_door_t *door, double *stats /* n */, int n, int reset);
/* on != 0: the dispatcher starts no further batch (calls queue up; CAE_DOOR_STAT_QUEUED = the number waiting); 0: it
 * resumes.  Lets a caller -- and the tests -- form batches deterministically. */
int cae_door_hold(cae_door_t *door, int on);

#ifdef __cplusplus
}
#endif
#endif /* CAE_HIP_H */

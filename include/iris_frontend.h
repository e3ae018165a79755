/*
 * iris_frontend.h -- C ABI of the MI355X (gfx950) audio feature frontend.
 *
 * Drop-in boundary for the reference's feature hot path.  The reference
 * (IRIS-AUDIO/challenge) has no FFI of its own -- its boundary is a set of
 * Python callables.  This header is what a binding for
 * those callables binds to; every entry point cites the reference interface it
 * replaces (file:line in the upstream tree).  INTEGRATION.md shows the ctypes
 * stub a maintainer would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch / framework types.
 *  - Every tensor argument is a DEVICE pointer owned by the caller (e.g. the
 *    PyTorch caching allocator).  Entry points never allocate, never free and
 *    never synchronise: all work is enqueued on the caller's `stream`
 *    (a hipStream_t passed as void*; NULL = the default stream).  They are
 *    therefore safe to capture into a hipGraph.
 *  - A plan owns its device tables (window, twiddles, mel bands) and a small
 *    workspace.  A plan is bound to one device; calls on one plan must be
 *    issued from one thread at a time (the workspace is shared); distinct plans
 *    are independent.
 *  - Return value: 0 = ok; negative = IRIS_E_* (bad argument / unsupported
 *    shape); positive = hipError_t passed through.  iris_last_error() returns a
 *    thread-local description of the last failure on this thread.
 *  - All arithmetic is IEEE fp32.  Layouts are row-major ("C order").
 *
 * Tensor layouts (reference conventions)
 *    wav   [B, C, L]        fp32 waveform, 16 kHz or any rate
 *    spec  [B, F, T, 2C]    complex STFT, re block then im block on the last
 *                           axis: [..., :C] = re, [..., C:] = im
 *                           (pipeline.py:131-133, data_utils.py:26-27)
 *    mel   [B, M, T, C]     (transforms.py:58-68)
 *    F = n_fft/2 + 1,  T = 1 + L / hop  (torch.stft center=True)
 */
#ifndef IRIS_FRONTEND_H
#define IRIS_FRONTEND_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRIS_ABI_VERSION 1

enum {
    IRIS_OK = 0,
    IRIS_E_INVALID = -1,     /* null pointer, non-positive size, inconsistent arguments */
    IRIS_E_UNSUPPORTED = -2, /* shape outside what the kernels are built for */
    IRIS_E_CAPACITY = -3,    /* batch / length larger than the plan was created for */
    IRIS_E_NOMEM = -4,
    IRIS_E_EPILOGUE_TIMEOUT = -5 /* an EARLIER fused-epilogue launch of this plan gave up a bounded wait (see iris_plan_set_epilogue) */
};

/* flags for iris_wav_to_logmel */
enum {
    IRIS_F_MINMAX = 1,    /* apply per-sample min-max (data_utils.py:37-47) */
    IRIS_F_LOG = 2,       /* apply ln(x + 1e-8)       (data_utils.py:50-55) */
    IRIS_F_NORMALIZE = 4  /* apply x / (10 rms) per clip first (data_utils.py:32-34) */
};

typedef struct iris_plan iris_plan;

int iris_abi_version(void);
const char* iris_last_error(void);

/*
 * Host-side helper.  W[F, M] of tf.signal.linear_to_mel_weight_matrix as the
 * reference builds it once per closure (transforms.py:55-56): HTK mel scale,
 * triangles linear in mel, DC bin zeroed, all arithmetic fp32.
 * `out` is HOST memory, n_bins * n_mel floats, row-major [n_bins][n_mel].
 */
int iris_mel_weight_matrix(int n_mel, int n_bins, float sample_rate,
                           float lower_hz, float upper_hz, float* out_host);

/*
 * Plan: the state the closure magphase_to_mel(num_mel_bins,
 * num_spectrogram_bins, sample_rate, **kw) (transforms.py:51-56) and
 * torchaudio.transforms.Spectrogram(n_fft, power=None) (data_utils.py:17) hold.
 *
 *   n_fft       power of two in [256, 2048]; window = periodic Hann(n_fft).
 *               0 = mel-only plan: any n_bins >= 2, only iris_magmel is available
 *               (magphase_to_mel on spectra of an arbitrary bin count)
 *   hop         > 0 (Spectrogram default: n_fft / 2)
 *   n_mel       M >= 1
 *   n_bins      must equal n_fft/2 + 1 (unless n_fft == 0)
 *   channels    C >= 1 audio channels per clip
 *   max_batch, max_len   capacity of the workspace (clips, samples per channel)
 *   mel_host    optional HOST [n_bins*n_mel] matrix to use instead of the
 *               built-in recipe (any non-negative matrix; band structure is
 *               detected); NULL = iris_mel_weight_matrix(...)
 */
int iris_plan_create(iris_plan** out, int device, int n_fft, int hop, int n_mel,
                     int n_bins, float sample_rate, float lower_hz, float upper_hz,
                     int channels, int max_batch, int max_len, const float* mel_host);
int iris_plan_destroy(iris_plan* plan);

/*
 * Precision of the mel contraction of iris_wav_to_logmel (transforms.py:65, tf.tensordot over the bin axis):
 *   IRIS_MEL_F32       (default) banded fp32 reduction on the vector units.  Stated accuracy of the mel magnitudes against
 *                      an fp64 evaluation of the same chain, for every shape, input level and seed:
 *                      |mel - ref| <= 1e-5 |ref| + 4 eps_fp32 xrms[b, t, c] sum_k W[k, m]  (1e-5 relative + the noise floor of
 *                      any fp32 transform: ~1.2 u x the rms of the frame's spectrum on every bin; measured <= 0.45 of this bound
 *                      over 50 seeds x 5 shapes, profiles/r4/hip_vs_fp64_sweep.log; torch.stft's fp32 engine: 0.82);
 *   IRIS_MEL_F16_MFMA  |X| and W in fp16, v_mfma_f32_16x16x32_f16 with fp32 accumulation (BASELINE configs[4]);
 *                      2e-3 relative.  Needs n_fft 512/1024/2048, n_mel <= 128, every band inside the lower
 *                      half of the spectrum and <= 256 bins per group of 16 bands, else IRIS_E_UNSUPPORTED
 *                      (the plan keeps its previous setting).  Calls with SpecAugment bands always run fp32.
 *                      Dynamic range: 2|X| is cast to fp16 (max 65504, 11 significant bits, subnormal below 6e-5), so
 *                      the waveform must be normalised - as load_wav does (data_utils.py:32-34) - or IRIS_F_NORMALIZE
 *                      set, which scales the samples before the transform; un-normalised PCM-range floats without
 *                      that flag overflow to inf.  The fp32 kernel has no such limit.
 */
#define IRIS_MEL_F32 0
#define IRIS_MEL_F16_MFMA 1
int iris_plan_set_mel_precision(iris_plan* plan, int precision);

/*
 * How iris_wav_to_logmel applies min-max / log (data_utils.py:37-55):
 *   IRIS_EPILOGUE_FUSED        (default) inside the fused kernel: ONE launch per call.  A clip's workgroups exchange
 *                              their (min, max) through the plan's device memory while the kernel runs, which needs
 *                              every workgroup of the launch resident (guaranteed: grid <= CUs) and a per-launch epoch
 *                              from the host - so calls made while `stream` is being captured into a hipGraph, the
 *                              fp16-MFMA variant and shapes whose chunk does not fit the LDS tile take the two-kernel
 *                              path by themselves.  Do not run several fused launches CONCURRENTLY on one device
 *                              (plans on different streams): each could hold CUs while waiting for workgroups the
 *                              other keeps from starting (so could a CU mask, or another process on the device).
 *                              The wait is bounded (~2 s), never a hang, and its failure is LOUD: the affected clips come
 *                              out as NaN, the kernel raises the plan's status word - which lives in host-visible memory -
 *                              and the next iris_wav_to_logmel on the plan (it reads the word on entry: no
 *                              synchronisation, no device access) enqueues nothing, switches the plan to TWO_KERNELS for
 *                              good and returns IRIS_E_EPILOGUE_TIMEOUT: outputs of this plan since the last successful
 *                              iris_plan_status / since that call's predecessor are suspect.  Use TWO_KERNELS from the
 *                              start for pipelines that overlap plans.
 *   IRIS_EPILOGUE_TWO_KERNELS  fused kernel (raw mel + per-wave partials), then the min-max / log kernel.
 *   IRIS_EPILOGUE_IN_PLACE     one launch like FUSED, without the LDS tile: the raw mel goes to `out` and the workgroup
 *                              that wrote a chunk finishes its rows in place once the clip's range is known (same
 *                              exchange, same co-residency rules, same bits), for chunks of any size.  Measured 3-5 %
 *                              SLOWER than TWO_KERNELS where the tile does not fit (c2 geometry from batch 128 on: the same
 *                              bytes move through the same fabric, all CUs at once) - an A/B form, never chosen by itself.
 * A plan created while ROC_GLOBAL_CU_MASK or HSA_CU_MASK is set starts on TWO_KERNELS (a CU mask breaks the co-residency).
 * IRIS_EPILOGUE=1 in the environment at plan creation selects TWO_KERNELS (test / A-B hook; so do IRIS_CHUNK_FRAMES=n,
 * frames per chunk of the fused kernel, and IRIS_MAGMEL_GENERIC - test hooks read once per plan, never per launch).
 * iris_plan_status: synchronises with the device, then 0 = every bounded wait so far completed, 1 = one gave up (the
 * word is reset and the plan stays on TWO_KERNELS from then on).
 * iris_plan_set_epilogue_timeout: the bound of those waits in microseconds (default 2,000,000; 0 = give up after the
 * first sweep - the test hook that makes the failure path reachable on a healthy device).
 */
#define IRIS_EPILOGUE_FUSED 0
#define IRIS_EPILOGUE_TWO_KERNELS 1
#define IRIS_EPILOGUE_IN_PLACE 2
int iris_plan_set_epilogue(iris_plan* plan, int mode);
/* form the plan's last iris_wav_to_logmel call took: IRIS_EPILOGUE_* (a FUSED plan reports TWO_KERNELS where it fell back by
 * itself - a chunk tile beyond the LDS, a capture in progress, an earlier epilogue timeout; IN_PLACE is only ever taken when it
 * was asked for), -1 before the first call or when neither min-max nor log was asked for */
int iris_plan_last_epilogue(const iris_plan* plan, int* form_out);
int iris_plan_status(iris_plan* plan, int* status_out);
int iris_plan_set_epilogue_timeout(iris_plan* plan, unsigned long long microseconds);

/* Name of the fused kernel iris_wav_to_logmel launches for this plan (with / without SpecAugment bands), as rocprofv3
 * prints it without the argument list, e.g. "k_wav_to_mel<10,0,false,false,1,true>" (last flag: min-max / log epilogue inside the kernel); HOST buffer. */
int iris_plan_kernel_name(const iris_plan* plan, int with_bands, char* out_host, int capacity);
/* Copy the plan's mel matrix [n_bins*n_mel] to HOST memory. */
int iris_plan_get_mel(const iris_plan* plan, float* out_host);
/* Frames for a clip of `len` samples: 1 + len / hop. */
int iris_plan_num_frames(const iris_plan* plan, int len);

/*
 * resample (data_utils.py:20-21: torchaudio.compliance.kaldi.resample_waveform(wav, r, 16000), the step of load_wav in front of
 * normalize + STFT).  torchaudio (unpinned in requirements.txt:5, not vendored) implements it as functional.resample with
 * lowpass_filter_width 6, rolloff 0.99 and a Hann-windowed sinc: a polyphase FIR of 2 ceil(6 o / (0.99 min(o, n))) + o taps per
 * output phase on the reduced ratio o : n; csrc/k_resample.h restates it (taps in fp64, rounded once).
 * wav: DEVICE [channels][len] fp32 at orig_freq; out: DEVICE [channels][iris_resample_len(len, orig_freq, new_freq)] fp32.
 * orig_freq == new_freq is a copy.  Runs on the current HIP device; the tap table of a rate pair is cached per device.
 */
long long iris_resample_len(long long len, int orig_freq, int new_freq);
int iris_resample(const float* wav, int channels, long long len, int orig_freq, int new_freq, float* out, void* stream);

/*
 * normalize (data_utils.py:32-34):  out[r] = wav[r] / (10 * sqrt(mean(wav[r]^2)))
 * per row r of wav[n_rows, row_len]; one row = all C*L samples of one clip (the
 * reference takes the rms over every channel jointly).  out may alias wav.
 * workspace: DEVICE scratch of iris_normalize_workspace(n_rows, row_len) floats.
 * Runs on the current HIP device.
 */
size_t iris_normalize_workspace(int n_rows, size_t row_len);
int iris_normalize(const float* wav, float* out, int n_rows, size_t row_len, float* workspace,
                   size_t workspace_floats, void* stream);

/*
 * STFT of load_wav (data_utils.py:17-27): reflect-pad n_fft/2, frame at `hop`,
 * periodic Hann, one-sided FFT, no normalisation, written in the reference's
 * [B, F, T, 2C] re-block / im-block layout.
 * flags: 0, or IRIS_F_NORMALIZE = `normalize` of load_wav (data_utils.py:22-23, :32-34: wav / (10 rms), rms over all
 *        channels of a clip jointly) folded into the transform: one partial-sums launch, then the spectrum is scaled by
 *        1 / (10 rms) as it is written (the STFT is linear) - the normalised waveform is never materialised.  Differs
 *        from normalising first by the fp32 rounding of the scaled samples (<= 2e-6 of the spectrum's peak).
 */
int iris_stft(iris_plan* plan, const float* wav, float* spec, int batch, int len, int flags,
              void* stream);

/*
 * complex_to_magphase (transforms.py:111-123) on n_outer rows of 2C floats:
 * out[..., :C] = sqrt(re^2 + im^2), out[..., C:] = atan2(im, re).
 * magphase_to_complex (transforms.py:126-134) is the inverse.
 */
int iris_complex_to_magphase(const float* in, float* out, size_t n_outer, int channels,
                             void* stream);
int iris_magphase_to_complex(const float* in, float* out, size_t n_outer, int channels,
                             void* stream);

/*
 * complex_to_magphase + magphase_to_mel fused (transforms.py:111-123, :58-70,
 * call sites sj_train.py:119-120): mel[b,m,t,c] = sum_f W[f,m] * |spec[b,f,t,c]|.
 * `is_magphase` != 0: `spec` already holds magnitudes in [..., :C] (the phase
 * half is ignored, transforms.py:64), i.e. plain magphase_to_mel.
 * t_bands / f_bands: optional DEVICE int32 [B, n, 2] (offset, size) bands zeroed
 * along time / linear frequency before the reduction -- SpecAugment `mask`
 * (transforms.py:12-40) as `augment` applies it (data_utils.py:58-61) and
 * stft_filter (data_utils.py:126-136).  NULL / 0 = none.
 */
int iris_magmel(iris_plan* plan, const float* spec, float* mel, int batch, int n_frames,
                int is_magphase, const int32_t* t_bands, int n_t_bands,
                const int32_t* f_bands, int n_f_bands, void* stream);

/*
 * minmax + log_on_mel (data_utils.py:37-55; fused twin trainer.py:63-77),
 * in place on x[n_rows, row_len]: per row (x - min) / max(max - min, eps_div)
 * when do_minmax, then ln(x + eps_log) when do_log.  A batched [B,M,T,C] tensor
 * is n_rows = B; the unbatched [M,T,C] call of metrics.py:53 is n_rows = M
 * (the reference reduces "every axis except 0").
 * workspace: DEVICE scratch of iris_minmax_log_workspace(n_rows, row_len)
 * floats (may be NULL when !do_minmax).  Runs on the current HIP device.
 */
size_t iris_minmax_log_workspace(int n_rows, size_t row_len);
int iris_minmax_log(float* x, int n_rows, size_t row_len, int do_minmax, int do_log, float eps_div,
                    float eps_log, float* workspace, size_t workspace_floats, void* stream);

/*
 * The fused hot path: [normalize ->] STFT -> magnitude -> band masks -> mel
 * [-> min-max] [-> log] without materialising the spectrum
 * (load_wav data_utils.py:22-23 + sj_train.py:108-123).  flags = IRIS_F_*.
 * One launch on `stream` (min-max / log inside the fused kernel, IRIS_EPILOGUE_FUSED) or two (the fused kernel, then
 * min-max / log over its per-wave partials) - see iris_plan_set_epilogue.  Partials and exchange slots live in the
 * plan's own workspace, so calls on ONE plan must be ordered on one stream (or otherwise serialised) - create one
 * plan per stream to run batches concurrently.  Safe under hipGraph capture and replay: a call made during capture
 * takes the two-kernel path, which keeps no per-call host state and has no atomics to reset.
 */
int iris_wav_to_logmel(iris_plan* plan, const float* wav, float* out, int batch, int len,
                       int flags, const int32_t* t_bands, int n_t_bands,
                       const int32_t* f_bands, int n_f_bands, void* stream);

/*
 * mask apply (transforms.py:12-40, deterministic part): x viewed as
 * [n_outer, axis_len, n_inner]; elements whose axis index falls in any band
 * [offset_i, offset_i + size_i) are set to zero (= multiply by the product of
 * the 0/1 masks, in x's dtype).  bands: DEVICE int32 [n_groups, n_bands, 2];
 * outer index o uses group o / outer_per_group (one group per sample).
 * elem_size 4 or 8 bytes (float32/int32 or float64/int64).  In place.
 */
int iris_mask_apply(void* x, size_t n_outer, size_t axis_len, size_t n_inner, int elem_size,
                    const int32_t* bands, int n_bands, size_t outer_per_group, void* stream);

/*
 * adaptive_clip_grad (sj_train.py:145-155 with unitwise_norm, utils.py:350-366) followed by the
 * optimiser's element-wise clipvalue (sj_train.py:435), in place on the gradients, one launch for
 * the whole model.  rows: DEVICE array, one record per output unit (a row of a Linear / LSTM
 * weight, an output channel of a conv kernel, or a whole 1-D tensor): for each row
 *     max_norm = max(||param||, eps) * clip_factor
 *     grad    *= max_norm / max(||grad||, 1e-6)      where ||grad|| >= max_norm
 *     grad     = clamp(grad, -clipvalue, +clipvalue)  when clipvalue > 0
 * Runs on the current HIP device.
 */
typedef struct {
    const float* param; /* first element of the row, contiguous */
    float* grad;
    int64_t len;
} iris_agc_row;
int iris_agc_clip(const iris_agc_row* rows_dev, size_t n_rows, float clip_factor, float eps,
                  float clipvalue, void* stream);

/*
 * The same launch carrying on into the optimiser (round 6): AGC + clipvalue + the Adam update of sj_train.py:434-435 / :176-182 in
 * one pass over parameters, gradients and both moments (torch.optim.Adam's arithmetic, no weight decay / amsgrad; the clipped
 * gradient is written back to `grad`).  use_agc == 0: clipvalue + Adam only.  lr_dev: nullable DEVICE float (a capturable
 * optimiser's learning-rate tensor), else lr_host; step_dev: DEVICE float holding the step count t >= 1 of THIS update (the
 * optimiser's already incremented counter).  Rows as for iris_agc_clip, plus the two moment rows in the parameter's own layout.
 */
typedef struct {
    float* param;
    float* grad;
    int64_t len;
    float* exp_avg;
    float* exp_avg_sq;
} iris_agc_adam_row;
int iris_agc_clip_adam(const iris_agc_adam_row* rows_dev, size_t n_rows, float clip_factor, float eps_agc, float clipvalue,
                       int use_agc, const float* lr_dev, float lr_host, double beta1, double beta2, float eps,
                       const float* step_dev, void* stream);

/*
 * Inference epilogue of ConvMPBlock's Conv2D + BatchNormalization + ReLU (+ MaxPool2D 2x2 'same'), sj_train.py:191-201,
 * once the eval-mode BatchNorm is folded into the convolution (sj_train.fold_batchnorm): the convolution itself stays
 * MIOpen (PyTorch-ROCm, per north_star); these replace the separate bias-add, ReLU and pooling passes over its output.
 *   iris_bias_relu:          x[o, c] = max(x[o, c] + bias[c], 0) in place on a channels-last tensor [n_outer, channels]
 *   iris_bias_relu_maxpool:  y[b, i, j, c] = max over the 2x2 window (stride 2, clipped at the edges = Keras 'same' /
 *                            ceil mode) of max(x[b, 2i+di, 2j+dj, c] + bias[c], 0);  x [B, H, W, C] -> y [B, ceil(H/2), ceil(W/2), C]
 * channels must be a multiple of 4; x, y, bias DEVICE, 16-byte aligned.  Run on the current HIP device.
 */
int iris_bias_relu(float* x, const float* bias, size_t n_outer, int channels, void* stream);
/* the same on a CONTIGUOUS (NCHW) activation x [batch, channels, inner] (inner = H * W, a multiple of 4), and the pooling
 * variant reading NCHW x [B, C, H, W] and writing the pooled tensor channels-last y [B, ceil(H/2), ceil(W/2), C]: block 1 of
 * the CRNN runs its convolutions in NCHW (MIOpen is faster there for 32 -> 32 channels at 64 x 512) and hands over in NHWC */
int iris_bias_relu_nchw(float* x, const float* bias, size_t batch, int channels, size_t inner, void* stream);
int iris_bias_relu_maxpool_nchw(const float* x, const float* bias, float* y, int batch, int height, int width, int channels,
                                void* stream);
int iris_bias_relu_maxpool(const float* x, const float* bias, float* y, int batch, int height, int width, int channels,
                           void* stream);

/*
 * Training-mode Conv2D bias + BatchNormalization + ReLU of ConvMPBlock (sj_train.py:191-201; Keras BatchNormalization
 * momentum 0.99 / epsilon 1e-3 = torch momentum 0.01) on the channels-last convolution output z [rows = N H W, channels]
 * in two passes each way instead of MIOpen's / ATen's seven forward and nine backward (bias add, mean / variance,
 * normalise, ReLU; ReLU', dscale / dbias, dx, bias gradient).  The convolution stays MIOpen and is run WITHOUT bias:
 * batch normalisation removes the batch mean, so the output does not depend on the bias - it only shifts the running
 * mean (conv_bias, nullable, is added there) - and the bias gradient is identically zero.
 *   forward : iris_bn_stats (sums_zeroed: DEVICE double [iris_bn_sums_len(channels)], zero on entry: per-channel sum and sum
 *             of squares of z - K, K = the channel's value in row 0 of z - shifted sums, so that a channel whose |mean| is
 *             far above its spread keeps its variance; spread over several copies so that the blocks' atomics do not queue
 *             on one address; the consumers below add the copies up and read K back from z)
 *             iris_bn_relu_apply: y = max(gamma (z - mean) rstd + beta, 0), biased variance; running_mean / running_var
 *             updated in place ((1 - momentum) old + momentum new, unbiased variance); save_mean / save_rstd [channels] out
 *   backward: iris_bn_relu_bwd_reduce (sums_zeroed, same length: sum g, sum g xhat with g = dy [y > 0]; the mask is
 *             recomputed from z with the forward's own expression, y is not read)
 *             iris_bn_relu_bwd_dx: dz = gamma rstd (g - sum_g / M - xhat sum_gx / M); dgamma = sum g xhat, dbeta = sum g
 * channels: a multiple of 4, <= 4096; every pointer DEVICE, 16-byte aligned tensors.  Run on the current HIP device.
 * NOT in-place: y (and p below) must not overlap z - every block of the apply passes reads K back from row 0 of z while
 * others write their outputs (IRIS_E_INVALID otherwise).
 */
/* Round 6: the forward statistics can come out of the CONVOLUTION's epilogue instead (iris_conv3x3_wino_bn, iris_conv3x3_wino_b3_bn,
 * iris_conv3x3_c32_bn: the bare convolution + sum z / sum z^2 per output channel into a zeroed [iris_bn_sums_len] buffer, accumulated
 * from the registers z is stored from) - the iris_bn_stats pass over z disappears; the *_sums0 forms of the apply entry points
 * consume sums taken about zero (iris_bn_stats' are taken about row 0 of z).  Replaces the same layers of sj_train.py:191-201. */
size_t iris_bn_sums_len(int channels);
int iris_bn_relu_apply_sums0(const float* z, float* y, size_t rows, int channels, const double* sums, const float* gamma,
                             const float* beta, const float* conv_bias, float eps, float momentum, float* running_mean,
                             float* running_var, float* save_mean, float* save_rstd, void* stream);
int iris_bn_relu_pool_apply_sums0(const float* z, float* p, int batch, int height, int width, int channels, const double* sums,
                                  const float* gamma, const float* beta, const float* conv_bias, float eps, float momentum,
                                  float* running_mean, float* running_var, float* save_mean, float* save_rstd, void* stream);
int iris_conv3x3_wino_bn(const float* x, const float* packed, float* y, int batch, int height, int width, int cin, int cout, int flags,
                         double* bn_sums_zeroed, void* stream);
int iris_conv3x3_wino_b3_bn(const float* x, const float* packed, float* y, int batch, int height, int width, int cin, int cout, int flags,
                            double* bn_sums_zeroed, void* stream);
int iris_conv3x3_c32_bn(const float* x, const float* weight, long stride_o, long stride_i, long stride_h, long stride_w, float* y,
                        int batch, int height, int width, double* bn_sums_zeroed, void* stream);

/*
 * All the weight packings of a training step in ONE launch (round 6; replaces N calls of iris_wino_pack_weights_device /
 * iris_wino_b3_pack_weights_device - 23 per step of the CRNN, 0.16 ms of launches and tails).  jobs: HOST array (read during the
 * call, passed to the kernel by value: nothing device-side to keep alive, capturable into a hipGraph); per job the arguments of the
 * single-layer entry points; first_block is filled in by the call.  split_bf16 != 0: the iris_wino_b3 packing (cin % 16 == 0).
 * Any number of jobs (48 per launch).
 */
typedef struct iris_pack_job {
    const float* weight;
    float* packed;
    long stride_o, stride_i, stride_h, stride_w;
    int cin, cout, transposed, first_block;
} iris_pack_job;
int iris_wino_pack_weights_device_multi(iris_pack_job* jobs_host, int n_jobs, int split_bf16, void* stream);
int iris_bn_stats(const float* z, size_t rows, int channels, double* sums_zeroed, void* stream);
int iris_bn_relu_apply(const float* z, float* y, size_t rows, int channels, const double* sums, const float* gamma,
                       const float* beta, const float* conv_bias, float eps, float momentum, float* running_mean,
                       float* running_var, float* save_mean, float* save_rstd, void* stream);
int iris_bn_relu_bwd_reduce(const float* z, const float* dy, size_t rows, int channels, const float* save_mean,
                            const float* save_rstd, const float* gamma, const float* beta, double* sums_zeroed, void* stream);
int iris_bn_relu_bwd_dx(const float* z, const float* dy, float* dz, size_t rows, int channels, const float* save_mean,
                        const float* save_rstd, const float* gamma, const float* beta, const double* sums, float* dgamma,
                        float* dbeta, void* stream);

/*
 * The same passes for the LAST convolution of a ConvMPBlock, with its MaxPool2D(2, 2, 'same') folded in (sj_train.py:199-200:
 * pooling follows the block's last Conv-BN-ReLU): z [batch, height, width, channels] channels-last; p and dp
 * [batch, ceil(height / 2), ceil(width / 2), channels].  The full-size y is never written and the full-size dy never exists:
 *   forward : iris_bn_stats as above (the statistics are those of the full-size z), then
 *             iris_bn_relu_pool_apply: p = max over the window of max(gamma (z - mean) rstd + beta, 0)
 *   backward: iris_bn_relu_pool_bwd_reduce / iris_bn_relu_pool_bwd_dx: g = dp at the window's first maximum in (h, w) scan
 *             order (max_pool2d's index rule) where that maximum is > 0, 0 elsewhere; sums, dz, dgamma, dbeta as above.
 * Same constraints as above.
 */
int iris_bn_relu_pool_apply(const float* z, float* p, int batch, int height, int width, int channels, const double* sums,
                            const float* gamma, const float* beta, const float* conv_bias, float eps, float momentum,
                            float* running_mean, float* running_var, float* save_mean, float* save_rstd, void* stream);
int iris_bn_relu_pool_bwd_reduce(const float* z, const float* dp, int batch, int height, int width, int channels,
                                 const float* save_mean, const float* save_rstd, const float* gamma, const float* beta,
                                 double* sums_zeroed, void* stream);
int iris_bn_relu_pool_bwd_dx(const float* z, const float* dp, float* dz, int batch, int height, int width, int channels,
                             const float* save_mean, const float* save_rstd, const float* gamma, const float* beta,
                             const double* sums, float* dgamma, float* dbeta, void* stream);

/*
 * Second layer of the CRNN's first block for inference: Conv2D(32 -> 32, 3x3 'same') + (BatchNorm-folded) bias + ReLU, and
 * with pool != 0 the block's MaxPool2D(2, 2, 'same') behind it (sj_train.py:191-201, 244), as an implicit GEMM on the fp32
 * matrix cores (exact fp32).  x [batch, height, width, 32] channels-last, weight [32, 32, 3, 3] contiguous, bias [32];
 * y [batch, height, width, 32] or, pooled, [batch, ceil(height / 2), ceil(width / 2), 32]; with out_chunked != 0 the same
 * values in the channel-chunked layout [batch][4][Ho][Wo][8] that iris_conv3x3_wino_bias_relu reads.  x 16-byte aligned.
 */
int iris_conv3x3_c32_bias_relu(const float* x, const float* weight, const float* bias, float* y, int batch, int height, int width,
                               int pool, int out_chunked, void* stream);
/* The same kernel as the bare convolution of a TRAINING pass (BatchNorm follows: no bias, no ReLU, no pooling): y = conv3x3_same(x, w)
 * with transposed == 0 (forward) or y = conv3x3_same(x, w') with w'[ci][co][ky][kx] = w[co][ci][2 - ky][2 - kx] (transposed != 0:
 * the backward-data pass, x = dz).  x, y channels-last [batch, height, width, 32]; weight [32, 32, 3, 3] with element strides
 * stride_o / _i / _h / _w (a channels_last parameter as it is). */
int iris_conv3x3_c32(const float* x, const float* weight, long stride_o, long stride_i, long stride_h, long stride_w, int transposed,
                     float* y, int batch, int height, int width, void* stream);

/*
 * Conv2D(cin -> cout, 3x3 'same', stride 1) as a Winograd F(2x2, 3x3) transform on the fp32 matrix cores: 16 instead of 36
 * multiplies per 2 x 2 output tile and channel pair - MIOpen's implicit GEMMs already run these layers at the fp32 MFMA rate.
 * Blocks 2-5 of the CRNN (sj_train.py:191-201, 222-242): for inference with the (BatchNorm-folded) bias, ReLU and the block's
 * MaxPool2D(2, 2, 'same') fused; for training as the bare convolution of the deep layers' forward and backward-data passes.
 * fp32 throughout; against an fp64 convolution the error is that of a direct fp32 convolution (<= 1e-6 of the output's peak).
 * cin % 8 == 0, cout % 64 == 0.
 *   x      channel-chunked activation [batch][cin / 8][height][width][8] (8 channels of a pixel contiguous) - or, with
 *          IRIS_WINO_IN_NHWC, channels-last [batch][height][width][cin] (14 % slower: every gather touches 4x the cache lines);
 *          DEVICE, 16-byte aligned, fewer than 2^30 elements
 *   packed U = G g G^T in the kernel's LDS order, 16 cin cout floats (iris_wino_packed_len), DEVICE, 16-byte aligned:
 *          iris_wino_pack_weights (HOST buffers in and out, weight [cout][cin][3][3] contiguous; upload the result once per layer) or
 *          iris_wino_pack_weights_device (DEVICE, on `stream`, weight with arbitrary element strides - a channels_last parameter as
 *          it is -; transposed != 0 packs the weights of the backward-data pass, dx = conv(dz, W') with W'[ci][co][i][j] =
 *          W[co][ci][2 - i][2 - j]: `cin` then counts the original OUTPUT channels)
 *   bias   [cout] or NULL
 *   y      [batch][cout / 8][Ho][Wo][8] - the next layer's x - or, with IRIS_WINO_OUT_NHWC, channels-last [batch][Ho][Wo][cout];
 *          Ho x Wo = height x width, or ceil(height / 2) x ceil(width / 2) with IRIS_WINO_POOL
 *   flags  IRIS_WINO_POOL | IRIS_WINO_OUT_NHWC | IRIS_WINO_IN_NHWC | IRIS_WINO_RELU
 */
enum { IRIS_WINO_POOL = 1, IRIS_WINO_OUT_NHWC = 2, IRIS_WINO_IN_NHWC = 4, IRIS_WINO_RELU = 8 };
size_t iris_wino_packed_len(int cin, int cout);
int iris_wino_pack_weights(const float* weight_host, int cin, int cout, float* packed_host);
int iris_wino_pack_weights_device(const float* weight, long stride_o, long stride_i, long stride_h, long stride_w, int cin, int cout,
                                  int transposed, float* packed, void* stream);
int iris_conv3x3_wino(const float* x, const float* packed, const float* bias, float* y, int batch, int height, int width, int cin,
                      int cout, int flags, void* stream);

/*
 * The same convolution with its 16 GEMMs on the BF16 matrix cores AT FP32 ACCURACY (round 6; opt-in: IRIS_WINO_SPLIT_BF16=1 on the
 * Python side): v_mfma_f32_32x32x16_bf16 delivers 16x the FLOP per clock of the fp32 MFMA that bounds iris_conv3x3_wino.  Every fp32
 * operand is the EXACT sum of three bf16 values (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)); U is split once per
 * layer when it is packed, V in registers; six of the nine partial products (all down to 2^-24 of the product) are accumulated in
 * fp32 by the matrix cores.  Against an fp64 convolution the error is 0.65 - 1.14x that of iris_conv3x3_wino on the CRNN's
 * geometries (profiles/r6/wino_b3_check_and_time.log; tests/test_transforms_gpu.py holds it to 1.5x and to 1e-6 of the peak), the twelve
 * layers of the forward run 1.13 - 1.57x faster.  Same contract as iris_conv3x3_wino (layouts, flags, bias, pooling) except:
 *   cin % 16 == 0 (one MFMA K-step), cout % 64 == 0
 *   packed  iris_wino_b3_pack_weights_device only: 24 cin cout floats = 96 cin cout bytes (iris_wino_b3_packed_len, in floats)
 * Replaces: the same Conv2D layers of define_keras_model (sj_train.py:191-201, 222-242).
 */
size_t iris_wino_b3_packed_len(int cin, int cout);
int iris_wino_b3_pack_weights_device(const float* weight, long stride_o, long stride_i, long stride_h, long stride_w, int cin, int cout,
                                     int transposed, float* packed, void* stream);
int iris_conv3x3_wino_b3(const float* x, const float* packed, const float* bias, float* y, int batch, int height, int width, int cin,
                         int cout, int flags, void* stream);

/*
 * The WEIGHT GRADIENT of the same convolution, y = conv3x3_same(x, w), as a Winograd F(2x2, 3x3) transform on the fp32 matrix
 * cores - model.fit's backward pass through blocks 2-5 (sj_train.py:191-201, 222-242, 408), which MIOpen runs as implicit
 * GEMMs at 82-127 TFLOP/s: dU[p] = sum over the tiles of the batch of (A dY A^T)[p] (B^T d B)[p] per position p, dW = G^T dU G -
 * 16 instead of 36 multiplies per 2 x 2 tile and channel pair.  Both free dimensions of that GEMM are channels, so with
 * channels-last activations a lane loads its channel's patch straight from memory into the MFMA operand layout (no LDS).
 * Deterministic (the tile rows are split over workgroups whose partial sums are added in a fixed order); against an fp64
 * gradient the error is that of a direct fp32 one (2e-7 .. 6e-7 of the gradient's peak).  cin % 32 == 0, cout % 32 == 0.
 *   x, dy      channels-last [batch][height][width][cin] / [...][cout], DEVICE, fewer than 2^31 bytes each
 *   dw         the gradient, element strides stride_o / _i / _h / _w over [cout][cin][3][3] (a channels_last parameter's
 *              gradient is written as it lies); accumulate != 0 adds to what is there
 *   workspace  DEVICE scratch of at least iris_wino_wrw_workspace_len(batch, height, width, cin, cout) floats (64 MiB on 256 CUs)
 */
size_t iris_wino_wrw_workspace_len(int batch, int height, int width, int cin, int cout);
int iris_conv3x3_wino_wrw(const float* x, const float* dy, float* dw, long stride_o, long stride_i, long stride_h, long stride_w,
                          int batch, int height, int width, int cin, int cout, int accumulate, float* workspace,
                          size_t workspace_len, void* stream);

/*
 * The CRNN's first layer in TRAINING mode - Conv2D(3x3 'same', 1 or 2 input channels) + BatchNormalization + ReLU
 * (sj_train.py:191-201, 244) - with the convolution recomputed from x wherever its output is needed (9-18 FMAs per value
 * against 4 bytes of traffic): z is never stored.  x [batch, in_channels, height, width] contiguous; weight
 * [out_channels, in_channels, 3, 3] contiguous; y / dy [batch, height, width, out_channels] (channels-last); out_channels
 * in {4, 8, ..., 256} dividing 1024.  The convolution runs WITHOUT bias (see iris_bn_stats); x gets no gradient.
 *   forward : iris_conv0_stats (sums_zeroed as for iris_bn_stats), then iris_conv0_bn_relu (arguments as iris_bn_relu_apply)
 *   backward: iris_conv0_bn_relu_backward: two launches (sum g / sum g xhat, then dz and the weight gradient);
 *             sums_zeroed [iris_bn_sums_len(out_channels)] and dweight_zeroed [iris_conv0_dweight_len(...)] DEVICE doubles,
 *             zero on entry; dweight = several partial copies [copies][out_channels][in_channels][3][3] that the caller
 *             adds up (the blocks' atomics are spread over them); dgamma, dbeta [out_channels] floats.
 */
size_t iris_conv0_dweight_len(int in_channels, int out_channels);
int iris_conv0_stats(const float* x, const float* weight, int batch, int in_channels, int out_channels, int height, int width,
                     double* sums_zeroed, void* stream);
int iris_conv0_bn_relu(const float* x, const float* weight, float* y, int batch, int in_channels, int out_channels, int height,
                       int width, const double* sums, const float* gamma, const float* beta, const float* conv_bias, float eps,
                       float momentum, float* running_mean, float* running_var, float* save_mean, float* save_rstd, void* stream);
int iris_conv0_bn_relu_backward(const float* x, const float* weight, const float* dy, int batch, int in_channels,
                                int out_channels, int height, int width, const float* save_mean, const float* save_rstd,
                                const float* gamma, const float* beta, double* sums_zeroed, double* dweight_zeroed, float* dgamma,
                                float* dbeta, void* stream);

/*
 * First convolution of the CRNN (ConvMPBlock's Conv2D(32, 3, padding='same') on the n_chan-channel log-mel input,
 * sj_train.py:191-201, 244) with its (BatchNorm-folded) bias and ReLU in ONE pass, NCHW: x [batch, in_channels (1 or 2),
 * height, width], weight [out_channels, in_channels, 3, 3], bias [out_channels] -> y [batch, out_channels, height, width].
 * The layer writes 16-32x what it reads; every output byte is written once.  width a multiple of 4; x, y 16-byte aligned.
 */
int iris_conv3x3_small_bias_relu_nchw(const float* x, const float* weight, const float* bias, float* y, int batch,
                                      int in_channels, int out_channels, int height, int width, void* stream);
/* the same with a channels-last output y [batch, height, width, out_channels] (out_channels in {4, 8, ..., 256} dividing
 * 1024; width <= 2048; bias, y 16-byte aligned), which is what iris_conv3x3_c32_bias_relu reads */
int iris_conv3x3_small_bias_relu_nhwc(const float* x, const float* weight, const float* bias, float* y, int batch,
                                      int in_channels, int out_channels, int height, int width, void* stream);

/*
 * Recurrent half of the v9 CRNN's Bidirectional(LSTM(128, return_sequences=True)) (sj_train.py:252), one launch for all
 * time steps and both directions.  gx [batch, steps, 2, 512] = the input pre-activations x_t W_ih^T + b_ih + b_hh of
 * direction 0 (forward) and 1 (backward), gate rows in the order i, f, g, o (one GEMM for all steps, done by the caller);
 * w_hh [2, 512, 128] the recurrent matrices (16-byte aligned); out [batch, steps, 256] = (h forward, h backward) per step;
 * h_0 = c_0 = 0.  act: NULL for inference; for training [batch, steps, 2, 5, 128] receives (i, f, g, o, c) of every step.
 * iris_bilstm128_backward: dout [batch, steps, 256] and the saved act -> dgx [batch, steps, 2, 512], the gradient of the
 * pre-activations (back-propagation through time inside the launch); dW_ih, db, dx follow from dgx through the caller's GEMM,
 * dW_hh[d] = dgx[:, :, d, :]^T . h_prev with h_prev = out shifted by one step of direction d.
 * fp32, device pointers, current HIP device.
 */
int iris_bilstm128_forward(const float* gx, const float* w_hh, float* out, float* act, int batch, int steps, void* stream);
int iris_bilstm128_backward(const float* dout, const float* act, const float* w_hh, float* dgx, int batch, int steps,
                            void* stream);

/*
 * Sample synthesis in the complex-STFT domain, deterministic half of
 * merge_complex_specs (pipeline.py:6-110) for a whole batch: every output sample
 * is   background crop (tiled along time, pipeline.py:29-35)
 *    + sum over its voices of gain * crop(zero-padded voice) * no_overlap (:48-84)
 *    + sum over its noises of gain * crop(zero-padded noise)              (:86-106)
 * with the frame labels of the accepted voices (:55-66, :78-84).  All random draws
 * (which sources, offsets, gains) are made by the caller and passed in the source
 * table; the data-dependent parts - a voice frame is "active" when max over
 * (freq, chan2) of the frame is > 0 (:57), a voice is dropped when accepting it
 * would make two labels overlap (:78-80) - are evaluated here.
 * Additions happen in table order with separately rounded multiply and add, so
 * the result equals the reference's op-by-op evaluation bit for bit.
 *
 * srcs: DEVICE table, the sources of sample b are srcs[first[b] .. first[b+1]) in
 *       the order background, voices (by slot), noises.  first: DEVICE int32 [B + 1].
 *   src    DEVICE [F, T, C2] fp32 spectrogram (time axis 1, re block | im block last)
 *   active DEVICE [T] fp32 0/1 flags of a voice's frames from iris_mix_frame_active (kind 1;
 *          they depend on the source only, so they are computed once per corpus, not per use)
 *   T      frames in src;  pad: zero frames virtually added on both sides (>= 0);
 *   off    crop offset in the padded (voice, noise) or tiled (background) source;
 *   gain   linear gain (ignored for the background);  kind 0 background, 1 voice, 2 noise, -1 unused slot (skipped);
 *   slot   row of the labels output (kind 1);  label_row: row of label_vecs (kind 1).
 * label_vecs: DEVICE [n_label_rows, n_classes] fp32 (one-hot rows in the reference).
 * spec_out: DEVICE [B, F, n_frame, C2];  labels_out: DEVICE [B, V, n_frame, n_classes].
 * workspace: DEVICE scratch of iris_mix_workspace(n_srcs, n_frame) floats.
 * Runs on the current HIP device.
 */
typedef struct {
    const float* src;
    const float* active;
    int32_t T, pad, off;
    float gain;
    int32_t kind, slot, label_row, reserved;
} iris_mix_src;
/* active_out[t] = 1 when max over (freq, chan2) of src[:, t, :] is > 0, else 0 (pipeline.py:57) */
int iris_mix_frame_active(const float* src, int n_bins, int n_frames, int chan2, float* active_out,
                          void* stream);
size_t iris_mix_workspace(int n_srcs, int n_frame);
int iris_mix_specs(const iris_mix_src* srcs_dev, int n_srcs, const int32_t* first_dev,
                   const float* label_vecs_dev, float* spec_out, float* labels_out, int batch,
                   int n_bins, int n_frame, int chan2, int max_voices, int n_classes,
                   float* workspace, size_t workspace_floats, void* stream);

/*
 * Waveform-domain variant of the batched merge_complex_specs (the STFT is linear, so
 * mixing before it equals mixing after it up to frame-granular cropping - output frames whose window straddles a crop,
 * pad or tiling boundary differ, all others agree to fp32 rounding).  The corpus stays resident as waveforms (a
 * quarter of the bytes of its spectrograms at n_fft 1024 / hop 256) and the mixed batch goes straight into
 * iris_wav_to_logmel; no spectrum is ever materialised.
 * Same source table and the same label rule as iris_mix_specs, with these readings of the record:
 *   src      DEVICE [C, len] fp32 waveform;  reserved = len (samples per channel)
 *   T, pad, off   in FRAMES, as for spectrograms (T = 1 + len / hop); a frame is `hop` samples:
 *            voice / noise sample s of the output reads source sample s + (off - pad) * hop (zero outside [0, len)),
 *            background sample s reads (off * hop + s) mod len
 *   active   [T] flags from iris_mix_wave_frame_active: frame t is active when any sample under the support of its
 *            periodic-Hann window, [t*hop - n_fft/2 + 1, t*hop + n_fft/2 - 1] clipped to the clip, is non-zero in any
 *            channel.  Equivalent to the reference's rule (max over the frame's kept half-spectrum, re and im parts,
 *            > 0: pipeline.py:57) except for degenerate frames: a non-zero windowed frame whose half-spectrum has no
 *            strictly positive re / im component (e.g. x and the window such that every Re X[k], Im X[k], k <= n_fft/2,
 *            is <= 0) is inactive there and active here; tests/test_oracle.py compares the two rules on the synthetic
 *            corpus (they agree on every frame)
 * wav_out: DEVICE [B, C, out_len] with out_len = (n_frame - 1) * hop, i.e. n_frame STFT frames (center=True).
 */
int iris_mix_wave_frame_active(const float* wav, int channels, int len, int n_fft, int hop, float* active_out,
                               void* stream);
int iris_mix_waves(const iris_mix_src* srcs_dev, int n_srcs, const int32_t* first_dev,
                   const float* label_vecs_dev, float* wav_out, float* labels_out, int batch,
                   int channels, int hop, int n_frame, int max_voices, int n_classes,
                   float* workspace, size_t workspace_floats, void* stream);

/*
 * The random half of a batch drawn ON THE DEVICE (no host round trip, replayable from a hipGraph): which sources
 * merge_complex_specs mixes and with which offsets / gains (pipeline.py:29-106; source picking of the dataset graph,
 * :147-174), written as the source table iris_mix_specs / iris_mix_waves consume, and the SpecAugment bands of
 * `augment` (data_utils.py:58-61 -> transforms.py:25-26).  Generator: Philox4x32-10 keyed by `seed`, counter = (call
 * counter, sample, column, purpose); distributions as in the reference:
 *   background: the next source of a shuffled, repeated stream; crop offset ~ U{0 .. reps*T - n_frame}
 *   voices:     the next max_voices sources of their stream, padded to the longest of the group;
 *               n_voices ~ U{1 .. max_voices-1} (1 when max_voices == 1); gain = 10^-U[0, -snr/10);
 *               offset ~ U{0 .. padded_len - n_frame - 1} (0 when that range is empty)
 *   noises:     n_noises ~ U{0 .. max_noises-1}; gain = 10^-U[0, 2); offset ~ U{0 .. max(padded_len - n_frame, 0)}
 *   bands:      size ~ U{0 .. max_mask-1}, offset ~ U{0 .. axis_len - size - 1}
 * "Shuffled, repeated stream" = one keyed pseudo-random permutation of the corpus per epoch (every source exactly once
 * per epoch).  A corpus is described by DEVICE arrays of `n` entries: source addresses, (voices) addresses of their
 * frame-activity flags, frames per source, and - waveform corpora - samples per channel (NULL for spectrogram corpora).
 * state_dev: DEVICE uint64[4] {call counter, background / voice / noise stream positions} (iris_augment_draw: uint64[1]),
 * zero-initialised by the caller once and advanced by every call ON THE DEVICE.
 * table_out: DEVICE [batch * (1 + max_voices + max_noises)] records, FIXED stride per sample (first_out[b] = b * stride,
 * DEVICE int32 [batch + 1]); unused voice / noise slots carry kind = -1, which the mix kernels skip.  Pass
 * n_srcs = batch * stride to iris_mix_specs / iris_mix_waves.  t_bands_out / f_bands_out: DEVICE int32 [batch, n, 2].
 */
typedef struct {
    const float* const* src;    /* DEVICE [n] */
    const float* const* active; /* DEVICE [n] (voices) or NULL */
    const int32_t* T;           /* DEVICE [n] frames per source */
    const int32_t* len;         /* DEVICE [n] samples per channel (waveform corpora) or NULL */
    int32_t n;
} iris_mix_corpus;
int iris_mix_draw(const iris_mix_corpus* backgrounds, const iris_mix_corpus* voices, const iris_mix_corpus* noises /* nullable */,
                  int batch, int n_frame, int max_voices, int max_noises, float min_ratio, float min_noise_ratio, float snr,
                  uint64_t seed, uint64_t* state_dev, iris_mix_src* table_out, int32_t* first_out, void* stream);
int iris_augment_draw(int batch, int n_time, int n_time_masks, int max_time_mask, int n_freq, int n_freq_masks,
                      int max_freq_mask, uint64_t seed, uint64_t* state_dev, int32_t* t_bands_out, int32_t* f_bands_out,
                      void* stream);

/*
 * Per-kernel timing for bench.py: with enable = n > 0 every n-th call of
 * iris_wav_to_logmel carries a start/stop hipEvent pair around each of its
 * kernels on the launch stream (n = 1: every call; an event pair costs a few
 * microseconds of stream time, so time throughput with timing off and the
 * kernels in a separate pass); 0 switches it off.  The first 4 calls after
 * enabling are never sampled (first dispatch on an idle GPU, clock ramp).
 * iris_timing_samples synchronises the events and copies the durations (ms) of
 * kernel 0 (the fused kernel) or 1 (the min-max / log kernel, when the call
 * ran it) recorded since the last enable / read to HOST memory; it does not
 * reset.  iris_timing_read returns count and mean of kernel 0 and resets both.
 */
int iris_timing_enable(iris_plan* plan, int enable);
int iris_timing_read(iris_plan* plan, int* n_launches, float* mean_ms);
int iris_timing_samples(iris_plan* plan, int kernel, float* out_ms_host, int capacity, int* n_samples);

#ifdef __cplusplus
}
#endif
#endif /* IRIS_FRONTEND_H */

/* mia.h — C ABI of libmia_hip.so, the MI355X (gfx950) spectral front-end.
 *
 * Drop-in boundary for the reference's optional native extension
 * `mlx_audio.primitives._ext` (nanobind module, python/csrc/bindings.cpp:13-598)
 * and for the third-party array ops the reference's Python path calls
 * (mx.fft.rfft / irfft / matmul).  Citations below are relative to the
 * reference checkout.
 *
 * Conventions
 *   - plain C types only; device buffers are raw pointers into HIP device
 *     memory owned by the caller (PyTorch-ROCm tensors on the Python side);
 *   - all device work is enqueued on `stream` (a hipStream_t passed as void*);
 *     no function synchronises, allocates or frees device memory except
 *     mia_plan_create / mia_plan_destroy;
 *   - inputs and outputs are contiguous float32 / interleaved complex64 unless
 *     strides are passed;
 *   - every function returns MIA_OK or a negative MIA_ERR_*; the message for
 *     the calling thread is available from mia_last_error().
 */
#ifndef MIA_H_
#define MIA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIA_OK 0
#define MIA_ERR_INVALID (-1)     /* bad argument (maps to ValueError)          */
#define MIA_ERR_UNSUPPORTED (-2) /* e.g. n_fft without a compiled FFT plan     */
#define MIA_ERR_HIP (-3)         /* HIP runtime failure (maps to RuntimeError) */

#define MIA_PAD_CONSTANT 0
#define MIA_PAD_EDGE 1
#define MIA_PAD_REFLECT 2

#define MIA_EPI_NONE 0        /* mel power as is                                */
#define MIA_EPI_WHISPER 1     /* log10(max(.,1e-10)), max-8 clamp, (.+4)/4      */
#define MIA_EPI_DB 2          /* coef*log10(max(.,amin)/max(ref,amin)), top_db  */

typedef struct mia_plan mia_plan;

const char* mia_last_error(void);
int mia_version(void);

/* ---- host-side table builders (float64 arithmetic, float32 results) -------- */

/* Window taps.  Replaces _ext.generate_window (bindings.cpp:107-132,
 * csrc/primitives/windows.cpp:179-228); follows the NumPy fallback
 * primitives/windows.py:19-122,166-186 (no symmetrisation step).  `type` is one
 * of hann|hanning|hamming|blackman|bartlett|triangular|rectangular|boxcar|ones.
 * periodic != 0: length+1 symmetric points without the last. */
int mia_generate_window(const char* type, int length, int periodic, float* out);

/* Replace _ext.hz_to_mel / _ext.mel_to_hz (bindings.cpp:135-183); formulas of
 * primitives/mel.py:42-104 (Slaney or HTK), float64 in and out. */
int mia_hz_to_mel(const double* hz, int n, int htk, double* out);
int mia_mel_to_hz(const double* mels, int n, int htk, double* out);

/* Replaces _ext.mel_filterbank (bindings.cpp:185-223); follows the builder the
 * reference actually calls, primitives/mel.py:111-179: eps-regularised slopes,
 * cast to float32, then slaney scale.  norm: 0 = None, 1 = "slaney".
 * out: n_mels x (n_fft/2+1) row-major. */
int mia_mel_filterbank(int sr, int n_fft, int n_mels, double fmin, double fmax,
                       int htk, int norm, float* out);

/* Replaces _ext.get_dct_matrix (bindings.cpp:315-340); float64 basis cast to
 * float32 as primitives/mfcc.py:23-66.  out: n_out x n_in row-major. */
int mia_dct_matrix(int n_out, int n_in, int ortho, float* out);

/* ---- plans ------------------------------------------------------------------ */

typedef struct mia_plan_desc {
  int n_fft;
  int hop_length;
  int center;            /* 0/1 */
  int pad_mode;          /* MIA_PAD_*  */
  const float* window;   /* host, n_fft taps (already centre-padded)            */
  int n_mels;            /* 0: no mel stage                                     */
  const float* mel_basis;/* host, n_mels x (n_fft/2+1)                          */
  float power;           /* exponent applied to |X| before the projection       */
  int n_mfcc;            /* 0: no DCT stage                                     */
  const float* dct_basis;/* host, n_mfcc x n_mels                               */
  const float* lifter;   /* host, n_mfcc multipliers or NULL                    */
} mia_plan_desc;

/* Uploads window, twiddles, the band form of the filterbank and the DCT basis
 * to the current device.  Stands in for the reference's table caches
 * (primitives/stft.py:27-106, mel.py:107-120,239-253, mfcc.py:19-66,
 * csrc/primitives/dct.cpp:19-42). */
int mia_plan_create(const mia_plan_desc* desc, mia_plan** out);
int mia_plan_destroy(mia_plan* plan);
/* Ablation and test knobs of a plan, and facts about it; neither has a counterpart in the
 * reference.  Options: "w400_table_sweep" (1: the n_fft = 400 kernel walks its filterbank
 * from tables even when the bank is one of the two compiled-in Whisper banks; results are
 * bit-identical either way), "w400_vgpr_staging" (1: that kernel stages its samples through
 * registers instead of LDS-DMA loads and takes the Cooley-Tukey split of the transform;
 * results agree to float32 round-off), "w400_mfma_mel" (1: the register-staging kernel
 * projects onto the mel bank with v_mfma_f32_16x16x4_f32 over 16-filter bands instead of
 * sweeping; slower, kept for A/B measurements; needs n_mels <= 128), "w400_dpp_pass2" (1: the
 * LDS-DMA kernel runs the radix-16 pass of its transform across the lanes with DPP operands
 * instead of exchanging through LDS; slower, kept for A/B measurements), "fwd_lane" (1: frame-
 * major mel rows at n_fft 1024 from the kernel with four frames per wave), "wave_fft" (0: an
 * n_fft served by the one-wave-per-frame transform back on the whole-row engine),
 * "wave_istft_fused" (0: a framing whose inverse runs on the one-wave-per-frame transform's
 * fused kernel back on inverse transform + frame workspace + overlap-add).  Info keys: "w400_bank"
 * (80 / 128 when the compiled sweep applies, else 0), "w400_tables" (1 when the n_fft = 400
 * kernel serves the plan), "w400_mfma_tables" (1 when the matrix-core mode is available),
 * "compute_units", "wave_fixed_plan" (1 when a compiled-in plan of the one-wave-per-frame transform
 * serves the plan's n_fft, 0 for its run-time form, -1 for any other plan), "istft_fused" (1 when
 * mia_istft runs in one launch without workspace).  Not thread-safe against launches that use the plan. */
int mia_plan_set_option(mia_plan* plan, const char* key, int value);
int mia_plan_get_info(const mia_plan* plan, const char* key);
/* 1 + (T + 2*pad - n_fft) / hop, or a negative error if the signal is too short
 * (primitives/_frame_impl.py:116-122). */
int64_t mia_plan_num_frames(const mia_plan* plan, int64_t T);
int mia_fft_size_supported(int n_fft);

/* ---- the hot path ------------------------------------------------------------ */

/* Fused pad + frame + window + real FFT.  Replaces the compiled core
 * primitives/stft.py:118-130 (incl. _ext.pad_signal / _ext.frame_signal and
 * mx.fft.rfft).  y: (B, T) float32.  out: (B, F, n_fft/2+1) complex64 — the
 * physical layout of the reference's result before its lazy transpose
 * (stft.py:213-216).  Every value is multiplied by `scale`. */
int mia_stft(const mia_plan* plan, const float* y, int64_t B, int64_t T,
             float scale, void* out, void* stream);
/* The same for every plan.  An n_fft without a compiled kernel (mia_fft_size_supported() == 0)
 * is served by an any-length engine - mx.fft.rfft takes any length, primitives/stft.py:130 -
 * that materialises the frames in `workspace` (mia_stft_workspace_bytes() bytes; 0 for the
 * compiled sizes, where this is mia_stft) and transforms them by mixed-radix passes (2-3-5-7
 * smooth lengths) or a chirp-z convolution (all others, odd and prime lengths included).
 * mia_istft, mia_melspectrogram and mia_mfcc serve such plans through the workspaces they
 * already take; mia_stft, mia_irfft_frames, mia_spectral_stats and mia_melspectrogram_static
 * reject them. */
size_t mia_stft_workspace_bytes(const mia_plan* plan, int64_t B, int64_t T);
int mia_stft_ws(const mia_plan* plan, const float* y, int64_t B, int64_t T, float scale,
                void* out, void* workspace, size_t workspace_bytes, void* stream);

/* Inverse real FFT of every frame.  Replaces mx.fft.irfft at
 * primitives/stft.py:295.  spec is addressed as
 * spec[b*stride_b + k*stride_bin + f*stride_frame] (complex elements), so both
 * the (B, bins, F) contiguous layout and the transposed view that mia_stft
 * produces are read in place.  frames: (B, F, n_fft) float32. */
int mia_irfft_frames(const mia_plan* plan, const void* spec, int64_t B, int64_t F,
                     int64_t stride_b, int64_t stride_bin, int64_t stride_frame,
                     float* frames, void* stream);

/* Gather-formulated overlap-add with window-sum normalisation.  Replaces
 * _ext.overlap_add (bindings.cpp:17-48, csrc/primitives/overlap_add.cpp:195-233)
 * and the inline kernel primitives/stft.py:548-596:
 *   out[b,i] = sum_f w[i-f*hop] * frames[b,f,i-f*hop] / max(sum_f w[i-f*hop]^2, 1e-8)
 * frames: (B, F, n_fft); window: device, n_fft; out: (B, out_len).  Every
 * value is multiplied by `scale` after the division. */
int mia_overlap_add(const float* frames, const float* window, int64_t B, int64_t F,
                    int n_fft, int hop, int64_t out_len, float scale, float* out,
                    void* stream);

/* Inverse STFT = mia_irfft_frames + mia_overlap_add (primitives/stft.py:292-312).
 * `workspace` must hold mia_istft_workspace_bytes() bytes of device memory.  For
 * hop = n_fft / 4 with n_fft in {1024, 2048, 4096} both steps run in one launch that
 * keeps the overlapping frames in LDS (same result bit for bit) and the workspace
 * size is 0; so they do (same overlap-add arithmetic, transforms equal to float32 round-off)
 * for n_fft in {256, 320 .. 2048 with a compiled-in one-wave-per-frame plan} at any hop of at
 * least n_fft / 8.  Always size the workspace with mia_istft_workspace_bytes(). */
size_t mia_istft_workspace_bytes(const mia_plan* plan, int64_t B, int64_t F);
int mia_istft(const mia_plan* plan, const void* spec, int64_t B, int64_t F,
              int64_t stride_b, int64_t stride_bin, int64_t stride_frame,
              int64_t out_len, float scale, float* out, void* workspace,
              size_t workspace_bytes, void* stream);

/* Fused STFT -> |X|^power -> mel projection -> optional log epilogue.
 * Replaces primitives/mel.py:345-373 (+ :265-276) and, by epilogue,
 * models/whisper/inference.py:111-113 or primitives/convert.py:42-57.
 * out: (B, n_mels, F) float32, frames contiguous.
 *   MIA_EPI_NONE   : mel power.
 *   MIA_EPI_WHISPER: (max(log10(max(mel,1e-10)), gmax-8)+4)/4.
 *   MIA_EPI_DB     : max(coef*log10(max(mel,amin)/max(ref,amin)), gmax-top_db);
 *                    top_db < 0 means "no clamp".
 * gmax is the maximum over each clip (per_clip != 0) or over the whole batch
 * (per_clip == 0, what the reference computes for the array it is given).
 * `workspace` must hold mia_mel_workspace_bytes() bytes. */
size_t mia_mel_workspace_bytes(const mia_plan* plan, int64_t B, int64_t T);
int mia_melspectrogram(const mia_plan* plan, const float* y, int64_t B, int64_t T,
                       int epilogue, float ref, float amin, float coef, float top_db,
                       int per_clip, float* out, void* workspace,
                       size_t workspace_bytes, void* stream);

/* The same with the result written frame-major, out: (B, F, n_mels) float32 - the layout the
 * reference's CLAP features have ([T, F] after its transpose, models/clap/
 * feature_extractor.py:205-212): a tile of a few frames then leaves the kernel as one contiguous
 * run instead of n_mels short row pieces.  Served by the compiled generic kernels when
 * n_mels % 4 == 0 (mia_mel_frames_supported(plan) == 1); other plans return
 * MIA_ERR_UNSUPPORTED and the caller transposes. */
int mia_mel_frames_supported(const mia_plan* plan);
int mia_melspectrogram_frames(const mia_plan* plan, const float* y, int64_t B, int64_t T,
                              int epilogue, float ref, float amin, float coef, float top_db,
                              int per_clip, float* out, void* workspace,
                              size_t workspace_bytes, void* stream);

/* mia_melspectrogram for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc): the same kernels and the
 * same result bit for bit, but the clamp keys live in a device array of the library itself
 * (512 KiB per GPU, held until the process ends) instead of a caller workspace.  It has no
 * counterpart in the reference.  At most MIA_STATIC_MAX_CLIPS clips per call (more: use
 * mia_melspectrogram); calls that overlap in time on one GPU share the array and race — issue
 * them on one stream or order them yourself.  Calls no hipMalloc / hipFree. */
#define MIA_STATIC_MAX_CLIPS 65535
int mia_melspectrogram_static(const mia_plan* plan, const float* y, int64_t B, int64_t T,
                              int epilogue, float ref, float amin, float coef, float top_db,
                              int per_clip, float* out, void* stream);

/* MFCC from waveform: the chain above with MIA_EPI_DB(ref=1, amin=1e-10,
 * top_db=80) followed by DCT-II along the mel axis and the lifter
 * (primitives/mfcc.py:226-281).  out: (B, n_mfcc, F). */
size_t mia_mfcc_workspace_bytes(const mia_plan* plan, int64_t B, int64_t T);
int mia_mfcc(const mia_plan* plan, const float* y, int64_t B, int64_t T, float top_db,
             int per_clip, float* out, void* workspace, size_t workspace_bytes,
             void* stream);

/* ---- stand-alone ops of the reference extension ------------------------------- */

/* _ext.pad_signal (bindings.cpp:78-104, csrc/primitives/pad_signal.cpp:133-162):
 * (B, T) -> (B, T + 2*pad). */
int mia_pad_signal(const float* x, int64_t B, int64_t T, int64_t pad, int mode,
                   float* out, void* stream);
/* _ext.frame_signal (bindings.cpp:50-76, csrc/primitives/frame_signal.cpp:119-155):
 * (B, T) -> (B, 1 + (T - L)/hop, L). */
int mia_frame_signal(const float* x, int64_t B, int64_t T, int frame_length, int hop,
                     float* out, void* stream);
/* primitives/convert.py:14-57 on an existing array: x viewed as (B, n) —
 * out = coef*log10(max(x,amin)/max(ref,amin)), clamped to (max - top_db) per row
 * (per_clip) or over everything; top_db < 0: no clamp.  workspace: >= 8*(B+1) bytes (row maxima and
 * minima: rows that do not reach below their floor are not touched a second time). */
int mia_to_db(const float* x, int64_t B, int64_t n, float ref, float amin, float coef,
              float top_db, int per_clip, float* out, void* workspace, void* stream);
/* primitives/convert.py:97-126 (db_to_power, divisor 10) and :166-195 (db_to_amplitude,
 * divisor 20): out[i] = ref * 10^(x[i] / divisor), n values, one pass. */
int mia_from_db(const float* x, int64_t n, float ref, float divisor, float* out, void* stream);
/* _ext.dct (bindings.cpp:342-370, csrc/primitives/dct.cpp:103-159) along the
 * middle axis of x viewed as (outer, n_in, inner):
 *   out[o,k,i] = sum_n basis[k,n] * x[o,n,i];  basis: device, n_out x n_in. */
int mia_dct2(const float* x, int64_t outer, int n_in, int64_t inner, const float* basis,
             int n_out, float* out, void* stream);
/* |z| and atan2(im, re) of n complex64 values (primitives/stft.py:347-379). */
int mia_magnitude(const void* z, int64_t n, float* out, void* stream);
int mia_phase(const void* z, int64_t n, float* out, void* stream);
/* The same (op 0: |z|, op 1: atan2) for the spectrum as mia_stft writes it, (rows, frames, bins),
 * delivered in the order the reference's stft / magnitude / phase return, (rows, bins, frames):
 * magnitude(stft(y)) without the transposing copy in between. */
int mia_spectrum_part(const void* z, int64_t rows, int64_t frames, int64_t bins, int op, float* out,
                      void* stream);

/* ---- sample-rate conversion (primitives/resample.py) -------------------------------
 * Padding modes of the polyphase filter (the `padtype` values of resample_poly,
 * primitives/resample.py:213-214, that extend the signal by copying samples). */
#define MIA_RS_PAD_CONSTANT 0
#define MIA_RS_PAD_EDGE 1
#define MIA_RS_PAD_REFLECT 2
#define MIA_RS_PAD_SYMMETRIC 3
#define MIA_RS_PAD_WRAP 4
#define MIA_RS_PAD_SMOOTH 5        /* the end segments continued as straight lines */
#define MIA_RS_PAD_LINE 6          /* the line through the first and the last sample */
#define MIA_RS_PAD_ANTISYMMETRIC 7 /* ... -x1 -x0 | x0 x1 ... */
#define MIA_RS_PAD_ANTIREFLECT 8   /* point reflection about the end samples */
/* resample_poly (primitives/resample.py:209-302, which hands float32 rows to
 * scipy.signal.resample_poly at :275): upsample by `up`, FIR-filter, keep every `down`-th
 * sample.  x: device (rows, n_in).  phase_table: device (up, taps_per_phase), row t = the
 * taps of phase t, oldest input sample first (the zero-padded filter h, h[(P-1-j)*up + t]).
 * Output o is the float32 sum over j, in order, of x[q/up - P + 1 + j] * table[q%up][j]
 * with q = (o + skip) * down; `skip` = outputs of the full convolution dropped in front
 * (the filter's delay).  out: device (rows, n_out).  Bit-identical to SciPy's float32 path. */
int mia_upfirdn(const float* x, int64_t rows, int64_t n_in, const float* phase_table, int up,
                int down, int taps_per_phase, int64_t skip, int64_t n_out, int pad_mode, float* out,
                void* stream);
/* The same filter (primitives/resample.py:275) by the kernel that shares a lane's sample window
 * between adjacent outputs: same arguments and the same bits out, but the phase table's
 * rows carry `margin` zeros in front of and behind their taps_per_phase taps — device
 * (up, margin + taps_per_phase + margin).  mia_upfirdn_margin returns the margin this kernel
 * needs for the ratio, or 0 when it does not apply (spans beyond the LDS, short filters whose
 * lanes would meet in one LDS bank, rows shorter than one run of outputs): then call
 * mia_upfirdn. */
int mia_upfirdn_margin(int up, int down, int taps_per_phase, int64_t n_out);
/* Bytes of LDS one workgroup of mia_upfirdn_shared takes for the ratio (staged samples plus the
 * table of step constants; at most the CU's 160 KiB), or 0 when the kernel does not apply.
 * Touches no device.  (Plan query of this library; the reference has no counterpart.) */
int64_t mia_upfirdn_shared_lds_bytes(int up, int down, int taps_per_phase);
int mia_upfirdn_shared(const float* x, int64_t rows, int64_t n_in, const float* padded_table,
                       int margin, int up, int down, int taps_per_phase, int64_t skip,
                       int64_t n_out, int pad_mode, float* out, void* stream);
/* resample(res_type="linear") (primitives/resample.py:171-196): out[r,i] =
 * float(((1-f)*x[r,lo] + f*x[r,min(lo+1,n_in-1)]) * gain) in float64, at position
 * i*step (the last output sits exactly on n_in-1), lo = floor, f = fraction.
 * step = (n_in-1)/(n_out-1) (0 when n_out == 1); gain = target_sr/orig_sr when the
 * caller asked for scaling, else 1. */
int mia_resample_linear(const float* x, int64_t rows, int64_t n_in, int64_t n_out, double step,
                        double gain, float* out, void* stream);
/* resample(res_type="fft") (primitives/resample.py:84-136, which hands float32 rows to
 * scipy.signal.resample at :120): rfft over the whole row of n_in samples, the spectrum
 * truncated or zero-extended to n_out/2+1 bins (a Nyquist bin shared between the two lengths
 * is doubled when shortening, halved when lengthening), irfft of length n_out, times
 * n_out/n_in and `gain` (target_sr/orig_sr when the caller asked for scaling, else 1).
 * Any n_in, n_out in [1, 2^27]: the transforms are chirp-z convolutions over power-of-two
 * FFTs.  x: device (rows, n_in); out: device (rows, n_out).  `tables` (device,
 * mia_resample_fft_table_bytes) is filled once per (n_in, n_out) by mia_resample_fft_tables,
 * whose workspace needs mia_resample_fft_workspace_bytes(1, ...); `workspace` (device) must
 * hold mia_resample_fft_workspace_bytes(rows, ...).  float32 arithmetic: agrees with SciPy's
 * float32 result to a few 1e-6 of the row's peak (not bit-identical). */
size_t mia_resample_fft_table_bytes(int64_t n_in, int64_t n_out);
size_t mia_resample_fft_workspace_bytes(int64_t rows, int64_t n_in, int64_t n_out);
int mia_resample_fft_tables(int64_t n_in, int64_t n_out, void* tables, void* workspace,
                            size_t workspace_bytes, void* stream);
int mia_resample_fft(const float* x, int64_t rows, int64_t n_in, int64_t n_out, float gain,
                     const void* tables, float* out, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ---- spectral features (primitives/features.py) -------------------------------------
 * Per-frame statistics over the frequency axis of S: device (rows, bins, frames), frames
 * contiguous; freq: device (bins) bin frequencies; out: device (rows, frames).
 *   MIA_STAT_CENTROID   spectral_centroid  (features.py:59-136): sum(f S) / (sum S + 1e-8)
 *   MIA_STAT_BANDWIDTH  spectral_bandwidth (features.py:139-241): aux = centroid (rows, frames);
 *                       (sum(S |f - c|^p) / (sum S + 1e-8))^(1/p), unnormalised if norm == 0
 *   MIA_STAT_ROLLOFF    spectral_rolloff   (features.py:244-382, _ext.spectral_rolloff):
 *                       f of the first bin whose running sum reaches fraction * total
 *   MIA_STAT_FLATNESS   spectral_flatness  (features.py:385-464): geometric / arithmetic mean of
 *                       max(S^power, amin) (power = 1 when S already is the power spectrogram) */
#define MIA_STAT_CENTROID 0
#define MIA_STAT_BANDWIDTH 1
#define MIA_STAT_ROLLOFF 2
#define MIA_STAT_FLATNESS 3
#define MIA_STAT_ALL 4 /* the four at once, two reads of S: out (rows, 4, frames) in the order above */
int mia_spectral_stat(const float* S, int64_t rows, int bins, int64_t frames, const float* freq,
                      const float* aux, int mode, float p, int norm, float fraction, float amin,
                      float power, float* out, void* stream);
/* The four statistics straight from the waveform, one launch and no spectrogram in memory:
 * the forward kernel (pad, frame, window, FFT, magnitude) with the statistics as its epilogue
 * (features.py:26-56 + :59-464 for y given).  plan: any plan (its mel / DCT stages are not
 * used); freq: device (n_fft/2 + 1); out: device (B, 4, frames) = centroid, bandwidth
 * (exponent p_exp, normalised if norm), roll-off (roll_percent), flatness (of
 * max(magnitude^flat_power, flat_amin)). */
int mia_spectral_stats(const mia_plan* plan, const float* y, int64_t B, int64_t T, const float* freq,
                       float p_exp, int norm, float roll_percent, float flat_power, float flat_amin,
                       float* out, void* stream);
/* spectral_contrast (features.py:467-639; _ext.spectral_contrast): S as above; bands: device
 * int32 (n_bands_out, 3) = {first bin, one past the last bin, values averaged at each end} per
 * octave band, built by the caller from the bin frequencies as the reference does
 * (features.py:566-599); max_quantile = the largest third entry (<= 64).  out: device
 * (rows, n_bands_out, frames) = peak - valley (linear) or 10 log10 peak - 10 log10 valley. */
int mia_spectral_contrast(const float* S, int64_t rows, int bins, int64_t frames, const int* bands,
                          int n_bands_out, int max_quantile, int linear, float* out, void* stream);
/* zero_crossing_rate (features.py:642-769): y device (rows, n); frames of frame_length every
 * hop_length samples after centre padding (zeros, or the edge sample if pad_edge) when
 * `center`; out device (rows, 1 + (n + 2*pad - frame_length) / hop_length) = sign changes
 * inside the frame / frame_length; sign = (x >= 0), or the IEEE sign bit if by_signbit
 * (the reference's use_mlx=False variant). */
int mia_zero_crossing_rate(const float* y, int64_t rows, int64_t n, int frame_length, int hop_length,
                           int center, int pad_edge, int by_signbit, float* out, void* stream);

/* ---- onset strength (primitives/onset.py) -------------------------------------------
 * onset_strength's spectral flux (onset.py:125-147): S device (rows, bands, frames), frames
 * contiguous; out device (rows, frames): 0 for the first `lag` frames (the reference pads
 * them back, :170-171), else the mean / max / median over the bands of
 * max(log1p(1e4 S[m,t]) - log1p(1e4 S[m,t-lag]), 0). */
#define MIA_AGG_MEAN 0
#define MIA_AGG_MAX 1
#define MIA_AGG_MEDIAN 2
int mia_onset_flux(const float* S, int64_t rows, int bands, int64_t frames, int lag, int aggregate,
                   float* out, void* stream);
/* The envelope filters of onset_strength (onset.py:150-166, scipy.ndimage.maximum_filter1d /
 * uniform_filter1d with their default reflecting boundary) over rows of n samples (row strides
 * in elements; src and dst must not overlap): mode 0 = maximum over the window
 * [i - size/2, i + (size-1)/2]; mode 1 = max(src[i] - mean over that window, 0). */
int mia_envelope_filter(const float* src, int64_t rows, int64_t n, int64_t src_stride, int size,
                        int mode, float* dst, int64_t dst_stride, void* stream);

/* in: complex64 (rows, bins, frames) with frames contiguous (the layout the reference's stft
 * returns and its istft takes, stft.py:222, 225); out: the same values as (rows, frames, bins),
 * the layout mia_stft writes and mia_istft reads fastest (stride_bin = 1).  For callers who hold
 * a spectrum in the first layout: transposing it first and handing mia_istft whole rows is several
 * times faster than its strided path. */
int mia_transpose_spectrum(const void* in, int64_t rows, int64_t bins, int64_t frames, void* out,
                           void* stream);
/* ---- Griffin-Lim (primitives/griffinlim.py) -----------------------------------------
 * out[i] = mag[i] * exp(i angle[i]) (griffinlim.py:121, 247): n complex64 values. */
int mia_polar(const float* mag, const float* angle, int64_t n, void* out, void* stream);
/* One phase update (griffinlim.py:165-176): with u = c_new / |c_new| (1 where c_new = 0),
 * r = mag * u;  rebuilt = r + momentum * (r - tprev) (r itself if momentum == 0);  tprev = r.
 * c_new, tprev, rebuilt: n complex64 values, mag: n floats, all in the same element order. */
int mia_griffinlim_update(const void* c_new, const float* mag, void* tprev, void* rebuilt,
                          int64_t n, float momentum, void* stream);
/* Set-up of the loop (griffinlim.py:118-123) in one pass: S and angle, device float (rows, bins,
 * frames) with frames contiguous (angle may be NULL: phase zero), become mag_t: float (rows,
 * frames, bins), and rebuilt = tprev = S * exp(i angle): complex64 (rows, frames, bins) - the
 * order mia_istft / mia_stft / mia_griffinlim_update work in. */
int mia_griffinlim_init(const float* S, const float* angle, int64_t rows, int64_t bins, int64_t frames,
                        float* mag_t, void* rebuilt, void* tprev, void* stream);

/* ---- spectral gating (primitives/spectral_gate.py) ------------------------------------
 * The host arithmetic the reference runs in NumPy / scipy.ndimage between its STFT and inverse
 * STFT, on magnitudes (rows, bins, frames), frames contiguous.
 * mia_frame_energy: mean over bins of mag^2 per frame (spectral_gate.py:162).
 * mia_noise_threshold: per (row, bin) mean + n_std * population std over the frames listed in
 *   frame_idx (rows, n_selected) int32, or over all frames if frame_idx is null
 *   (spectral_gate.py:151-154, 168-177).
 * mia_time_min_filter: scipy.ndimage.minimum_filter1d(x, size, axis=-1, mode="nearest") over
 *   rows of n values (spectral_gate.py:183-186, 322-324).
 * mia_box_filter: scipy.ndimage.uniform_filter1d(x, size, mode="nearest") along the middle axis
 *   of x viewed as (outer, n, inner); two calls make the reference's uniform_filter with size
 *   (1, a, b) (spectral_gate.py:207-212, 336).
 * mia_gate_mask: clip((mag - thr * gain) / (mag + 1e-8), 0, 1) (spectral_gate.py:190-196); thr is
 *   (rows, bins, frames), or with per_bin (rows_thr, bins) with rows_thr = rows or 1.
 * mia_mask_op: op 0: 1 - p * (1 - a) (spectral_gate.py:215); op 1: a * b (:218). */
int mia_frame_energy(const float* mag, int64_t rows, int bins, int64_t frames, float* energy,
                     void* stream);
int mia_noise_threshold(const float* mag, int64_t rows, int bins, int64_t frames, const int* frame_idx,
                        int n_selected, float n_std, float* thr, void* stream);
int mia_time_min_filter(const float* x, int64_t rows, int64_t n, int size, float* out, void* stream);
int mia_box_filter(const float* x, int64_t outer, int64_t n, int64_t inner, int size, float* out,
                   void* stream);
int mia_gate_mask(const float* mag, const float* thr, int64_t rows, int bins, int64_t frames,
                  int per_bin, int64_t rows_thr, float gain, float* mask, void* stream);
int mia_mask_op(const float* a, const float* b, int64_t n, int op, float p, float* out, void* stream);
/* mia_gate_mask, optionally mia_mask_op(op 0) and a mia_box_filter over size_freq bins in one
 * pass, bit-identical to the separate calls (size_freq = 1: no averaging). */
int mia_gate_mask_box(const float* mag, const float* thr, int64_t rows, int bins, int64_t frames,
                      int per_bin, int64_t rows_thr, float gain, int keep, float p, int size_freq,
                      float* out, void* stream);
/* The closing steps in one pass (spectral_gate.py:207-221): optionally a mia_box_filter of the
 * mask over size_time frames (1: none), m = keep ? 1 - p (1 - mask) : mask, out = mag * m *
 * exp(i phase); inputs (rows, bins, frames) floats, out complex64 in the (rows, frames, bins)
 * order mia_istft reads fastest. */
int mia_gate_apply(const float* mag, const float* mask, const float* phase, int64_t rows, int64_t bins,
                   int64_t frames, int keep, float p, int size_time, void* out, void* stream);
/* The same steps for a caller that still holds the complex spectrum the magnitudes came from:
 * mag * exp(i phase) is that spectrum, so out = spec * m, spec and out complex64 in
 * (rows, frames, bins) order, mask (rows, bins, frames).  Before the time average the mask may
 * also be averaged over size_freq bins (1: none), bit-identical to a mia_box_filter launch
 * along the bins.  Equal to mia_gate_apply up to the rounding of its sine / cosine; no magnitude
 * or phase planes are read. */
int mia_gate_apply_spectrum(const void* spec, const float* mask, int64_t rows, int64_t bins,
                            int64_t frames, int keep, float p, int size_freq, int size_time,
                            void* out, void* stream);
/* The stationary gate without a magnitude plane: the same three steps taken from the complex
 * spectrum itself, spec complex64 (rows, frames, bins) as mia_stft writes it, |z| = hypotf(re, im)
 * on the fly.  mia_noise_threshold_spec and mia_gate_mask_box_spec (per-bin thresholds, windows
 * of up to 33 bins: MIA_ERR_UNSUPPORTED beyond) are bit-identical to mia_noise_threshold and
 * mia_gate_mask_box on mia_magnitude's plane; mia_frame_energy_spec adds the bins in another
 * order (equal to float32 rounding).  Outputs as those calls': energy (rows, frames), thr (rows,
 * bins), out (rows, bins, frames). */
int mia_frame_energy_spec(const void* spec, int64_t rows, int bins, int64_t frames, float* energy,
                          void* stream);
int mia_noise_threshold_spec(const void* spec, int64_t rows, int bins, int64_t frames, const int* frame_idx,
                             int n_selected, float n_std, float* thr, void* stream);
int mia_gate_mask_box_spec(const void* spec, const float* thr, int64_t rows, int bins, int64_t frames,
                           int64_t rows_thr, float gain, int keep, float p, int size_freq, float* out,
                           void* stream);

/* The stationary estimate's choice of frames (spectral_gate.py:166, np.argsort(frame_energy[b])
 * [:n_quiet]): frame_idx (rows, n_selected) int32 = the frames with the smallest energies of each
 * row of energy (rows, frames), ascending, equal energies in frame order (a stable sort's
 * choice; the reference's unstable sort leaves it open), NaN last.  Rows of up to 8192 frames;
 * MIA_ERR_UNSUPPORTED beyond (the caller sorts). */
int mia_quietest_frames(const float* energy, int64_t rows, int64_t frames, int n_selected, int* frame_idx,
                        void* stream);

/* mia_stft with the frame energies the stationary gate ranks its frames by (spectral_gate.py:160,
 * np.mean(audio_mag ** 2, axis=1)) written beside the spectrum: energy (B, F) = mean over bins of
 * re^2 + im^2 of the values stored, summed in a fixed order (a shuffle tree per wave, waves in
 * order), so a row's energies do not depend on the batch around it; equal to
 * mia_frame_energy_spec on that spectrum to float32 rounding.  Plans of 1024, 2048 and 4096
 * points (mia_stft_energy_supported; MIA_ERR_UNSUPPORTED otherwise). */
int mia_stft_energy_supported(const mia_plan* plan);
int mia_stft_energy(const mia_plan* plan, const float* y, int64_t B, int64_t T, float scale, void* out,
                    float* energy, void* stream);

/* The stationary gate's mask pipeline and its inverse STFT in one launch (spectral_gate.py:
 * 124-238 from the soft mask to the waveform): spec is the UN-gated STFT, complex64 (B, F, bins)
 * rows as mia_stft writes them, thr (rows_thr, bins) with rows_thr = B or 1, and every bin enters
 * the inverse transform scaled by m' where m = clip((|z| - thr * gain) / (|z| + 1e-8), 0, 1) is
 * averaged over size_freq bins, then over size_freq2 bins, then over size_time frames (each a
 * mia_box_filter: ends repeated, 1 = none) and m' = keep ? 1 - p (1 - m) : m.  Neither the mask
 * nor the gated spectrum is written anywhere; the result equals mia_gate_mask_box_spec +
 * mia_gate_apply_spectrum + mia_istft to float32 rounding.  Plans of 1024, 2048 and 4096 points
 * at hop n_fft / 4, windows of up to 16 / 4 bins and 8 frames (mia_istft_gated_supported says so;
 * MIA_ERR_UNSUPPORTED otherwise).  out (B, out_len) as mia_istft's, no workspace. */
int mia_istft_gated_supported(const mia_plan* plan, int size_freq, int size_freq2, int size_time);
int mia_istft_gated(const mia_plan* plan, const void* spec, int64_t B, int64_t F, int64_t out_len,
                    float scale, const float* thr, int64_t rows_thr, float gain, int keep, float p,
                    int size_freq, int size_freq2, int size_time, float* out, void* stream);

/* ---- HTDemucs input normalisation (models/demucs/model.py:206-233) ---------------------
 * mia_normalize_rows: per row of n floats, mean and population standard deviation, then
 *   out = (x - mean) / (std + eps) (the time branch, model.py:231-233, a batch item's channels
 *   and samples being one row); stats: device (rows, 2) = {mean, std + eps}, what the model
 *   later multiplies and adds back (model.py:367); sums: device scratch of rows * 16 bytes.
 * mia_normalize_cac: the frequency branch (model.py:209-227): spec is a complex64 array
 *   [batch, channels, bins, frames] addressed through strides in complex elements (e.g. the
 *   strided view of mia_stft's (rows, frames, bins) output); its real and imaginary parts
 *   become planes 2c and 2c+1 of out: device float (batch, 2*channels, bins, frames),
 *   normalised by the mean and std over all of a batch item's planes. */
int mia_normalize_rows(const float* x, int64_t rows, int64_t n, float eps, float* out, float* stats,
                       void* sums, void* stream);
int mia_normalize_cac(const void* spec, int64_t batch, int channels, int64_t bins, int64_t frames,
                      int64_t stride_b, int64_t stride_c, int64_t stride_f, int64_t stride_t,
                      float eps, float* out, float* stats, void* sums, void* stream);
/* The two halves of mia_normalize_cac for a caller that computes the spectrum itself:
 * mia_stft_moments is mia_stft that also adds up, per group of `group` consecutive rows (a
 * batch item's channels), the sum and the sum of squares of the real and imaginary parts of
 * bins < bins_counted of frames [frame_lo, frame_hi) - the part of the spectrum HTDemucs keeps
 * (model.py:378-417) - into sums: device (B / group, 2) doubles, zeroed by the call (compiled FFT
 * sizes only, MIA_ERR_UNSUPPORTED otherwise).  mia_normalize_cac_sums is mia_normalize_cac
 * with those sums given instead of computed: no second pass over the spectrum. */
int mia_stft_moments(const mia_plan* plan, const float* y, int64_t B, int64_t T, float scale, void* out,
                     int group, int bins_counted, int64_t frame_lo, int64_t frame_hi, void* sums,
                     void* stream);
int mia_normalize_cac_sums(const void* spec, int64_t batch, int channels, int64_t bins, int64_t frames,
                           int64_t stride_b, int64_t stride_c, int64_t stride_f, int64_t stride_t,
                           float eps, float* out, float* stats, const void* sums, void* stream);
/* The way back (model.py:432-506, the complex-as-channels branch before the inverse STFT):
 * planes: device float (rows, 2, bins, frames), real then imaginary part of each row's
 * spectrogram, frames contiguous; out: device complex64 (rows, frames + 2 pad_frames,
 * bins + pad_bins) in the order mia_istft reads fastest, the appended bins (HTDemucs: the one
 * Nyquist bin it dropped) and the pad_frames frames on either side (HTDemucs: two) zero. */
int mia_cac_to_spectrum(const float* planes, int64_t rows, int64_t bins, int64_t frames, int pad_bins,
                        int pad_frames, void* out, void* stream);
/* The waveform side of the same fusion (model.py:231-233, 378-390): mia_pad_rows_moments writes
 * out[row] = zero_left zeros, x[row, :n], zero_right zeros (the padding HTDemucs applies before its
 * STFT) and, from the same read, adds the sum and the sum of squares of x over each group of `group`
 * consecutive rows into sums: device (rows / group, 2) doubles, zeroed by the call.
 * mia_normalize_rows_sums is mia_normalize_rows with those sums given (rows = the groups, n = group
 * times the row length). */
int mia_pad_rows_moments(const float* x, int64_t rows, int64_t n, int64_t zero_left, int64_t zero_right,
                         int group, float* out, void* sums, void* stream);
int mia_normalize_rows_sums(const float* x, int64_t rows, int64_t n, float eps, float* out, float* stats,
                            const void* sums, void* stream);

/* ---- Whisper encoder stem (models/whisper/layers/encoder.py:72-109) ----------------------
 * What AudioEncoder.__call__ does before its transformer blocks: x = mel^T (frames, n_mels);
 * gelu(conv1(x)) with Conv1d(n_mels, n_state, kernel 3, padding 1); gelu(conv2(.)) with
 * Conv1d(n_state, n_state, kernel 3, stride 2, padding 1); + positional_embedding[:frames'].
 * GELU is the exact (erf) form of mlx.nn.gelu.  `precision` chooses the arithmetic of the two
 * convolutions: MIA_STEM_F32 keeps weights, spectrogram and the intermediate in float32 and
 * multiplies on the float32 matrix instruction (v_mfma_f32_32x32x2_f32: a chain of fused
 * multiply-adds in float32) - what the reference's nn.Conv1d computes on a float32 checkpoint,
 * its default; MIA_STEM_F16 rounds weights, spectrogram and the conv1 -> GELU intermediate to
 * float16 and accumulates in float32 (the precision of a float16 checkpoint, 16 x the rate).
 * Bias, GELU and embedding are float32 either way.
 * conv*_w are in the reference's own layout (out_channels, kernel, in_channels), float32,
 * device memory; `packed` (mia_whisper_stem_weight_bytes bytes, 16-byte aligned device memory)
 * receives the operand form of that precision once per model.  mel: device float32 (clips,
 * n_mels, frames); out: device (clips, (frames - 1) / 2 + 1, n_state), float16 when out_f16
 * else float32; pos: device float32 (>= out frames, n_state) or NULL; workspace:
 * mia_whisper_stem_workspace_bytes bytes of device memory.  n_state must be a multiple of 128
 * (every Whisper size is); clips * frames < 2^31 per call. */
#define MIA_STEM_F16 0
#define MIA_STEM_F32 1
size_t mia_whisper_stem_weight_bytes(int n_mels, int n_state, int precision);
int mia_whisper_stem_pack(const float* conv1_w, const float* conv2_w, int n_mels, int n_state,
                          int precision, void* packed, void* stream);
size_t mia_whisper_stem_workspace_bytes(int64_t clips, int64_t frames, int n_mels, int n_state,
                                        int precision);
int mia_whisper_stem(const float* mel, int64_t clips, int64_t frames, int n_mels, int n_state,
                     int precision, const void* packed, const float* conv1_b, const float* conv2_b,
                     const float* pos, void* out, int out_f16, void* workspace, size_t workspace_bytes,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIA_H_ */

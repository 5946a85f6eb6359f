// Internal declarations shared by the translation units of libmia_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/mia.h"

// Shape of the n_fft = 400 / hop = 160 kernel's per-wave tables (logmel400.hip).
#define W400_WAVES 16     /* waves (FFT roles) per workgroup */
#define W400_MEL_MAXC 10  /* 8-bin chunks per wave, at most */
#define W400_MEL_REC 32   /* dwords per chunk record: wA[8], wB[8], ctrl, pad */

namespace mia {

void set_error(const char* fmt, ...);

#define MIA_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      ::mia::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                       __FILE__, __LINE__);                                        \
      return MIA_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

#define MIA_REQUIRE(cond, ...)                                                     \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      ::mia::set_error(__VA_ARGS__);                                               \
      return MIA_ERR_INVALID;                                                      \
    }                                                                              \
  } while (0)

// Sortable integer key of a float (monotone for all non-NaN values), so that a
// float maximum can be taken with an integer atomicMax.
__host__ __device__ inline int float_to_key(float f) {
  int b = __builtin_bit_cast(int, f);
  return b >= 0 ? b : (b ^ 0x7FFFFFFF);
}
__host__ __device__ inline float key_to_float(int k) {
  int b = k >= 0 ? k : (k ^ 0x7FFFFFFF);
  return __builtin_bit_cast(float, b);
}
constexpr int kKeyMinusInf = (int)0x80000000;  // below the key of -inf
constexpr int kKeyPlusInf = 0x7FFFFFFF;        // above the key of +inf
constexpr int kKeyNaN = 0x7FC00000;            // a NaN sorts above +inf: it wins every maximum,
                                               // as it does in the reference's max(...)
// Key of a value that may be a NaN of either sign.
__host__ __device__ inline int float_to_key_nan(float f) { return f != f ? kKeyNaN : float_to_key(f); }
// max(v, floor) with the reference's NaN rules (maximum(NaN, x) = maximum(x, NaN) = NaN).
__host__ __device__ inline float floor_nan(float v, float thr) {
  return thr != thr ? thr : (v < thr ? thr : v);
}

// Sample i of a signal of length T with centre padding folded in.
__device__ __forceinline__ float fetch_sample(const float* __restrict__ y, long long T,
                                              long long i, int pad_mode) {
  if (i >= 0 && i < T) return y[i];
  if (pad_mode == MIA_PAD_CONSTANT) return 0.0f;
  if (pad_mode == MIA_PAD_EDGE) return y[i < 0 ? 0 : T - 1];
  long long r = i < 0 ? -i : 2 * (T - 1) - i;  // reflect without repeating the edge
  r = r < 0 ? 0 : (r > T - 1 ? T - 1 : r);
  return y[r];
}
__device__ __forceinline__ float fetch_sample_any(const float* __restrict__ y, long long T,
                                                  long long i, int pad_mode) {
  return fetch_sample(y, T, i, pad_mode);
}

}  // namespace mia

// The plan: per-device tables for one (n_fft, hop, window, mel, dct) setting.
struct mia_plan {
  int device = 0;
  int n_fft = 0, hop = 0, center = 1, pad_mode = 0;
  float* d_window = nullptr;     // n_fft
  float2* d_tw = nullptr;        // n_fft entries exp(-2 pi i k / n_fft)
  // mel band table (row-gather form of the filterbank)
  int n_mels = 0, nnz = 0, max_row_len = 0;
  int* d_row_lo = nullptr;
  int* d_row_len = nullptr;
  int* d_row_off = nullptr;
  float* d_wts = nullptr;
  float power = 2.0f;
  // dct
  int n_mfcc = 0;
  float* d_dct = nullptr;        // n_mfcc x n_mels, rows pre-multiplied by the lifter
  // extra tables of the n_fft = 400 / hop = 160 kernel (null when not applicable)
  float* d_win16 = nullptr;      // [16][32] window taps t + 16 r of role t (25 used)
  float2* d_tw16 = nullptr;      // [16][16] exp(-2 pi i t k1 / 400), k1 = 0..12 used
  float* d_winp = nullptr;       // [16][32] window taps (25 n1 + 16 n2) mod 400 of role n1
  int* d_sweep_meta = nullptr;   // [16][16] per-wave sweep header: first bin, chunks, first filter, pre-emits
  float* d_sweep_w = nullptr;    // [16][W400_MEL_MAXC][32] per-wave sweep program (see logmel400.hip)
  int w400_bank = 0;             // 80 / 128 when the filterbank is that compiled-in Whisper bank
  int cus = 256;                 // compute units of the device (sizes the persistent grids)
  // any-length plan (n_fft without a compiled LDS kernel): whole-row engine of bigfft.hip
  int generic = 0;
  int wave_fft = 0;              // ... whose forward transform (spectrum and mel modes) the per-wave kernel of
                                 // spectral_wave.hip serves without a workspace (7-smooth: even n_fft <= 2048, odd <= 1023)
  float2* d_chirp = nullptr;     // chirp-z path only: exp(i pi n^2 / n_fft)
  float2* d_bspec = nullptr;     // ... and the spectrum of the convolution kernel
  int opt_table_sweep = 0;       // mia_plan_set_option("w400_table_sweep"): ablation / test knob
  int opt_vgpr_staging = 0;      // mia_plan_set_option("w400_vgpr_staging"): ablation / test knob
  int opt_mfma_mel = 0;          // mia_plan_set_option("w400_mfma_mel"): A/B knob, matrix-core projection
  int opt_dpp_pass2 = 0;         // mia_plan_set_option("w400_dpp_pass2"): A/B knob, cross-lane (DPP) radix-16 pass
  int opt_two_kernel_istft = 0;  // mia_plan_set_option("wave_istft_fused", 0): the per-wave inverse through its frame workspace (tests, A/B runs)
  int opt_lane = 0;              // mia_plan_set_option("fwd_lane"): A/B knob, frame-major mel rows at n_fft 1024 by
                                 // the wave-private kernel of spectral_lane.hip
  int* d_mfma_meta = nullptr;    // tables of that projection (n_mels <= 128)
  float* d_mfma_a = nullptr;
};

namespace mia {

// Tile order for kernels whose neighbouring tiles (along ty) touch the same cache lines.
// Workgroups are dealt to the eight XCDs in turn, each with its own L2, so a launch of
// xcd_tile_grid(gx, gy) workgroups along x hands XCD k the k-th eighth of the gx * gy tiles, ty
// running fastest: neighbours along ty sit behind one L2 and every XCD gets the same number of
// tiles.  xcd_tile() returns false for the few workgroups past the last tile.
static inline unsigned xcd_tile_grid(unsigned gx, unsigned gy) { return 8u * ((gx * gy + 7u) / 8u); }
__device__ __forceinline__ bool xcd_tile(unsigned gx, unsigned gy, unsigned& tx, unsigned& ty) {
  const unsigned total = gx * gy, per = (total + 7u) / 8u;
  const unsigned t = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  if (t >= total) return false;
  tx = t / gy;
  ty = t - tx * gy;
  return true;
}

// Arithmetic of the spectral gate shared by its kernels (so that a fused form gives the bits of
// the separate launches).  Mean of a window from its double-precision sum: one multiplication
// by the reciprocal a thread takes once.
constexpr float kGateEps = 1e-8f;  // the reference's DIVISION_EPSILON
__device__ __forceinline__ double window_inv(int size) { return 1.0 / (double)size; }
__device__ __forceinline__ float window_mean(double sum, double inv) { return (float)(sum * inv); }
// The soft mask value clip((m - t * gain) / (m + eps), 0, 1), with the product and the difference
// as one fused multiply-add spelled out: left to the compiler, the contraction differs between
// kernels.
__device__ __forceinline__ float gate_mask_value(float m, float t, float gain) {
  return fminf(fmaxf(__builtin_fmaf(-t, gain, m) / (m + kGateEps), 0.0f), 1.0f);
}

struct FwdParams {
  const float* y;
  long long B, T, F;
  int hop, pad, pad_mode, tiles, tiles_per_wg;
  const float* win;
  const float2* tw;
  float scale;
  float2* cout;
  // mel epilogue
  int n_mels;
  const int* row_lo;
  const int* row_len;
  const int* row_off;
  const float* wts;
  int nnz;         // entries of wts
  int lds_tables;  // the launcher found room for row_lo/row_len/row_off/wts in LDS
  int power_mode;  // 0: |X|^2, 1: |X|, 2: |X|^p
  float power;
  float* mout;
  int epi;         // 0: none, 1: whisper log10 -> (y+4)/4, 2: dB
  float amin, inv_ref, coef;
  float db_floor;      // coef*log10(amin/max(ref,amin)), the value of clamped inputs
  float db_floor_shift;  // coef*log10(1/max(ref,amin)): additive form of the division by ref
  int* clip_max_key;   // B (+1) keys of the per-clip maximum, or nullptr
  int* clip_min_key;   // B keys of the per-clip minimum of the stored values
  int frame_major;     // generic mel kernel: write (B, F, n_mels) rows instead of (B, n_mels, F)
  // moments epilogue (launch_forward mode 5): besides storing the spectrum, the sum and the sum
  // of squares of the real and imaginary parts of bins < mom_bins of frames [mom_f_lo, mom_f_hi)
  // are added to mom_sums[2 * (row / mom_group)] and [... + 1] (double, one atomic pair per tile)
  double* mom_sums;
  int mom_group, mom_bins;
  float* energy;  // mode 7: (B, F) mean over bins of |X|^2 of every frame, beside the spectrum
  long long mom_f_lo, mom_f_hi;
  int dbg;
  // spectral-statistics epilogue (epi == kEpiStats): centroid, bandwidth, roll-off, flatness
  const float* freq;   // bin frequencies
  float stat_p;        // bandwidth exponent
  int stat_norm;       // bandwidth normalised by the magnitude sum
  float stat_roll;     // roll-off fraction
  float flat_power;    // flatness is taken of magnitude^flat_power ...
  float flat_amin;     // ... floored here
};
constexpr int kEpiStats = 3;  // internal epilogue code (the public ones are MIA_EPI_*)

struct W400Tables {
  const float* win16;
  const float2* tw16;
  const float* winp;      // [16][32] window taps (25 n1 + 16 n2) mod 400 of role n1 (prime-factor split)
  const int* sweep_meta;
  const float* sweep_w;
  int compiled_bank;      // 80 / 128: the filterbank equals that compiled-in Whisper bank; else 0
  int force_table_sweep;  // testing: take the table-driven sweep although a compiled one exists
  int force_vgpr_staging; // testing: stage the samples through registers, not by LDS-DMA
  int cus;                // compute units of the plan's device
  const int* mfma_meta;   // matrix-core projection: [4 groups][2 blocks]{block, first bin, steps, A offset}
  const float* mfma_a;    // ... its A operands, [step][lane], zero-padded; null: filterbank not eligible
  int force_mfma;         // testing / A-B: project on the matrix cores instead of sweeping
  int force_dpp_pass2;    // testing / A-B: LDS-DMA kernel with the radix-16 pass across the lanes (DPP)
};

struct InvParams {
  const float2* spec;
  long long B, F;
  long long sb, sk, sf;  // strides in complex elements: batch, bin, frame
  const float2* tw;
  float* frames;         // (B, F, N)
  int tiles;
  // fused istft (inverse FFT + overlap-add) only
  const float* win;
  int hop, tiles_per_wg;
  long long out_len;
  float scale;
  float* out;            // (B, out_len)
};

// The gate applied to the spectrum on its way into the fused inverse STFT.
struct GateParams {
  const float* thr;      // (rows_thr, bins) thresholds
  long long thr_stride;  // bins, or 0: one row of thresholds for every batch row
  float gain, prop;
  int keep;              // m -> 1 - prop (1 - m) after the averages
  int s1, s2, st;        // windows: first and second average over bins, then over frames (1: none)
};

int launch_forward(const mia_plan* plan, const FwdParams& p, int mode, hipStream_t s);
bool istft_gated_supported(int n_fft, int hop, int s1, int s2, int st);
int launch_istft_gated(const mia_plan* plan, const InvParams& p, const GateParams& g, hipStream_t s);
int launch_inverse(const mia_plan* plan, const InvParams& p, hipStream_t s);
bool istft_fused_supported(int n_fft, int hop);
int launch_istft_fused(const mia_plan* plan, const InvParams& p, hipStream_t s);
bool istft_wave_fused_supported(int n_fft, int hop);
int forward_wave_fixed_plan(int n_fft);
int launch_istft_wave_fused(const mia_plan* plan, const InvParams& p, hipStream_t s);
int launch_init_keys(int* max_keys, int* min_keys, long long n, hipStream_t s);
int launch_logmel400(const FwdParams& p, const W400Tables& tb, hipStream_t s);
int w400_compiled_bank(const float* basis, int n_mels, int bins);
bool fft_size_supported(int n_fft);
long long generic_row_len(int n_fft);
long long generic_chunk_rows(int n_fft, long long rows);
size_t generic_fft_bytes(int n_fft, long long rows);
int generic_build_tables(int n_fft, float2** d_chirp, float2** d_bspec);
int launch_forward_generic(const mia_plan* plan, const FwdParams& p, int mode, void* workspace,
                           size_t workspace_bytes, hipStream_t s);
int launch_inverse_generic(const mia_plan* plan, const InvParams& p, void* workspace,
                           size_t workspace_bytes, hipStream_t s);
int forward_tile_frames(int n_fft, int mode);
bool forward_frame_major_ok(int n_fft, int n_mels);
bool forward_lane_ok(int n_fft, int hop, int n_mels, int nnz);
int launch_forward_lane(const FwdParams& p, int cus, hipStream_t s);
bool forward_wave_plan(int n_fft, int* n_pass, int radix[6], int* lanes_per_frame = nullptr);
bool forward_wave_frame_major_ok(int n_fft, int n_mels);
int launch_forward_wave(const mia_plan* plan, const FwdParams& p, int mode, hipStream_t s);
int launch_inverse_wave(const mia_plan* plan, const InvParams& p, hipStream_t s);
bool dct2_frame_major_ok(int n_in, int n_out);
int launch_dct2_frame_major(const float* x, long long outer, int n_in, long long inner, const float* basis,
                            int n_out, float* out, const int* keys, long long key_stride, float top,
                            hipStream_t s);

}  // namespace mia

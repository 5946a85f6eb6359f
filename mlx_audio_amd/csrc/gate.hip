// Spectral gating: the steps between the forward and the inverse STFT of a noise gate, on
// magnitudes laid out (rows, bins, frames) with frames contiguous.  The elementwise steps and
// the reductions are one thread per output element (per frame / per bin) with consecutive lanes
// on consecutive frames, so loads are coalesced row segments.  The two kernels a default call
// spends its time in - mask + first frequency average, and the closing kernel that averages the
// mask once more and scales the complex spectrum by it - work on LDS tiles (every mask value
// computed once, loads independent of each other), handed out in xcd_tile() order; their
// one-thread-per-output forms stay for windows wider than the tiles' halo.
#include "mia_internal.h"

namespace mia {

// energy[row, f] = mean over bins of mag^2.  A block takes 32 frames; eight groups of threads
// each sum every eighth bin (one thread per frame could keep only a few hundred waves in
// flight, each behind a chain of `bins` loads), and the eight partial sums are added in a fixed
// order, so the result does not depend on the launch.
__global__ __launch_bounds__(256) void frame_energy_kernel(const float* __restrict__ mag, int bins,
                                                           long long F, float* __restrict__ energy) {
  __shared__ float part[8][33];
  const long long row = blockIdx.y;
  const int fx = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long long f = (long long)blockIdx.x * 32 + fx;
  float acc = 0.0f;
  if (f < F) {
    const float* __restrict__ col = mag + row * bins * F + f;
#pragma unroll 8
    for (int k = g; k < bins; k += 8) {
      const float v = col[(long long)k * F];
      acc += v * v;
    }
  }
  part[g][fx] = acc;
  __syncthreads();
  if (g == 0 && f < F) {
    float sum = part[0][fx];
#pragma unroll
    for (int q = 1; q < 8; ++q) sum += part[q][fx];
    energy[row * F + f] = sum / (float)bins;
  }
}

// thr[row, bin] = mean + n_std * std (population) of mag[row, bin, idx[row, q]], q < n_sel;
// idx == nullptr selects every frame.
__global__ __launch_bounds__(256) void noise_threshold_kernel(const float* __restrict__ mag,
                                                              long long rows, int bins, long long F,
                                                              const int* __restrict__ idx, int n_sel,
                                                              float n_std, float* __restrict__ thr) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * bins) return;
  const long long row = g / bins;
  const float* __restrict__ line = mag + g * F;
  const int* __restrict__ sel = idx ? idx + row * n_sel : nullptr;
  const int n = sel ? n_sel : (int)F;
  float sum = 0.0f;
  for (int q = 0; q < n; ++q) sum += line[sel ? sel[q] : q];
  const float mean = sum / (float)n;
  float ss = 0.0f;
  for (int q = 0; q < n; ++q) {
    const float d = line[sel ? sel[q] : q] - mean;
    ss = __builtin_fmaf(d, d, ss);
  }
  thr[g] = mean + n_std * sqrtf(ss / (float)n);
}

// out[r, i] = min of x[r, clamp(j)] over j in [i - size/2, i + (size-1)/2]
template <class Index>
__global__ __launch_bounds__(256) void time_min_filter_kernel(const float* __restrict__ x,
                                                              long long rows, long long n, int size,
                                                              float* __restrict__ out) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * n) return;
  const Index rq = (Index)g / (Index)n;
  const long long r = (long long)rq, i = g - r * n;
  const float* __restrict__ line = x + r * n;
  float m = INFINITY;
  for (long long j = i - size / 2; j <= i + (size - 1) / 2; ++j) {
    const long long q = j < 0 ? 0 : (j >= n ? n - 1 : j);
    m = fminf(m, line[q]);
  }
  out[g] = m;
}

// Moving average along the middle axis of x viewed as (outer, n, inner), window
// [i - size/2, i + (size-1)/2] with indices clamped to the ends; the sum runs in double.
template <class Index>  // unsigned when the element count fits 32 bits: index division is most of the cost
__global__ __launch_bounds__(256) void box_filter_kernel(const float* __restrict__ x,
                                                         long long outer, long long n,
                                                         long long inner, int size,
                                                         float* __restrict__ out) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= outer * n * inner) return;
  const Index gi = (Index)g, q1 = gi / (Index)inner;
  const long long in = (long long)(gi - q1 * (Index)inner);
  const Index q2 = q1 / (Index)n;
  const long long i = (long long)(q1 - q2 * (Index)n);
  const long long o = (long long)q2;
  const float* __restrict__ line = x + o * n * inner + in;
  double sum = 0.0;
  for (long long j = i - size / 2; j <= i + (size - 1) / 2; ++j) {
    const long long q = j < 0 ? 0 : (j >= n ? n - 1 : j);
    sum += (double)line[q * inner];
  }
  out[g] = window_mean(sum, window_inv(size));
}

// mask = clip((mag - thr * gain) / (mag + eps), 0, 1); thr is (rows, bins, frames) or, with
// per_bin, (rows_thr, bins) broadcast over frames (rows_thr = rows, or 1 shared by all rows)
__global__ __launch_bounds__(256) void gate_mask_kernel(const float* __restrict__ mag,
                                                        const float* __restrict__ thr, long long rows,
                                                        int bins, long long F, int per_bin,
                                                        long long rows_thr, float gain,
                                                        float* __restrict__ mask) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * bins * F) return;
  float t;
  if (per_bin) {
    const long long rb = g / F;  // row * bins + bin
    t = thr[rows_thr == 1 ? rb % bins : rb];
  } else {
    t = thr[g];
  }
  const float m = mag[g];
  mask[g] = gate_mask_value(m, t, gain);
}

// The soft mask, optionally the keep step, and a moving average over `size` bins in one pass:
// out[row, k, f] = mean over kk in [k - size/2, k + (size-1)/2] (clamped) of m(row, kk, f),
// m = clip((mag - thr * gain) / (mag + eps), 0, 1), then 1 - p (1 - m) if keep.  Every m is the
// float32 value the separate kernels would have stored, the sum runs in double: bit-identical
// to gate_mask_kernel (+ mask_op) + box_filter_kernel along the bins.
__global__ __launch_bounds__(256) void gate_mask_box_kernel(const float* __restrict__ mag,
                                                            const float* __restrict__ thr,
                                                            unsigned rows, unsigned bins, unsigned F,
                                                            int per_bin, unsigned rows_thr, float gain,
                                                            int keep, float p, int size,
                                                            float* __restrict__ out) {
  const unsigned f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const unsigned k = blockIdx.y, row = blockIdx.z;
  const size_t plane = (size_t)row * bins * F;
  const double inv = window_inv(size);
  double sum = 0.0;
  for (int j = (int)k - size / 2; j <= (int)k + (size - 1) / 2; ++j) {
    const unsigned q = j < 0 ? 0u : (j >= (int)bins ? bins - 1 : (unsigned)j);
    const size_t i = plane + (size_t)q * F + f;
    const float t = per_bin ? thr[(rows_thr == 1 ? 0 : (size_t)row * bins) + q] : thr[i];
    const float m = mag[i];
    float v = gate_mask_value(m, t, gain);
    if (keep) v = 1.0f - p * (1.0f - v);
    sum += (double)v;
  }
  out[plane + (size_t)k * F + f] = size == 1 ? (float)sum : window_mean(sum, inv);
}

// gate_mask_box_kernel for windows of up to kGateMaxWin bins, bit-identical to it: a block takes
// 32 bins x 64 frames, computes the mask values of its bins and the window's halo once, in one
// sweep of independent loads, into LDS, and sums the windows from there (the one-thread-per-
// output form re-derives every mask value `size` times behind a chain of dependent loads).
constexpr int kGateKT = 32, kGateFT = 64, kGateMaxWin = 33;
__global__ __launch_bounds__(256) void gate_mask_box_tiled_kernel(
    const float* __restrict__ mag, const float* __restrict__ thr, unsigned bins, unsigned F,
    int per_bin, unsigned rows_thr, float gain, int keep, float p, int size,
    float* __restrict__ out) {
  __shared__ float vt[kGateKT + kGateMaxWin - 1][kGateFT];
  const unsigned row = blockIdx.z;
  unsigned tile_f, tile_k;
  if (!xcd_tile((F + kGateFT - 1) / kGateFT, (bins + kGateKT - 1) / kGateKT, tile_f, tile_k)) return;
  const int k0 = (int)tile_k * kGateKT;
  const unsigned fx = threadIdx.x & (kGateFT - 1), ky = threadIdx.x / kGateFT;
  const unsigned f = tile_f * kGateFT + fx;
  const unsigned fc = f < F ? f : F - 1;
  const size_t plane = (size_t)row * bins * F;
  const int nrows = kGateKT + size - 1;
  const double inv = window_inv(size);
  for (int kk = (int)ky; kk < nrows; kk += 256 / kGateFT) {
    const int j = k0 - size / 2 + kk;
    const unsigned q = j < 0 ? 0u : (j >= (int)bins ? bins - 1 : (unsigned)j);
    const size_t i = plane + (size_t)q * F + fc;
    const float t = per_bin ? thr[(rows_thr == 1 ? 0 : (size_t)row * bins) + q] : thr[i];
    const float m = mag[i];
    float v = gate_mask_value(m, t, gain);
    if (keep) v = 1.0f - p * (1.0f - v);
    vt[kk][fx] = v;
  }
  __syncthreads();
  if (f >= F) return;
  for (int kl = (int)ky; kl < kGateKT; kl += 256 / kGateFT) {
    const unsigned k = (unsigned)(k0 + kl);
    if (k >= bins) break;
    double sum = 0.0;
    for (int j = 0; j < size; ++j) sum += (double)vt[kl + j][fx];
    out[plane + (size_t)k * F + f] = size == 1 ? (float)sum : window_mean(sum, inv);
  }
}

// op 0: out = 1 - p (1 - a);  op 1: out = a * b
__global__ __launch_bounds__(256) void mask_op_kernel(const float* __restrict__ a,
                                                      const float* __restrict__ b, long long n,
                                                      int op, float p, float* __restrict__ out) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  out[g] = op == 0 ? 1.0f - p * (1.0f - a[g]) : a[g] * b[g];
}

// The last steps of the gate in one pass: optionally the moving average of the mask over
// size_time frames, m = keep ? 1 - p (1 - mask) : mask, then
// mag * m * exp(i phase), read as (rows, bins, frames) planes and written as the
// (rows, frames, bins) complex spectrum the inverse STFT kernel reads with coalesced rows;
// 32 x 32 tiles through LDS.  grid (tiles over frames, tiles over bins, rows), block (32, 8).
__global__ __launch_bounds__(256) void gate_apply_kernel(const float* __restrict__ mag,
                                                         const float* __restrict__ mask,
                                                         const float* __restrict__ phase,
                                                         long long bins, long long F, int keep,
                                                         float p, int size_time,
                                                         float2* __restrict__ out) {
  __shared__ float2 tile[32][33];
  const long long row = blockIdx.z;
  const long long f0 = (long long)blockIdx.x * 32, k0 = (long long)blockIdx.y * 32;
  const long long base = row * bins * F;
  const double inv_time = window_inv(size_time);
  for (int j = threadIdx.y; j < 32; j += 8) {  // j: bin in the tile, x: frame
    const long long k = k0 + j, f = f0 + threadIdx.x;
    float2 v = {0.0f, 0.0f};
    if (k < bins && f < F) {
      const long long i = base + k * F + f;
      float m = mask[i];
      if (size_time > 1) {  // moving average over the frames first, as box_filter_kernel does it
        double sum = 0.0;
        for (long long j = f - size_time / 2; j <= f + (size_time - 1) / 2; ++j) {
          const long long q = j < 0 ? 0 : (j >= F ? F - 1 : j);
          sum += (double)mask[base + k * F + q];
        }
        m = window_mean(sum, inv_time);
      }
      if (keep) m = 1.0f - p * (1.0f - m);
      const float a = mag[i] * m;
      float sn, cs;
      sincosf(phase[i], &sn, &cs);
      v = {a * cs, a * sn};
    }
    tile[j][threadIdx.x] = v;
  }
  __syncthreads();
  float2* __restrict__ dst = out + base;
  for (int j = threadIdx.y; j < 32; j += 8) {  // j: frame in the tile, x: bin
    const long long f = f0 + j, k = k0 + threadIdx.x;
    if (k < bins && f < F) dst[f * bins + k] = tile[threadIdx.x][j];
  }
}

// gate_apply_kernel for callers that still hold the complex spectrum: mag * exp(i phase) is the
// spectrum itself, so out[row, f, k] = spec[row, f, k] * m(row, k, f) with the mask prepared as
// above and, before that, optionally averaged over size_freq bins (each frequency average
// rounded to float32 as the separate box_filter_kernel launch would store it, so the result is
// bit-identical to that launch followed by this kernel).  The spectrum is read and written in
// its own (rows, frames, bins) order; only the mask tile turns through LDS.  No magnitude or
// phase planes are read and no sine / cosine is taken.
__global__ __launch_bounds__(256) void gate_apply_spectrum_kernel(const float2* __restrict__ spec,
                                                                  const float* __restrict__ mask,
                                                                  long long bins, long long F,
                                                                  int keep, float p, int size_freq,
                                                                  int size_time,
                                                                  float2* __restrict__ out) {
  __shared__ float tile[32][33];
  const long long row = blockIdx.z;
  const long long f0 = (long long)blockIdx.x * 32, k0 = (long long)blockIdx.y * 32;
  const long long base = row * bins * F;
  const double inv_freq = window_inv(size_freq), inv_time = window_inv(size_time);
  for (int j = threadIdx.y; j < 32; j += 8) {  // j: bin in the tile, x: frame
    const long long k = k0 + j, f = f0 + threadIdx.x;
    float m = 0.0f;
    if (k < bins && f < F) {
      auto over_bins = [&](long long q) {  // the mask at (k, frame q) after the frequency average
        if (size_freq == 1) return mask[base + k * F + q];
        double sum = 0.0;
        for (long long j0 = k - size_freq / 2; j0 <= k + (size_freq - 1) / 2; ++j0) {
          const long long kk = j0 < 0 ? 0 : (j0 >= bins ? bins - 1 : j0);
          sum += (double)mask[base + kk * F + q];
        }
        return window_mean(sum, inv_freq);
      };
      if (size_time > 1) {
        double sum = 0.0;
        for (long long q0 = f - size_time / 2; q0 <= f + (size_time - 1) / 2; ++q0) {
          sum += (double)over_bins(q0 < 0 ? 0 : (q0 >= F ? F - 1 : q0));
        }
        m = window_mean(sum, inv_time);
      } else {
        m = over_bins(f);
      }
      if (keep) m = 1.0f - p * (1.0f - m);
    }
    tile[j][threadIdx.x] = m;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {  // j: frame in the tile, x: bin
    const long long f = f0 + j, k = k0 + threadIdx.x;
    if (k < bins && f < F) {
      const long long i = base + f * bins + k;
      const float2 v = spec[i];
      const float m = tile[threadIdx.x][j];
      out[i] = {v.x * m, v.y * m};
    }
  }
}

// gate_apply_spectrum_kernel for time windows of up to kGateMaxWin frames, bit-identical to it,
// in two phases with one barrier, like a plain transposing copy.  Phase one: the block's patch of
// the mask (32 bins x 64 frames plus the time window's halo), averaged over size_freq bins on the
// way in (independent loads that hit the caches), lands in LDS with frames along the rows.
// Phase two: each thread takes the time average of its element straight from that patch (read
// across the rows: conflict-free with the odd pitch), applies the keep step and scales the
// spectrum values it fetched before phase one.
constexpr int kApplyTF = 64;  // frames per tile of the closing kernel (32: 3 % slower, shorter runs of the mask rows)
__global__ __launch_bounds__(256) void gate_apply_spectrum_tiled_kernel(
    const float2* __restrict__ spec, const float* __restrict__ mask, long long bins, long long F,
    int keep, float p, int size_freq, int size_time, float2* __restrict__ out) {
  constexpr int W = kApplyTF + kGateMaxWin - 1;  // frames of the patch at most
  __shared__ float avg[32][W + 1];
  const long long row = blockIdx.z;
  unsigned tile_f, tile_k;
  if (!xcd_tile((unsigned)((F + kApplyTF - 1) / kApplyTF), (unsigned)((bins + 31) / 32), tile_f, tile_k)) return;
  const long long f0 = (long long)tile_f * kApplyTF, k0 = (long long)tile_k * 32;
  const long long base = row * bins * F;
  const int tid = threadIdx.y * 32 + threadIdx.x;
  const int nf = kApplyTF + size_time - 1;
  const double inv_freq = window_inv(size_freq), inv_time = window_inv(size_time);
  constexpr int kPer = kApplyTF / 8;
  float2 sv[kPer];  // the thread's spectrum values, on their way while the mask is prepared
#pragma unroll
  for (int jj = 0; jj < kPer; ++jj) {
    const long long f = f0 + threadIdx.y + 8 * jj, k = k0 + threadIdx.x;
    sv[jj] = (k < bins && f < F) ? spec[base + f * bins + k] : float2{0.0f, 0.0f};
  }
  const unsigned magic = (unsigned)((0x100000000ULL + (unsigned)nf - 1) / (unsigned)nf);
  for (int idx = tid; idx < 32 * nf; idx += 256) {
    const int kl = (int)__umulhi((unsigned)idx, magic);  // idx / nf (exact: idx < 2^16)
    const int ff = idx - kl * nf;
    const long long fq = f0 - size_time / 2 + ff;
    const long long fc = fq < 0 ? 0 : (fq >= F ? F - 1 : fq);
    const long long k = k0 + kl;
    float v;
    if (size_freq == 1) {
      v = mask[base + (k < bins ? k : bins - 1) * F + fc];
    } else {
      double sum = 0.0;
#pragma unroll 4
      for (long long j0 = k - size_freq / 2; j0 <= k + (size_freq - 1) / 2; ++j0) {
        const long long kc = j0 < 0 ? 0 : (j0 >= bins ? bins - 1 : j0);
        sum += (double)mask[base + kc * F + fc];
      }
      v = window_mean(sum, inv_freq);
    }
    avg[kl][ff] = v;
  }
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < kPer; ++jj) {  // frame threadIdx.y + 8 jj of the tile, x: bin
    const int j = threadIdx.y + 8 * jj;
    const long long f = f0 + j, k = k0 + threadIdx.x;
    if (k < bins && f < F) {
      float m = avg[threadIdx.x][j];
      if (size_time > 1) {
        double sum = 0.0;
        for (int q = 0; q < size_time; ++q) sum += (double)avg[threadIdx.x][j + q];
        m = window_mean(sum, inv_time);
      }
      if (keep) m = 1.0f - p * (1.0f - m);
      out[base + f * bins + k] = {sv[jj].x * m, sv[jj].y * m};
    }
  }
}

// ---------------------------------------------------------------------------
// The stationary gate straight from the complex spectrum (rows, frames, bins) the STFT kernel
// wrote: frame energy, noise threshold and mask + first frequency average take |z| = hypotf(re,
// im) as they go - the value the magnitude plane would have held - so that plane (one write and
// three reads of a spectrum-sized float array, plus its launch) is never made.
// ---------------------------------------------------------------------------
// energy[row, f] = mean over bins of |z|^2: one wave per frame (its bins are contiguous), lanes
// stride the bins, a fixed shuffle tree adds the lanes.
__global__ __launch_bounds__(256) void frame_energy_spec_kernel(const float2* __restrict__ spec,
                                                                long long rows, int bins, long long F,
                                                                float* __restrict__ energy) {
  const long long frame = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // over rows * F
  if (frame >= rows * F) return;  // whole waves leave together
  const int lane = threadIdx.x & 63;
  const float2* __restrict__ line = spec + frame * bins;
  float acc = 0.0f;
#pragma unroll 4
  for (int k = lane; k < bins; k += 64) {
    const float2 z = line[k];
    const float m = hypotf(z.x, z.y);
    acc += m * m;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) energy[frame] = acc / (float)bins;
}

// thr[row, bin] = mean + n_std * std (population) of |spec[row, idx[row, q], bin]|, q < n_sel (idx ==
// nullptr: every frame), summed in the order noise_threshold_kernel sums them: bit-identical to
// it on the magnitude plane.  A block takes 64 bins of a row: its four waves turn 64 frames'
// values at a time into magnitudes in LDS (sixteen independent loads per thread, each a run of
// 64 complex values), then one thread per bin adds them up in frame order - the sums are
// serial by definition, the loads and square roots in front of them are not.
constexpr int kThrBins = 64, kThrFrames = 64;
__global__ __launch_bounds__(256) void noise_threshold_spec_kernel(const float2* __restrict__ spec,
                                                                   int bins, long long F,
                                                                   const int* __restrict__ idx, int n_sel,
                                                                   float n_std, float* __restrict__ thr) {
  __shared__ float mags[kThrFrames][kThrBins];
  __shared__ float means[kThrBins];
  const long long row = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = blockIdx.x * kThrBins + lane;
  const int kc = k < bins ? k : bins - 1;
  const float2* __restrict__ col = spec + row * F * bins + kc;
  const int* __restrict__ sel = idx ? idx + row * n_sel : nullptr;
  const int n = sel ? n_sel : (int)F;
  auto stage = [&](int c0) {  // magnitudes of frames c0 .. c0 + 63 of the selection
#pragma unroll
    for (int i = 0; i < kThrFrames / 4; ++i) {
      const int q = w + 4 * i;
      if (c0 + q < n) {
        const float2 z = col[(long long)(sel ? sel[c0 + q] : c0 + q) * bins];
        mags[q][lane] = hypotf(z.x, z.y);
      }
    }
    __syncthreads();
  };
  float sum = 0.0f;
  for (int c0 = 0; c0 < n; c0 += kThrFrames) {
    stage(c0);
    if (w == 0) {
      const int m = n - c0 < kThrFrames ? n - c0 : kThrFrames;
      for (int q = 0; q < m; ++q) sum += mags[q][lane];
    }
    __syncthreads();
  }
  if (w == 0) means[lane] = sum / (float)n;
  float ss = 0.0f;
  for (int c0 = 0; c0 < n; c0 += kThrFrames) {
    if (n > kThrFrames || c0 > 0) stage(c0); else __syncthreads();  // a single chunk is still there
    if (w == 0) {
      const float mean = means[lane];
      const int m = n - c0 < kThrFrames ? n - c0 : kThrFrames;
      for (int q = 0; q < m; ++q) {
        const float d = mags[q][lane] - mean;
        ss = __builtin_fmaf(d, d, ss);
      }
    }
    __syncthreads();
  }
  if (w == 0 && k < bins) thr[row * bins + k] = means[lane] + n_std * sqrtf(ss / (float)n);
}

// gate_mask_box_tiled_kernel with the magnitudes taken from the spectrum: the tile's bins (and
// the window's halo) of 64 frames are read frame by frame (runs of up to 64 complex values),
// turned into mask values and laid down in LDS with frames along the rows; the window sums and
// the (rows, bins, frames) store are those of the magnitude form, bit for bit.  thr is per bin.
__global__ __launch_bounds__(256) void gate_mask_box_spec_kernel(
    const float2* __restrict__ spec, const float* __restrict__ thr, unsigned bins, unsigned F,
    unsigned rows_thr, float gain, int keep, float p, int size, float* __restrict__ out) {
  __shared__ float vt[kGateKT + kGateMaxWin - 1][kGateFT + 1];
  const unsigned row = blockIdx.z;
  unsigned tile_f, tile_k;
  if (!xcd_tile((F + kGateFT - 1) / kGateFT, (bins + kGateKT - 1) / kGateKT, tile_f, tile_k)) return;
  const int k0 = (int)tile_k * kGateKT;
  const unsigned f0 = tile_f * kGateFT;
  const int nrows = kGateKT + size - 1;
  const double inv = window_inv(size);
  const float2* __restrict__ plane = spec + (size_t)row * F * bins;
  const float* __restrict__ thr_row = thr + (rows_thr == 1 ? 0 : (size_t)row * bins);
  {
    const int kk = threadIdx.x & 63;  // bin of the patch (nrows <= 64)
    if (kk < nrows) {
      const int j = k0 - size / 2 + kk;
      const unsigned q = j < 0 ? 0u : (j >= (int)bins ? bins - 1 : (unsigned)j);
      const float t = thr_row[q];
#pragma unroll 4
      for (unsigned fl = threadIdx.x >> 6; fl < (unsigned)kGateFT; fl += 4) {
        const unsigned f = f0 + fl, fc = f < F ? f : F - 1;
        const float2 z = plane[(size_t)fc * bins + q];
        const float m = hypotf(z.x, z.y);
        float v = gate_mask_value(m, t, gain);
        if (keep) v = 1.0f - p * (1.0f - v);
        vt[kk][fl] = v;
      }
    }
  }
  __syncthreads();
  const unsigned fx = threadIdx.x & (kGateFT - 1), ky = threadIdx.x / kGateFT;
  const unsigned f = f0 + fx;
  if (f >= F) return;
  const size_t oplane = (size_t)row * bins * F;
  for (int kl = (int)ky; kl < kGateKT; kl += 256 / kGateFT) {
    const unsigned k = (unsigned)(k0 + kl);
    if (k >= bins) break;
    double sum = 0.0;
    for (int j = 0; j < size; ++j) sum += (double)vt[kl + j][fx];
    out[oplane + (size_t)k * F + f] = size == 1 ? (float)sum : window_mean(sum, inv);
  }
}

// idx[row, r] = the frame with the r-th smallest energy of its row, r < n_sel, equal energies in
// frame order (a stable ascending sort's first n_sel indices; NaN sorts last).  A frame's rank
// is the count of frames that sort before it, counted against the row's keys in LDS (every lane
// reads the same key: a broadcast).  A block ranks 64 frames, its four waves a quarter of the
// row each - a few hundred frames a row, where a general sort spends most of its time in launches.
constexpr int kQuietMaxFrames = 8192;
__global__ __launch_bounds__(256) void quietest_frames_kernel(const float* __restrict__ energy, int F,
                                                              int n_sel, int* __restrict__ idx) {
  extern __shared__ int quiet_keys[];  // F keys, then 4 x 64 partial ranks
  int* part = quiet_keys + F;
  const long long row = blockIdx.y;
  for (int i = threadIdx.x; i < F; i += 256) quiet_keys[i] = float_to_key_nan(energy[row * F + i] + 0.0f);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int ki = quiet_keys[i < F ? i : F - 1];
  const int per = (F + 3) / 4, j0 = w * per, j1 = j0 + per < F ? j0 + per : F;
  int rank = 0;
#pragma unroll 8
  for (int j = j0; j < j1; ++j) {
    const int kj = quiet_keys[j];
    rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
  }
  part[w * 64 + lane] = rank;
  __syncthreads();
  if (w == 0 && i < F) {
    rank = part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane];
    if (rank < n_sel) idx[row * n_sel + rank] = i;
  }
}

static inline unsigned gate_blocks(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace mia

extern "C" {

int mia_frame_energy(const float* mag, int64_t rows, int bins, int64_t frames, float* energy,
                     void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 1 && frames >= 0, "rows, bins, frames must be >= 0, 1, 0");
  if (rows == 0 || frames == 0) return MIA_OK;
  MIA_REQUIRE(mag && energy, "null pointer");
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {  // a launch takes 65535 rows
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    hipLaunchKernelGGL(frame_energy_kernel, dim3((unsigned)((frames + 31) / 32), (unsigned)nr),
                       dim3(256), 0, (hipStream_t)stream, mag + (size_t)r0 * bins * frames, bins,
                       (long long)frames, energy + (size_t)r0 * frames);
  }
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_noise_threshold(const float* mag, int64_t rows, int bins, int64_t frames, const int* frame_idx,
                        int n_selected, float n_std, float* thr, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 1 && frames >= 1, "rows, bins, frames must be >= 0, 1, 1");
  MIA_REQUIRE(frame_idx == nullptr || n_selected >= 1, "n_selected must be >= 1");
  MIA_REQUIRE(frames <= 0x7fffffffLL, "too many frames");
  if (rows == 0) return MIA_OK;
  MIA_REQUIRE(mag && thr, "null pointer");
  hipLaunchKernelGGL(noise_threshold_kernel, dim3(gate_blocks(rows * bins)), dim3(256), 0,
                     (hipStream_t)stream, mag, (long long)rows, bins, (long long)frames, frame_idx,
                     n_selected, n_std, thr);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_time_min_filter(const float* x, int64_t rows, int64_t n, int size, float* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && n >= 0 && size >= 1, "rows, n, size must be >= 0, 0, 1");
  if (rows == 0 || n == 0) return MIA_OK;
  MIA_REQUIRE(x && out && x != out, "null or aliased pointers");
  if ((long long)rows * n < (1LL << 31)) {
    hipLaunchKernelGGL(time_min_filter_kernel<unsigned>, dim3(gate_blocks(rows * n)), dim3(256), 0,
                       (hipStream_t)stream, x, (long long)rows, (long long)n, size, out);
  } else {
    hipLaunchKernelGGL(time_min_filter_kernel<unsigned long long>, dim3(gate_blocks(rows * n)),
                       dim3(256), 0, (hipStream_t)stream, x, (long long)rows, (long long)n, size, out);
  }
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_box_filter(const float* x, int64_t outer, int64_t n, int64_t inner, int size, float* out,
                   void* stream) {
  using namespace mia;
  MIA_REQUIRE(outer >= 0 && n >= 0 && inner >= 0 && size >= 1,
              "outer, n, inner, size must be >= 0, 0, 0, 1");
  if (outer == 0 || n == 0 || inner == 0) return MIA_OK;
  MIA_REQUIRE(x && out && x != out, "null or aliased pointers");
  if ((long long)outer * n * inner < (1LL << 31)) {
    hipLaunchKernelGGL(box_filter_kernel<unsigned>, dim3(gate_blocks(outer * n * inner)), dim3(256),
                       0, (hipStream_t)stream, x, (long long)outer, (long long)n, (long long)inner,
                       size, out);
  } else {
    hipLaunchKernelGGL(box_filter_kernel<unsigned long long>, dim3(gate_blocks(outer * n * inner)),
                       dim3(256), 0, (hipStream_t)stream, x, (long long)outer, (long long)n,
                       (long long)inner, size, out);
  }
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_gate_mask(const float* mag, const float* thr, int64_t rows, int bins, int64_t frames,
                  int per_bin, int64_t rows_thr, float gain, float* mask, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 1 && frames >= 0, "rows, bins, frames must be >= 0, 1, 0");
  MIA_REQUIRE(rows_thr == rows || (per_bin && rows_thr == 1),
              "the threshold must have one row per signal row, or a single row of per-bin values");
  if (rows == 0 || frames == 0) return MIA_OK;
  MIA_REQUIRE(mag && thr && mask, "null pointer");
  hipLaunchKernelGGL(gate_mask_kernel, dim3(gate_blocks(rows * bins * frames)), dim3(256), 0,
                     (hipStream_t)stream, mag, thr, (long long)rows, bins, (long long)frames,
                     per_bin, (long long)rows_thr, gain, mask);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_gate_mask_box(const float* mag, const float* thr, int64_t rows, int bins, int64_t frames,
                      int per_bin, int64_t rows_thr, float gain, int keep, float p, int size_freq,
                      float* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 1 && frames >= 0 && size_freq >= 1,
              "rows, bins, frames, size_freq must be >= 0, 1, 0, 1");
  MIA_REQUIRE(rows_thr == rows || (per_bin && rows_thr == 1),
              "the threshold must have one row per signal row, or a single row of per-bin values");
  if (rows == 0 || frames == 0) return MIA_OK;
  MIA_REQUIRE(mag && thr && out, "null pointer");
  MIA_REQUIRE(bins <= 65535 && frames <= 0x7fffffffLL, "shape too large for one launch");
  if (rows > 65535) {  // a launch takes 65535 rows: the rest follows in further launches
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
      const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
      const float* thr_r = rows_thr == rows ? thr + (size_t)r0 * (per_bin ? bins : bins * frames) : thr;
      const int rc = mia_gate_mask_box(mag + (size_t)r0 * bins * frames, thr_r, nr, bins, frames,
                                       per_bin, rows_thr == rows ? nr : 1, gain, keep, p, size_freq,
                                       out + (size_t)r0 * bins * frames, stream);
      if (rc != MIA_OK) return rc;
    }
    return MIA_OK;
  }
  if (size_freq <= kGateMaxWin) {
    // tiles of neighbouring bins read each other's halo rows: xcd_tile() order keeps them behind
    // one L2
    MIA_REQUIRE((unsigned long long)((frames + kGateFT - 1) / kGateFT) * ((bins + kGateKT - 1) / kGateKT) < (1ULL << 28),
                "shape too large for one launch");
    const dim3 tgrid(xcd_tile_grid((unsigned)((frames + kGateFT - 1) / kGateFT),
                                   (unsigned)((bins + kGateKT - 1) / kGateKT)), 1, (unsigned)rows);
    hipLaunchKernelGGL(gate_mask_box_tiled_kernel, tgrid, dim3(256), 0, (hipStream_t)stream, mag, thr,
                       (unsigned)bins, (unsigned)frames, per_bin, (unsigned)rows_thr, gain, keep, p,
                       size_freq, out);
    MIA_CHECK_HIP(hipGetLastError());
    return MIA_OK;
  }
  const dim3 grid((unsigned)((frames + 255) / 256), (unsigned)bins, (unsigned)rows);
  hipLaunchKernelGGL(gate_mask_box_kernel, grid, dim3(256), 0, (hipStream_t)stream, mag, thr,
                     (unsigned)rows, (unsigned)bins, (unsigned)frames, per_bin, (unsigned)rows_thr,
                     gain, keep, p, size_freq, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_gate_apply(const float* mag, const float* mask, const float* phase, int64_t rows, int64_t bins,
                   int64_t frames, int keep, float p, int size_time, void* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 0 && frames >= 0 && size_time >= 1, "negative shape or size");
  if (rows == 0 || bins == 0 || frames == 0) return MIA_OK;
  MIA_REQUIRE(mag && mask && phase && out, "null pointer");
  MIA_REQUIRE((bins + 31) / 32 <= 65535, "too many bins for one launch");
  if (rows > 65535) {  // a launch takes 65535 rows: the rest follows in further launches
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
      const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
      const size_t off = (size_t)r0 * bins * frames;
      const int rc = mia_gate_apply(mag + off, mask + off, phase + off, nr, bins, frames, keep, p,
                                    size_time, (float2*)out + off, stream);
      if (rc != MIA_OK) return rc;
    }
    return MIA_OK;
  }
  const dim3 grid((unsigned)((frames + 31) / 32), (unsigned)((bins + 31) / 32), (unsigned)rows);
  hipLaunchKernelGGL(gate_apply_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, mag, mask, phase,
                     (long long)bins, (long long)frames, keep, p, size_time, (float2*)out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_gate_apply_spectrum(const void* spec, const float* mask, int64_t rows, int64_t bins,
                            int64_t frames, int keep, float p, int size_freq, int size_time,
                            void* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 0 && frames >= 0 && size_freq >= 1 && size_time >= 1,
              "negative shape or size");
  if (rows == 0 || bins == 0 || frames == 0) return MIA_OK;
  MIA_REQUIRE(spec && mask && out, "null pointer");
  MIA_REQUIRE((bins + 31) / 32 <= 65535, "too many bins for one launch");
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {  // a launch takes 65535 rows
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    const size_t off = (size_t)r0 * bins * frames;
    const dim3 grid((unsigned)((frames + 31) / 32), (unsigned)((bins + 31) / 32), (unsigned)nr);
    if (size_time <= kGateMaxWin) {
      // tiles of neighbouring bins share the cache lines their 256-byte runs of a spectrum row
      // straddle: xcd_tile() order keeps them behind one L2
      const unsigned tiles_f = (unsigned)((frames + kApplyTF - 1) / kApplyTF);
      const dim3 tgrid(xcd_tile_grid(tiles_f, grid.y), 1, grid.z);
      hipLaunchKernelGGL(gate_apply_spectrum_tiled_kernel, tgrid, dim3(32, 8), 0, (hipStream_t)stream,
                         (const float2*)spec + off, mask + off, (long long)bins, (long long)frames,
                         keep, p, size_freq, size_time, (float2*)out + off);
    } else {
      hipLaunchKernelGGL(gate_apply_spectrum_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream,
                         (const float2*)spec + off, mask + off, (long long)bins, (long long)frames,
                         keep, p, size_freq, size_time, (float2*)out + off);
    }
    MIA_CHECK_HIP(hipGetLastError());
  }
  return MIA_OK;
}

int mia_frame_energy_spec(const void* spec, int64_t rows, int bins, int64_t frames, float* energy,
                          void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 1 && frames >= 0, "rows, bins, frames must be >= 0, 1, 0");
  if (rows == 0 || frames == 0) return MIA_OK;
  MIA_REQUIRE(spec && energy, "null pointer");
  MIA_REQUIRE((rows * frames + 3) / 4 <= 0x7fffffffLL, "too many frames for one launch");
  hipLaunchKernelGGL(frame_energy_spec_kernel, dim3((unsigned)((rows * frames + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)spec, (long long)rows, bins, (long long)frames,
                     energy);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_noise_threshold_spec(const void* spec, int64_t rows, int bins, int64_t frames, const int* frame_idx,
                             int n_selected, float n_std, float* thr, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 1 && frames >= 1, "rows, bins, frames must be >= 0, 1, 1");
  MIA_REQUIRE(frame_idx == nullptr || n_selected >= 1, "n_selected must be >= 1");
  MIA_REQUIRE(frames <= 0x7fffffffLL, "too many frames");
  if (rows == 0) return MIA_OK;
  MIA_REQUIRE(spec && thr, "null pointer");
  const unsigned chunks = (unsigned)((bins + kThrBins - 1) / kThrBins);
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {  // a launch takes at most 65535 rows
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    hipLaunchKernelGGL(noise_threshold_spec_kernel, dim3(chunks, (unsigned)nr), dim3(256), 0,
                       (hipStream_t)stream, (const float2*)spec + (size_t)r0 * frames * bins, bins,
                       (long long)frames, frame_idx ? frame_idx + (size_t)r0 * n_selected : nullptr,
                       n_selected, n_std, thr + (size_t)r0 * bins);
    MIA_CHECK_HIP(hipGetLastError());
  }
  return MIA_OK;
}

int mia_quietest_frames(const float* energy, int64_t rows, int64_t frames, int n_selected, int* frame_idx,
                        void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && frames >= 1, "rows, frames must be >= 0, 1");
  MIA_REQUIRE(n_selected >= 1 && n_selected <= frames, "n_selected must be in [1, frames], got %d", n_selected);
  if (frames > kQuietMaxFrames) {
    set_error("rows of %lld frames: this kernel ranks at most %d (sort the energies instead)",
              (long long)frames, kQuietMaxFrames);
    return MIA_ERR_UNSUPPORTED;
  }
  if (rows == 0) return MIA_OK;
  MIA_REQUIRE(energy && frame_idx, "null pointer");
  const unsigned chunks = (unsigned)((frames + 63) / 64);
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {  // a launch takes at most 65535 rows
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    hipLaunchKernelGGL(quietest_frames_kernel, dim3(chunks, (unsigned)nr), dim3(256),
                       sizeof(int) * ((size_t)frames + 256), (hipStream_t)stream,
                       energy + (size_t)r0 * frames, (int)frames, n_selected,
                       frame_idx + (size_t)r0 * n_selected);
    MIA_CHECK_HIP(hipGetLastError());
  }
  return MIA_OK;
}

int mia_gate_mask_box_spec(const void* spec, const float* thr, int64_t rows, int bins, int64_t frames,
                           int64_t rows_thr, float gain, int keep, float p, int size_freq, float* out,
                           void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 1 && frames >= 0 && size_freq >= 1,
              "rows, bins, frames, size_freq must be >= 0, 1, 0, 1");
  MIA_REQUIRE(rows_thr == rows || rows_thr == 1,
              "the per-bin threshold must have one row per signal row, or a single row");
  if (size_freq > kGateMaxWin) {
    set_error("size_freq=%d: the spectrum form takes windows of up to %d bins (use mia_gate_mask_box "
              "on the magnitudes)", size_freq, kGateMaxWin);
    return MIA_ERR_UNSUPPORTED;
  }
  if (rows == 0 || frames == 0) return MIA_OK;
  MIA_REQUIRE(spec && thr && out, "null pointer");
  MIA_REQUIRE(frames <= 0x7fffffffLL, "shape too large for one launch");
  const unsigned tiles_f = (unsigned)((frames + kGateFT - 1) / kGateFT);
  const unsigned tiles_k = (unsigned)((bins + kGateKT - 1) / kGateKT);
  MIA_REQUIRE((unsigned long long)tiles_f * tiles_k < (1ULL << 28), "shape too large for one launch");
  const size_t per_row = (size_t)bins * (size_t)frames;
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {  // a launch takes 65535 rows
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    hipLaunchKernelGGL(gate_mask_box_spec_kernel, dim3(xcd_tile_grid(tiles_f, tiles_k), 1, (unsigned)nr),
                       dim3(256), 0, (hipStream_t)stream, (const float2*)spec + (size_t)r0 * per_row,
                       rows_thr == 1 ? thr : thr + (size_t)r0 * bins, (unsigned)bins, (unsigned)frames,
                       (unsigned)(rows_thr == 1 ? 1 : nr), gain, keep, p, size_freq,
                       out + (size_t)r0 * per_row);
  }
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_mask_op(const float* a, const float* b, int64_t n, int op, float p, float* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(n >= 0 && (op == 0 || op == 1), "n must be >= 0 and op 0 or 1");
  if (n == 0) return MIA_OK;
  MIA_REQUIRE(a && out && (op == 0 || b), "null pointer");
  hipLaunchKernelGGL(mask_op_kernel, dim3(gate_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b,
                     (long long)n, op, p, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // extern "C"

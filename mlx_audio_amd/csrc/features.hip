// Per-frame spectral statistics of a magnitude / power spectrogram laid out (rows, bins,
// frames) with frames contiguous, and the zero-crossing rate of framed signals.
//
// Statistics: one thread per frame; consecutive lanes hold consecutive frames, so every
// load of bin k is one coalesced row segment and each thread walks the bins in order with
// its own float32 accumulators (the summation order of a sequential reduction over the
// frequency axis).  The spectrogram is read once (for the roll-off frequency once for the total
// and then the few rows around the crossing; twice when all four statistics come together).  Zero crossings: one wave per frame, lanes striding the samples.
#include "mia_internal.h"

namespace mia {

constexpr float kDivEps = 1e-8f;  // the reference's DIVISION_EPSILON

__device__ __forceinline__ float pow_in(float v, int power_mode, float power) {
  if (power_mode == 0) return v;
  if (power_mode == 1) return v * v;
  return powf(v, power);
}

// Walks a column in order, eight rows fetched ahead of the arithmetic that consumes them.
template <typename Fn>
__device__ __forceinline__ void for_bins(const float* __restrict__ col, int bins, long long F, Fn&& fn) {
  int k = 0;
  for (; k + 8 <= bins; k += 8) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = col[(long long)(k + i) * F];
#pragma unroll
    for (int i = 0; i < 8; ++i) fn(k + i, v[i]);
  }
  for (; k < bins; ++k) fn(k, col[(long long)k * F]);
}

// mode 0: centroid  sum(f S) / (sum S + eps)
// mode 1: bandwidth (sum(S |f - c|^p) [/ (sum S + eps)])^(1/p), c = aux[row, frame]
// mode 2: roll-off  f[first k with cumsum_k >= fraction * total], last bin if none
//         (spectral_rolloff_kernel below; here only as part of mode 4)
// mode 3: flatness  exp(mean log max(S^power, amin)) / (mean max(S^power, amin) + eps)
__global__ __launch_bounds__(256) void spectral_stat_kernel(
    const float* __restrict__ S, long long rows, int bins, long long F,
    const float* __restrict__ freq, const float* __restrict__ aux, int mode, float p, int norm,
    float fraction, float amin, int power_mode, float power, float* __restrict__ out) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * F) return;
  const long long row = g / F, f = g - row * F;
  const float* __restrict__ col = S + row * bins * F + f;
  float result = 0.0f;
  if (mode == 0) {
    float ws = 0.0f, ts = 0.0f;
    for_bins(col, bins, F, [&](int k, float v) {
      ws += freq[k] * v;
      ts += v;
    });
    result = ws / (ts + kDivEps);
  } else if (mode == 1) {
    const float c = aux[g];
    float ws = 0.0f, ts = 0.0f;
    for_bins(col, bins, F, [&](int k, float v) {
      const float d = fabsf(freq[k] - c);
      const float dp = p == 2.0f ? d * d : (p == 1.0f ? d : powf(d, p));
      ws += v * dp;
      ts += v;
    });
    if (norm) ws = ws / (ts + kDivEps);
    result = p == 2.0f ? sqrtf(ws) : (p == 1.0f ? ws : powf(ws, 1.0f / p));
  } else if (mode == 4) {
    // all four in two passes: sums first, then the spread around the centroid and the
    // roll-off point; out is (rows, 4, frames) = centroid, bandwidth, roll-off, flatness, each
    // value what the single-statistic modes compute
    float ws = 0.0f, ts = 0.0f, ls = 0.0f, as = 0.0f;
    for_bins(col, bins, F, [&](int k, float v) {
      ws += freq[k] * v;
      ts += v;
      float u = pow_in(v, power_mode, power);
      u = u < amin ? amin : u;
      ls += logf(u);
      as += u;
    });
    const float c = ws / (ts + kDivEps);
    const float thr = fraction * ts;
    float bw = 0.0f, cs = 0.0f;
    int hit = bins - 1;
    bool open = true;
    for_bins(col, bins, F, [&](int k, float v) {
      const float d = fabsf(freq[k] - c);
      bw += v * (p == 2.0f ? d * d : (p == 1.0f ? d : powf(d, p)));
      cs += v;
      if (open && cs >= thr) {
        hit = k;
        open = false;
      }
    });
    if (norm) bw = bw / (ts + kDivEps);
    bw = p == 2.0f ? sqrtf(bw) : (p == 1.0f ? bw : powf(bw, 1.0f / p));
    float* __restrict__ o = out + row * 4 * F + f;
    o[0] = c;
    o[F] = bw;
    o[2 * F] = freq[hit];
    o[3 * F] = expf(ls / (float)bins) / (as / (float)bins + kDivEps);
    return;
  } else {
    float ls = 0.0f, as = 0.0f;
    for_bins(col, bins, F, [&](int, float s) {
      float v = pow_in(s, power_mode, power);
      v = v < amin ? amin : v;  // a NaN stays a NaN, as in maximum(S, amin)
      ls += logf(v);
      as += v;
    });
    result = expf(ls / (float)bins) / (as / (float)bins + kDivEps);
  }
  out[g] = result;
}

// Roll-off with one and a bit passes.  The first pass sums the column and leaves the running sum
// at the end of every block of `cb` bins in LDS (kRollBlocks checkpoints per thread, laid out
// [block][thread]).  With no negative (or NaN) value in the wave's columns the running sum never
// decreases, so the first bin whose running sum reaches the threshold lies in the first block whose
// checkpoint does; the second pass restarts from the checkpoint in front of that block and adds the
// block's bins in the same order - the sums it compares are bit for bit those of the full scan.
// The wave re-reads the rows from its lanes' lowest block to their highest one (whole coalesced
// rows; neighbouring frames cross at neighbouring bins).  A wave that saw a negative value or a NaN
// scans everything again from the first bin.
constexpr int kRollBlocks = 32;
__global__ __launch_bounds__(256) void spectral_rolloff_kernel(
    const float* __restrict__ S, long long rows, int bins, long long F, const float* __restrict__ freq,
    float fraction, float* __restrict__ out) {
  __shared__ float ck[kRollBlocks][256];
  const int tid = threadIdx.x;
  const long long g0 = (long long)blockIdx.x * blockDim.x + tid;
  const bool live = g0 < rows * F;
  const long long g = live ? g0 : rows * F - 1;  // idle lanes shadow the last frame, store nothing
  const long long row = g / F, f = g - row * F;
  const float* __restrict__ col = S + row * bins * F + f;
  // blocks of whole chunks of 8 bins: the loops below fetch a chunk's 8 rows, then add them in order
  const int cb = 8 * ((bins + 8 * kRollBlocks - 1) / (8 * kRollBlocks));
  const int nb = (bins + cb - 1) / cb;
  float total = 0.0f;
  bool mono = true;
  for (int b = 0; b < nb; ++b) {
    const int k1 = min((b + 1) * cb, bins);
    int k = b * cb;
    for (; k + 8 <= k1; k += 8) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = col[(long long)(k + i) * F];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        total += v[i];
        mono = mono && v[i] >= 0.0f;
      }
    }
    for (; k < k1; ++k) {
      const float v = col[(long long)k * F];
      total += v;
      mono = mono && v >= 0.0f;
    }
    ck[b][tid] = total;
  }
  const float thr = fraction * total;
  int hit = bins - 1;
  float cs = 0.0f;
  bool open = true;
  int lo = 0, hi = bins, my_lo = 0, my_hi = bins;
  if (__all(mono)) {
    int j = nb;
    for (int b = nb - 1; b >= 0; --b) j = ck[b][tid] >= thr ? b : j;
    // a lane without a crossing (j == nb) takes no part in the range and keeps the last bin
    my_lo = lo = j < nb ? j * cb : bins;
    my_hi = hi = j < nb ? min((j + 1) * cb, bins) : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo = min(lo, __shfl_xor(lo, off));
      hi = max(hi, __shfl_xor(hi, off));
    }
    if (j > 0 && j < nb) cs = ck[j - 1][tid];
  }
  auto step = [&](int k, float v) {
    if (k >= my_lo && k < my_hi) {
      cs += v;
      if (open && cs >= thr) {
        hit = k;
        open = false;
      }
    }
  };
  int k = lo;
  for (; k + 8 <= hi; k += 8) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = col[(long long)(k + i) * F];
#pragma unroll
    for (int i = 0; i < 8; ++i) step(k + i, v[i]);
  }
  for (; k < hi; ++k) step(k, col[(long long)k * F]);
  if (live) out[g0] = freq[hit];
}

// Spectral contrast: per frame and band, the mean of the nq smallest (valley) and of the nq
// largest (peak) values among bins lo .. hi-1, or the mean of all of them when nq >= hi - lo.
// band[k] = {lo, hi, nq}; an empty band (hi <= lo) yields peak = valley = 0.  One thread per
// frame keeps the K smallest and K largest values seen so far in two sorted register arrays and
// pushes each bin through both with K min/max pairs; K is chosen per band (the smallest of 2, 3,
// 4, 6, 8, 12 ... 64 that holds its nq: the narrow low bands select one or two values, only the
// top band needs a long chain), rows are fetched four ahead of the chains.
template <int K>
__device__ __forceinline__ void contrast_select(const float* __restrict__ col, long long F, int lo,
                                                int hi, int nq, float& peak, float& valley) {
  float small[K], large[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    small[i] = INFINITY;
    large[i] = -INFINITY;
  }
  auto push = [&](float v) {
    float a = v, c = v;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const float s0 = small[i], l0 = large[i];
      small[i] = fminf(s0, a);
      a = fmaxf(s0, a);
      large[i] = fmaxf(l0, c);
      c = fminf(l0, c);
    }
  };
  int k = lo;
  for (; k + 4 <= hi; k += 4) {
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = col[(long long)(k + i) * F];
#pragma unroll
    for (int i = 0; i < 4; ++i) push(v[i]);
  }
  for (; k < hi; ++k) push(col[(long long)k * F]);
  float vs = 0.0f, ps = 0.0f;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    if (i < nq) {
      vs += small[i];
      ps += large[i];
    }
  }
  valley = vs / (float)nq;
  peak = ps / (float)nq;
}

template <int KMAX>  // the chain length the widest quantile of the call needs (sets the register count)
__global__ __launch_bounds__(256) void spectral_contrast_kernel(
    const float* __restrict__ S, long long rows, int bins, long long F, const int* __restrict__ band,
    int n_out, int linear, float* __restrict__ out) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * F) return;
  const long long row = g / F, f = g - row * F;
  const float* __restrict__ col = S + row * bins * F + f;
  for (int b = 0; b < n_out; ++b) {
    const int lo = band[3 * b], hi = band[3 * b + 1], nq = band[3 * b + 2];
    float peak = 0.0f, valley = 0.0f;
    if (hi > lo) {
      if (nq >= hi - lo) {
        float sum = 0.0f;
        for (int k = lo; k < hi; ++k) sum += col[(long long)k * F];
        peak = valley = sum / (float)(hi - lo);
      } else {
        // the shortest chain of 2, 3, 4, 6, 8, 12, ... 64 that holds nq values
        bool done = false;
#define MIA_TRY(KK)                                                \
  if constexpr (KMAX >= KK) {                                      \
    if (!done && (nq <= KK || KMAX == KK)) {                       \
      contrast_select<KK>(col, F, lo, hi, nq, peak, valley);       \
      done = true;                                                 \
    }                                                              \
  }
        MIA_TRY(2) MIA_TRY(3) MIA_TRY(4) MIA_TRY(6) MIA_TRY(8) MIA_TRY(12) MIA_TRY(16) MIA_TRY(24)
        MIA_TRY(32) MIA_TRY(48) MIA_TRY(64)
#undef MIA_TRY
      }
    }
    float r;
    if (linear) {
      r = peak - valley;
    } else {
      r = 10.0f * log10f(fmaxf(peak, 1e-10f)) - 10.0f * log10f(fmaxf(valley, 1e-10f));
    }
    out[(row * n_out + b) * F + f] = r;
  }
}

// out[row, frame] = (number of i in 1..L-1 with sign(x[i]) != sign(x[i-1])) / L over the
// frame's L samples; sign = (x >= 0) or, with by_signbit, the IEEE sign bit.  Centre padding
// (pad samples on both sides) by zeros or by the edge sample.
__global__ __launch_bounds__(256) void zcr_kernel(const float* __restrict__ y, long long rows,
                                                  long long T, int L, int hop, long long pad,
                                                  int pad_edge, int by_signbit, long long F,
                                                  float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= rows * F) return;
  const long long row = wave / F, f = wave - row * F;
  const float* __restrict__ yr = y + row * T;
  const long long s0 = f * hop - pad;
  auto sample = [&](long long i) {
    if (i >= 0 && i < T) return yr[i];
    return pad_edge ? yr[i < 0 ? 0 : T - 1] : 0.0f;
  };
  auto positive = [&](float v) {
    return by_signbit ? (__float_as_uint(v) >> 31) == 0u : v >= 0.0f;
  };
  int count = 0;
  for (int i = 1 + lane; i < L; i += 64) {
    count += positive(sample(s0 + i)) != positive(sample(s0 + i - 1)) ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) count += __shfl_xor(count, off);
  if (lane == 0) out[wave] = (float)count / (float)L;
}

// The same counts from one read of the signal.  A workgroup takes a run of consecutive frames of
// one row; their samples form one span, cut into groups of 64.  A wave walks a contiguous quarter
// of the groups, eight loads in flight: per group the ballot of the signs, xor-ed with itself
// shifted by one sample (the carry is the last sign of the group before), is the mask of the
// crossings; the mask and the wave's running count go to LDS.  A frame's count is then
// P(last sample) - P(first sample), P(j) = crossings up to and including sample j of the span =
// running count at the end of j's group minus the mask bits above j.  Integer counts: the result
// equals zcr_kernel's bit for bit.
constexpr int kZcrMaxGroups = 4096;
__global__ __launch_bounds__(256) void zcr_span_kernel(const float* __restrict__ y, long long T, int L,
                                                       int hop, long long pad, int pad_edge,
                                                       int by_signbit, long long F, int frames_per_wg,
                                                       float* __restrict__ out) {
  extern __shared__ unsigned long long zcr_lds[];  // sized by the launch: 12 bytes per group
  __shared__ int wave_total[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row = blockIdx.y;
  const long long f0 = (long long)blockIdx.x * frames_per_wg;
  const int nf = (int)min((long long)frames_per_wg, F - f0);
  const long long j0 = f0 * hop;                       // span start, in padded coordinates
  const long long len = (long long)(nf - 1) * hop + L;  // samples of the span
  const int n_g = (int)((len + 63) / 64);
  const int gpw = (n_g + 3) / 4;                       // groups per wave
  unsigned long long* const mask = zcr_lds;
  // crossings of the wave's groups up to the end of this one; behind the masks of a full run
  int* const upto = reinterpret_cast<int*>(zcr_lds + ((long long)(frames_per_wg - 1) * hop + L + 63) / 64);
  const float* __restrict__ yr = y + row * T;
  // sign of padded sample j0 + j: the load is unconditional (index clamped into the row, which is
  // also what edge padding asks for), so that a batch of them can be in flight
  auto sample_at = [&](long long j) {
    const long long i = j0 + j - pad;
    const float v = yr[i < 0 ? 0 : (i >= T ? T - 1 : i)];
    return (pad_edge || (i >= 0 && i < T)) ? v : 0.0f;
  };
  auto positive = [&](float v) { return by_signbit ? (__float_as_uint(v) >> 31) == 0u : v >= 0.0f; };
  const int g_begin = wave * gpw, g_end = min(g_begin + gpw, n_g);
  // sign of the sample in front of the wave's first group (the span's first sample has none: its
  // bit is never counted)
  unsigned long long carry = 0;
  if (g_begin < g_end && g_begin > 0) carry = positive(sample_at((long long)g_begin * 64 - 1)) ? 1ull : 0ull;
  int run = 0;
  for (int g = g_begin; g < g_end; g += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = sample_at((long long)(g + u) * 64 + lane);  // past the span: masked below
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (g + u < g_end) {
        const unsigned long long b = __ballot(positive(v[u]));
        const long long left = len - (long long)(g + u) * 64;  // samples of the span in this group
        const unsigned long long valid = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
        const unsigned long long m = (b ^ ((b << 1) | carry)) & valid;
        carry = b >> 63;
        run += __popcll(m);
        if (lane == 0) {
          mask[g + u] = m;
          upto[g + u] = run;
        }
      }
    }
  }
  if (lane == 0) wave_total[wave] = run;
  __syncthreads();
  auto prefix = [&](long long j) {  // crossings at samples 0 .. j of the span (bit 0 of group 0 is garbage
    const int g = (int)(j >> 6);    // and cancels: every frame subtracts a prefix that contains it)
    const int w = g / gpw;
    int base = upto[g];
    for (int k = 0; k < w; ++k) base += wave_total[k];
    const int bit = (int)(j & 63);
    const unsigned long long above = bit == 63 ? 0ull : (mask[g] >> (bit + 1));
    return base - __popcll(above);
  };
  for (int f = threadIdx.x; f < nf; f += 256) {
    const long long s = (long long)f * hop;
    const int count = prefix(s + L - 1) - prefix(s);
    out[row * F + f0 + f] = (float)count / (float)L;
  }
}

}  // namespace mia

extern "C" {

int mia_spectral_stat(const float* S, int64_t rows, int bins, int64_t frames, const float* freq,
                      const float* aux, int mode, float p, int norm, float fraction, float amin,
                      float power, float* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 1 && frames >= 0, "rows, bins, frames must be >= 0, 1, 0");
  MIA_REQUIRE(mode >= MIA_STAT_CENTROID && mode <= MIA_STAT_ALL, "unknown statistic %d", mode);
  if (rows == 0 || frames == 0) return MIA_OK;
  MIA_REQUIRE(S && out, "null pointer");
  MIA_REQUIRE(mode == MIA_STAT_FLATNESS || freq, "this statistic needs the bin frequencies");
  MIA_REQUIRE(mode != MIA_STAT_ALL || p > 0.0f, "p must be > 0");
  MIA_REQUIRE(mode != MIA_STAT_BANDWIDTH || (aux && p > 0.0f),
              "the bandwidth needs the centroid of every frame and p > 0");
  const long long total = (long long)rows * frames;
  if (total > 0x7fffffffLL * 256) {
    set_error("%lld frames do not fit one launch (split the batch)", total);
    return MIA_ERR_UNSUPPORTED;
  }
  const int power_mode = power == 1.0f ? 0 : (power == 2.0f ? 1 : 2);
  if (mode == MIA_STAT_ROLLOFF) {
    hipLaunchKernelGGL(spectral_rolloff_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, S, (long long)rows, bins, (long long)frames, freq, fraction,
                       out);
    MIA_CHECK_HIP(hipGetLastError());
    return MIA_OK;
  }
  hipLaunchKernelGGL(spectral_stat_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, S, (long long)rows, bins, (long long)frames, freq, aux,
                     mode, p, norm, fraction, amin, power_mode, power, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_spectral_contrast(const float* S, int64_t rows, int bins, int64_t frames, const int* bands,
                          int n_bands_out, int max_quantile, int linear, float* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 1 && frames >= 0 && n_bands_out >= 1,
              "rows, bins, frames, n_bands_out must be >= 0, 1, 0, 1");
  MIA_REQUIRE(max_quantile >= 1, "max_quantile must be >= 1");
  if (rows == 0 || frames == 0) return MIA_OK;
  MIA_REQUIRE(S && bands && out, "null pointer");
  if (max_quantile > 64) {
    set_error("spectral contrast: %d values per quantile exceed the supported 64", max_quantile);
    return MIA_ERR_UNSUPPORTED;
  }
  const long long total = (long long)rows * frames;
  if (total > 0x7fffffffLL * 256) {
    set_error("%lld frames do not fit one launch (split the batch)", total);
    return MIA_ERR_UNSUPPORTED;
  }
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
#define MIA_CONTRAST(KK)                                                                        \
  hipLaunchKernelGGL((spectral_contrast_kernel<KK>), grid, dim3(256), 0, s, S, (long long)rows, \
                     bins, (long long)frames, bands, n_bands_out, linear, out)
  if (max_quantile <= 2) MIA_CONTRAST(2);
  else if (max_quantile <= 4) MIA_CONTRAST(4);
  else if (max_quantile <= 8) MIA_CONTRAST(8);
  else if (max_quantile <= 16) MIA_CONTRAST(16);
  else if (max_quantile <= 32) MIA_CONTRAST(32);
  else MIA_CONTRAST(64);
#undef MIA_CONTRAST
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_zero_crossing_rate(const float* y, int64_t rows, int64_t n, int frame_length, int hop_length,
                           int center, int pad_edge, int by_signbit, float* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && n >= 1, "rows and n must be >= 0 and 1");
  MIA_REQUIRE(frame_length >= 1 && hop_length >= 1, "frame_length and hop_length must be >= 1");
  const long long pad = center ? frame_length / 2 : 0;
  const long long padded = (long long)n + 2 * pad;
  MIA_REQUIRE(padded >= frame_length, "Input signal length (%lld) must be >= frame_length (%d)",
              padded, frame_length);
  const long long F = 1 + (padded - frame_length) / hop_length;
  if (rows == 0) return MIA_OK;
  MIA_REQUIRE(y && out, "null pointer");
  // one read of the signal where a run of frames fits the span kernel's tables
  {
    long long nf = 32768 / hop_length;
    if (nf < 1) nf = 1;
    if (nf > F) nf = F;
    const long long groups = ((nf - 1) * hop_length + frame_length + 63) / 64;
    if (groups <= kZcrMaxGroups && hop_length <= frame_length) {
      constexpr int64_t kRowsPerLaunch = 65535;  // grid.y
      for (int64_t r0 = 0; r0 < rows; r0 += kRowsPerLaunch) {
        const int64_t nr = rows - r0 < kRowsPerLaunch ? rows - r0 : kRowsPerLaunch;
        hipLaunchKernelGGL(zcr_span_kernel, dim3((unsigned)((F + nf - 1) / nf), (unsigned)nr), dim3(256),
                           (size_t)groups * 12, (hipStream_t)stream, y + (size_t)r0 * n, (long long)n, frame_length, hop_length,
                           pad, pad_edge, by_signbit, F, (int)nf, out + (size_t)r0 * F);
        MIA_CHECK_HIP(hipGetLastError());
      }
      return MIA_OK;
    }
  }
  const long long waves = (long long)rows * F;
  if (waves > 0x7fffffffLL * 4) {
    set_error("%lld frames do not fit one launch (split the batch)", waves);
    return MIA_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(zcr_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, y, (long long)rows, (long long)n, frame_length,
                     hop_length, pad, pad_edge, by_signbit, F, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // extern "C"

// Onset strength: spectral flux of a log-compressed mel spectrogram, and the two small 1-D
// filters (local maximum, running mean) applied to the envelope.
//
// Flux: S is (rows, bands, frames), frames contiguous.  One thread per frame t >= lag walks
// the bands in order: d = max(log1p(1e4 S[m,t]) - log1p(1e4 S[m,t-lag]), 0), aggregated over
// the bands as mean, max or median.  Consecutive lanes hold consecutive frames, so the two
// loads of a band are coalesced row segments (the lagged one hits the cache).  The median
// keeps the thread's column of differences in LDS and finds the two middle order statistics by
// counting, O(bands^2) comparisons per frame.
#include "mia_internal.h"

namespace mia {

// log(1 + x) for the compression step.  x = 1e4 * S >= 0, and what is needed is an absolute
// accuracy: rounding 1 + x costs at most 6e-8 in the logarithm, logf another ulp of a value
// below ~16 (1e-6).  log1pf's extra care for tiny x buys nothing here and costs 3x the whole
// kernel (1.07 ms against 0.355 ms on 4096 spectrograms of 128 x 938).
__device__ __forceinline__ float log_compress(float x) { return logf(1.0f + x); }

constexpr int kOnsetWG = 64;  // frames per workgroup (one wave): bounds the median's LDS tile

__global__ __launch_bounds__(kOnsetWG) void onset_flux_kernel(const float* __restrict__ S,
                                                              long long rows, int bands,
                                                              long long F, int lag, int aggregate,
                                                              float* __restrict__ out) {
  extern __shared__ float onset_tile[];  // [bands][kOnsetWG], median only
  const long long g = (long long)blockIdx.x * kOnsetWG + threadIdx.x;
  const bool valid = g < rows * F;
  const long long row = valid ? g / F : 0, t = valid ? g - row * F : 0;
  const bool live = valid && t >= lag;
  const float* __restrict__ col = S + row * bands * F + t;
  float result = 0.0f;
  if (aggregate != MIA_AGG_MEDIAN) {
    if (live) {
      float acc = aggregate == MIA_AGG_MAX ? -INFINITY : 0.0f;
#pragma unroll 4
      for (int m = 0; m < bands; ++m) {
        const float a = log_compress(col[(long long)m * F] * 1e4f);
        const float b = log_compress(col[(long long)m * F - lag] * 1e4f);
        const float d = fmaxf(a - b, 0.0f);
        acc = aggregate == MIA_AGG_MAX ? fmaxf(acc, d) : acc + d;
      }
      result = aggregate == MIA_AGG_MAX ? acc : acc / (float)bands;
    }
  } else {
    float* mine = onset_tile + threadIdx.x;
    if (live) {
      for (int m = 0; m < bands; ++m) {
        const float a = log_compress(col[(long long)m * F] * 1e4f);
        const float b = log_compress(col[(long long)m * F - lag] * 1e4f);
        mine[m * kOnsetWG] = fmaxf(a - b, 0.0f);
      }
      // order statistics k1 = (bands-1)/2 and k2 = bands/2 (equal for an odd count)
      const int k1 = (bands - 1) / 2, k2 = bands / 2;
      float v1 = 0.0f, v2 = 0.0f;
      for (int i = 0; i < bands; ++i) {
        const float v = mine[i * kOnsetWG];
        int less = 0, equal = 0;
        for (int j = 0; j < bands; ++j) {
          const float u = mine[j * kOnsetWG];
          less += u < v ? 1 : 0;
          equal += u == v ? 1 : 0;
        }
        if (less <= k1 && k1 < less + equal) v1 = v;
        if (less <= k2 && k2 < less + equal) v2 = v;
      }
      result = k1 == k2 ? v1 : (v1 + v2) * 0.5f;
    }
  }
  if (valid) out[g] = result;
}

// Mean / max aggregation with each log1p computed once: a wave covers 64 consecutive frames of a
// row, of which the first `lag` only supply the lagged values (through a lane shift) and the
// other 64 - lag produce outputs.  Needs lag <= 32.
__global__ __launch_bounds__(256) void onset_flux_shift_kernel(const float* __restrict__ S,
                                                               long long rows, int bands,
                                                               long long F, int lag, int aggregate,
                                                               long long waves_per_row,
                                                               float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= rows * waves_per_row) return;  // whole waves leave together
  const long long row = wave / waves_per_row, w = wave - row * waves_per_row;
  const int per = 64 - lag;
  const long long t = w * per + lane;
  const bool inside = t < F;
  const float* __restrict__ col = S + row * bands * F + (inside ? t : F - 1);
  float acc = aggregate == MIA_AGG_MAX ? -INFINITY : 0.0f;
  auto step = [&](float v) {
    const float a = log_compress(v * 1e4f);
    const float b = __shfl_up(a, lag);
    const float d = fmaxf(a - b, 0.0f);
    acc = aggregate == MIA_AGG_MAX ? fmaxf(acc, d) : acc + d;
  };
  int m = 0;
  for (; m + 8 <= bands; m += 8) {  // eight loads in flight (a lane shift blocks automatic unrolling)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = col[(long long)(m + u) * F];
#pragma unroll
    for (int u = 0; u < 8; ++u) step(v[u]);
  }
  for (; m < bands; ++m) step(col[(long long)m * F]);
  if (!inside) return;
  if (lane >= lag) {
    out[row * F + t] = aggregate == MIA_AGG_MAX ? acc : acc / (float)bands;
  } else if (w == 0) {
    out[row * F + t] = 0.0f;  // the first `lag` frames of the row
  }
}

// dst[r, i] over the window [i - size/2, i + (size-1)/2] of src[r, :n], indices reflected at
// the ends with the edge sample repeated (d c b a | a b c d | d c b a):
//   mode 0: the maximum;  mode 1: max(src[i] - mean, 0), the mean accumulated in double.
__global__ __launch_bounds__(256) void envelope_filter_kernel(const float* __restrict__ src,
                                                              long long rows, long long n,
                                                              long long src_stride, int size,
                                                              int mode, float* __restrict__ dst,
                                                              long long dst_stride) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * n) return;
  const long long r = g / n, i = g - r * n;
  const float* __restrict__ x = src + r * src_stride;
  const long long lo = i - size / 2, hi = i + (size - 1) / 2;
  float mx = -INFINITY;
  double sum = 0.0;
  for (long long j = lo; j <= hi; ++j) {
    long long q = j % (2 * n);
    if (q < 0) q += 2 * n;
    if (q >= n) q = 2 * n - 1 - q;
    const float v = x[q];
    mx = fmaxf(mx, v);
    sum += (double)v;
  }
  float o;
  if (mode == 0) {
    o = mx;
  } else {
    const float mean = (float)(sum / (double)size);
    o = fmaxf(x[i] - mean, 0.0f);
  }
  dst[r * dst_stride + i] = o;
}

}  // namespace mia

extern "C" {

int mia_onset_flux(const float* S, int64_t rows, int bands, int64_t frames, int lag, int aggregate,
                   float* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bands >= 1 && frames >= 0, "rows, bands, frames must be >= 0, 1, 0");
  MIA_REQUIRE(lag >= 1, "lag must be positive, got %d", lag);
  MIA_REQUIRE(aggregate >= MIA_AGG_MEAN && aggregate <= MIA_AGG_MEDIAN, "unknown aggregate %d",
              aggregate);
  if (rows == 0 || frames == 0) return MIA_OK;
  MIA_REQUIRE(S && out, "null pointer");
  size_t lds = 0;
  if (aggregate == MIA_AGG_MEDIAN) {
    lds = sizeof(float) * (size_t)bands * kOnsetWG;
    if (lds > 64 * 1024) {
      set_error("median over %d bands needs %zu bytes of LDS (limit 65536)", bands, lds);
      return MIA_ERR_UNSUPPORTED;
    }
  }
  if (aggregate != MIA_AGG_MEDIAN && lag <= 32) {
    const int per = 64 - lag;
    const long long body = frames > lag ? frames - lag : 0;
    long long wpr = (body + per - 1) / per;
    if (wpr < 1) wpr = 1;  // at least the wave that writes the leading zeros
    const long long waves = (long long)rows * wpr;
    if (waves > 0x7fffffffLL * 4) {
      set_error("%lld frames do not fit one launch (split the batch)", (long long)rows * frames);
      return MIA_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(onset_flux_shift_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, S, (long long)rows, bands, (long long)frames, lag,
                       aggregate, wpr, out);
    MIA_CHECK_HIP(hipGetLastError());
    return MIA_OK;
  }
  const long long total = (long long)rows * frames;
  if (total > 0x7fffffffLL * kOnsetWG) {
    set_error("%lld frames do not fit one launch (split the batch)", total);
    return MIA_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(onset_flux_kernel, dim3((unsigned)((total + kOnsetWG - 1) / kOnsetWG)),
                     dim3(kOnsetWG), lds, (hipStream_t)stream, S, (long long)rows, bands,
                     (long long)frames, lag, aggregate, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_envelope_filter(const float* src, int64_t rows, int64_t n, int64_t src_stride, int size,
                        int mode, float* dst, int64_t dst_stride, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && n >= 0 && size >= 1, "rows, n, size must be >= 0, 0, 1");
  MIA_REQUIRE(mode == 0 || mode == 1, "unknown filter mode %d", mode);
  MIA_REQUIRE(src_stride >= n && dst_stride >= n, "row strides must be >= n");
  if (rows == 0 || n == 0) return MIA_OK;
  MIA_REQUIRE(src && dst && src != dst, "null or aliased pointers");
  const long long total = (long long)rows * n;
  hipLaunchKernelGGL(envelope_filter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, (long long)rows, (long long)n,
                     (long long)src_stride, size, mode, dst, (long long)dst_stride);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // extern "C"

// Input normalisation of the HTDemucs front-end: per batch item, mean and standard deviation
// over everything that item holds (all channels, bins and frames of the spectrogram taken as
// real and imaginary planes; all channels and samples of the waveform), then (x - mean) /
// (std + eps).  Sums run in double (per-thread partials, a workgroup reduction, one atomic
// pair per workgroup); the spectrogram is read through its strides, as the strided view of the
// STFT kernel's (rows, frames, bins) output that the front-end hands over, and written as
// (batch, 2*channels, bins, frames) planes through an LDS tile so that both sides are coalesced.
// The spectrogram's sums may also come from the STFT kernel itself (mia_stft_moments), which
// saves the pass that adds them up here.
#include "mia_internal.h"

namespace mia {

__device__ __forceinline__ void block_accumulate(double s, double q, double* __restrict__ dst) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off);
    q += __shfl_xor(q, off);
  }
  __shared__ double red[2][4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wave] = s;
    red[1][wave] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(dst, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(dst + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// sums[row] += {sum x, sum x^2} over a contiguous row of n floats; grid (chunks, rows)
__global__ __launch_bounds__(256) void row_moments_kernel(const float* __restrict__ x, long long n,
                                                          double* __restrict__ sums) {
  const long long row = blockIdx.y;
  const float* __restrict__ line = x + row * n;
  // four independent partial sums per thread: four loads in flight instead of a chain of one
  double s4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double v = (double)line[i + k * stride];
      s4[k] += v;
      q4[k] += v * v;
    }
  }
  for (; i < n; i += stride) {
    const double v = (double)line[i];
    s4[0] += v;
    q4[0] += v * v;
  }
  block_accumulate((s4[0] + s4[1]) + (s4[2] + s4[3]), (q4[0] + q4[1]) + (q4[2] + q4[3]), sums + 2 * row);
}

// the same over the complex elements spec[b, c, f, t] (strides in complex elements), both
// parts; grid (chunks, batch * channels): a workgroup takes frames chunk, chunk + chunks, ... of
// one plane, its threads stride the bins (no index division in the loop)
__global__ __launch_bounds__(256) void cac_moments_kernel(const float2* __restrict__ spec, int C,
                                                          long long F, long long T, long long sb,
                                                          long long sc, long long sf, long long st,
                                                          double* __restrict__ sums) {
  const long long bc = blockIdx.y, b = bc / C, c = bc - b * C;
  const float2* __restrict__ plane = spec + b * sb + c * sc;
  double s = 0.0, q = 0.0;
  for (long long t = blockIdx.x; t < T; t += gridDim.x) {
    const float2* __restrict__ line = plane + t * st;
    for (long long f = threadIdx.x; f < F; f += 256) {
      const float2 z = line[f * sf];
      s += (double)z.x + (double)z.y;
      q += (double)z.x * (double)z.x + (double)z.y * (double)z.y;
    }
  }
  block_accumulate(s, q, sums + 2 * b);
}

// out[row, :] = zero_left zeros, x[row, :], zero_right zeros (what HTDemucs pads its waveform with
// before the STFT), and on the way the sum and sum of squares of x over each group of `group`
// consecutive rows (a batch item's channels) added to sums[2 * (row / group)] - the moments the
// waveform's normalisation needs, from the one pass that has to read the waveform anyway.
// grid (chunks, rows).
__global__ __launch_bounds__(256) void pad_rows_moments_kernel(const float* __restrict__ x, long long n,
                                                               long long zero_left, long long zero_right,
                                                               int group, float* __restrict__ out,
                                                               double* __restrict__ sums) {
  const long long row = blockIdx.y;
  const long long np = zero_left + n + zero_right;
  const float* __restrict__ line = x + row * n;
  float* __restrict__ dst = out + row * np;
  const long long stride = (long long)gridDim.x * 256;
  const long long start = (long long)blockIdx.x * 256 + threadIdx.x;
  for (long long i = start; i < zero_left; i += stride) dst[i] = 0.0f;
  for (long long i = start; i < zero_right; i += stride) dst[zero_left + n + i] = 0.0f;
  double s4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
  long long i = start;
  for (; i + 3 * stride < n; i += 4 * stride) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = line[i + k * stride];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dst[zero_left + i + k * stride] = v[k];
      s4[k] += (double)v[k];
      q4[k] += (double)v[k] * (double)v[k];
    }
  }
  for (; i < n; i += stride) {
    const float v = line[i];
    dst[zero_left + i] = v;
    s4[0] += (double)v;
    q4[0] += (double)v * (double)v;
  }
  block_accumulate((s4[0] + s4[1]) + (s4[2] + s4[3]), (q4[0] + q4[1]) + (q4[2] + q4[3]),
                   sums + 2 * (row / group));
}

__device__ __forceinline__ void mean_std(const double* __restrict__ sums, double count, float eps,
                                         float& mean, float& inv) {
  const double m = sums[0] / count;
  double var = sums[1] / count - m * m;
  if (var < 0.0) var = 0.0;
  mean = (float)m;
  inv = 1.0f / ((float)sqrt(var) + eps);
}

__global__ __launch_bounds__(256) void row_normalize_kernel(const float* __restrict__ x, long long n,
                                                            const double* __restrict__ sums,
                                                            float eps, float* __restrict__ out,
                                                            float* __restrict__ stats) {
  const long long row = blockIdx.y;
  float mean, inv;
  mean_std(sums + 2 * row, (double)n, eps, mean, inv);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = 1.0f / inv;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    out[row * n + i] = (x[row * n + i] - mean) * inv;
  }
}

// out[b, 2c + part, f, t] = (part of spec[b, c, f, t] - mean_b) / (std_b + eps); tiles of 32
// bins x 64 frames go through LDS: read with f fastest (256-byte runs of complex values), written
// with t fastest (256-byte runs of each plane).  grid (tiles_t, tiles_f, B*C), block (32, 8).
// Mean and 1 / (std + eps) come from one thread per workgroup (a double-precision division and
// square root per thread would cost more than the tile's arithmetic).
constexpr int kCacTT = 64;
__global__ __launch_bounds__(256) void cac_normalize_kernel(const float2* __restrict__ spec, int C,
                                                            long long F, long long T, long long sb,
                                                            long long sc, long long sf, long long st,
                                                            const double* __restrict__ sums,
                                                            float eps, float* __restrict__ out,
                                                            float* __restrict__ stats) {
  __shared__ float2 tile[kCacTT][33];
  __shared__ float mean_inv[2];
  unsigned tile_t, tile_f;
  if (!xcd_tile((unsigned)((T + kCacTT - 1) / kCacTT), (unsigned)((F + 31) / 32), tile_t, tile_f)) return;
  const long long bc = blockIdx.z, b = bc / C, c = bc - b * C;
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    float mean, inv;
    mean_std(sums + 2 * b, 2.0 * (double)C * (double)F * (double)T, eps, mean, inv);
    mean_inv[0] = mean;
    mean_inv[1] = inv;
    if (c == 0 && tile_t == 0 && tile_f == 0) {
      stats[2 * b] = mean;
      stats[2 * b + 1] = 1.0f / inv;
    }
  }
  const long long f0 = (long long)tile_f * 32, t0 = (long long)tile_t * kCacTT;
  const float2* __restrict__ src = spec + b * sb + c * sc;
#pragma unroll
  for (int j = threadIdx.y; j < kCacTT; j += 8) {  // j: frame within the tile, x: bin
    const long long f = f0 + threadIdx.x, t = t0 + j;
    float2 z = {0.0f, 0.0f};
    if (f < F && t < T) z = src[f * sf + t * st];
    tile[j][threadIdx.x] = z;
  }
  __syncthreads();
  const float mean = mean_inv[0], inv = mean_inv[1];
  float* __restrict__ re = out + ((b * 2 * C + 2 * c) * F) * T;
  float* __restrict__ im = re + F * T;
  const int tx = threadIdx.y % 2 * 32 + threadIdx.x;  // frame within the tile: 64 per bin row
#pragma unroll
  for (int j = threadIdx.y / 2; j < 32; j += 4) {  // j: bin within the tile
    const long long f = f0 + j, t = t0 + tx;
    if (f < F && t < T) {
      const float2 z = tile[tx][j];
      re[f * T + t] = (z.x - mean) * inv;
      im[f * T + t] = (z.y - mean) * inv;
    }
  }
}

// The same two steps for a spectrogram held with frames fastest ([batch, channels, bins, frames]
// as the reference holds it, or a slice of such an array: stride_t == 1, any stride_f): input and
// output then share their element order, so a workgroup walks rows of `frames` elements straight
// through (the tiled kernel above would read such an array through a stride of stride_f elements).
__global__ __launch_bounds__(256) void cac_moments_rows_kernel(const float2* __restrict__ spec, int C,
                                                               long long F, long long T, long long sb,
                                                               long long sc, long long sf,
                                                               double* __restrict__ sums) {
  const long long bc = blockIdx.y, b = bc / C, c = bc - b * C;
  const float2* __restrict__ plane = spec + b * sb + c * sc;
  double s2[2] = {0.0, 0.0}, q2[2] = {0.0, 0.0};
  for (long long f = blockIdx.x; f < F; f += gridDim.x) {
    const float2* __restrict__ line = plane + f * sf;
    long long t = threadIdx.x;
    for (; t + 256 < T; t += 512) {  // two loads in flight per thread
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float2 z = line[t + 256 * k];
        s2[k] += (double)z.x + (double)z.y;
        q2[k] += (double)z.x * (double)z.x + (double)z.y * (double)z.y;
      }
    }
    if (t < T) {
      const float2 z = line[t];
      s2[0] += (double)z.x + (double)z.y;
      q2[0] += (double)z.x * (double)z.x + (double)z.y * (double)z.y;
    }
  }
  block_accumulate(s2[0] + s2[1], q2[0] + q2[1], sums + 2 * b);
}

__global__ __launch_bounds__(256) void cac_normalize_rows_kernel(const float2* __restrict__ spec, int C,
                                                                 long long F, long long T, long long sb,
                                                                 long long sc, long long sf,
                                                                 const double* __restrict__ sums,
                                                                 float eps, float* __restrict__ out,
                                                                 float* __restrict__ stats) {
  __shared__ float mean_inv[2];
  const long long bc = blockIdx.y, b = bc / C, c = bc - b * C;
  if (threadIdx.x == 0) {
    float mean, inv;
    mean_std(sums + 2 * b, 2.0 * (double)C * (double)F * (double)T, eps, mean, inv);
    mean_inv[0] = mean;
    mean_inv[1] = inv;
    if (c == 0 && blockIdx.x == 0) {
      stats[2 * b] = mean;
      stats[2 * b + 1] = 1.0f / inv;
    }
  }
  __syncthreads();
  const float mean = mean_inv[0], inv = mean_inv[1];
  const float2* __restrict__ plane = spec + b * sb + c * sc;
  float* __restrict__ re = out + ((b * 2 * C + 2 * c) * F) * T;
  float* __restrict__ im = re + F * T;
  for (long long f = blockIdx.x; f < F; f += gridDim.x) {
    const float2* __restrict__ line = plane + f * sf;
    for (long long t = threadIdx.x; t < T; t += 256) {
      const float2 z = line[t];
      re[f * T + t] = (z.x - mean) * inv;
      im[f * T + t] = (z.y - mean) * inv;
    }
  }
}

// The way back (what HTDemucs does before its inverse STFT): real and imaginary planes
// planes[row, part, f, t] (t contiguous) become the complex spectrum out[row, t + pad_t, f] in the
// (rows, frames, bins) order the inverse STFT kernel reads with coalesced rows, with pad_k zero
// bins appended (the dropped Nyquist bin) and pad_t zero frames on either side, all written by
// this one kernel.  Tiles of 32 bins x 64 frames of the OUTPUT through LDS, in xcd_tile() order;
// 256 threads.
__global__ __launch_bounds__(256) void cac_to_spectrum_kernel(const float* __restrict__ planes,
                                                              long long F, long long T, int pad_k,
                                                              int pad_t, float2* __restrict__ out) {
  __shared__ float2 tile[kCacTT][33];
  const long long Fo = F + pad_k, To = T + 2LL * pad_t;
  unsigned tile_t, tile_f;
  if (!xcd_tile((unsigned)((To + kCacTT - 1) / kCacTT), (unsigned)((Fo + 31) / 32), tile_t, tile_f)) return;
  const long long row = blockIdx.z;
  const long long f0 = (long long)tile_f * 32, t0 = (long long)tile_t * kCacTT;
  const float* __restrict__ re = planes + row * 2 * F * T;
  const float* __restrict__ im = re + F * T;
  const int tx = threadIdx.y % 2 * 32 + threadIdx.x;  // frame within the tile: 64 per bin row
#pragma unroll
  for (int j = threadIdx.y / 2; j < 32; j += 4) {  // j: bin within the tile
    const long long f = f0 + j, t = t0 + tx - pad_t;
    float2 z = {0.0f, 0.0f};
    if (f < F && t >= 0 && t < T) z = {re[f * T + t], im[f * T + t]};
    tile[tx][j] = z;
  }
  __syncthreads();
  float2* __restrict__ dst = out + row * To * Fo;
#pragma unroll
  for (int j = threadIdx.y; j < kCacTT; j += 8) {  // j: frame within the tile, x: bin
    const long long f = f0 + threadIdx.x, t = t0 + j;
    if (f < Fo && t < To) dst[t * Fo + f] = tile[j][threadIdx.x];
  }
}

}  // namespace mia

extern "C" {

static int normalize_rows(const float* x, int64_t rows, int64_t n, float eps, float* out, float* stats,
                          void* sums, int have_sums, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && n >= 1, "rows and n must be >= 0 and 1");
  if (rows == 0) return MIA_OK;
  MIA_REQUIRE(x && out && stats && sums, "null pointer");
  if (rows > 65535) {  // a launch takes 65535 rows: the rest follows in further launches
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
      const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
      const int rc = normalize_rows(x + (size_t)r0 * n, nr, n, eps, out + (size_t)r0 * n,
                                    stats + 2 * (size_t)r0, (double*)sums + 2 * (size_t)r0, have_sums, stream);
      if (rc != MIA_OK) return rc;
    }
    return MIA_OK;
  }
  hipStream_t s = (hipStream_t)stream;
  if (!have_sums) MIA_CHECK_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)rows, s));
  // about 2048 workgroups in all: enough to fill the device, few enough that the two atomics
  // each of them ends with do not queue up on a row's pair of sums
  long long chunks = (n + 256 * 16 - 1) / (256 * 16);
  const long long cap = 2048 / rows > 1 ? 2048 / rows : 1;
  if (chunks > cap) chunks = cap;
  const dim3 grid((unsigned)chunks, (unsigned)rows);
  if (!have_sums) {
    hipLaunchKernelGGL(row_moments_kernel, grid, dim3(256), 0, s, x, (long long)n, (double*)sums);
  }
  hipLaunchKernelGGL(row_normalize_kernel, grid, dim3(256), 0, s, x, (long long)n,
                     (const double*)sums, eps, out, stats);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

static int normalize_cac(const void* spec, int64_t batch, int channels, int64_t bins, int64_t frames,
                         int64_t stride_b, int64_t stride_c, int64_t stride_f, int64_t stride_t,
                         float eps, float* out, float* stats, void* sums, int have_sums, void* stream) {
  using namespace mia;
  MIA_REQUIRE(batch >= 0 && channels >= 1 && bins >= 1 && frames >= 1,
              "batch, channels, bins, frames must be >= 0, 1, 1, 1");
  MIA_REQUIRE(channels <= 65535, "at most 65535 channels");
  if (batch == 0) return MIA_OK;
  MIA_REQUIRE(spec && out && stats && sums, "null pointer");
  if (batch * channels > 65535) {  // a launch takes 65535 planes: batch items go through in groups
    const int64_t per = 65535 / channels;
    for (int64_t b0 = 0; b0 < batch; b0 += per) {
      const int64_t nb = batch - b0 < per ? batch - b0 : per;
      const int rc = normalize_cac((const float2*)spec + (size_t)b0 * stride_b, nb, channels, bins,
                                   frames, stride_b, stride_c, stride_f, stride_t, eps,
                                   out + (size_t)b0 * 2 * channels * bins * frames,
                                   stats + 2 * (size_t)b0, (double*)sums + 2 * (size_t)b0, have_sums,
                                   stream);
      if (rc != MIA_OK) return rc;
    }
    return MIA_OK;
  }
  hipStream_t s = (hipStream_t)stream;
  if (stride_t == 1) {  // frames fastest: rows walked straight through, no transposition
    long long chunks = bins;
    const long long cap = 8192 / (batch * channels) > 1 ? 8192 / (batch * channels) : 1;
    if (chunks > cap) chunks = cap;
    const dim3 rgrid((unsigned)chunks, (unsigned)(batch * channels));
    if (!have_sums) {
      MIA_CHECK_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)batch, s));
      hipLaunchKernelGGL(cac_moments_rows_kernel, rgrid, dim3(256), 0, s, (const float2*)spec, channels,
                         (long long)bins, (long long)frames, (long long)stride_b, (long long)stride_c,
                         (long long)stride_f, (double*)sums);
    }
    hipLaunchKernelGGL(cac_normalize_rows_kernel, rgrid, dim3(256), 0, s, (const float2*)spec, channels,
                       (long long)bins, (long long)frames, (long long)stride_b, (long long)stride_c,
                       (long long)stride_f, (const double*)sums, eps, out, stats);
    MIA_CHECK_HIP(hipGetLastError());
    return MIA_OK;
  }
  if (!have_sums) {
    MIA_CHECK_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)batch, s));
    long long chunks = frames;
    const long long cap = 2048 / (batch * channels) > 1 ? 2048 / (batch * channels) : 1;
    if (chunks > cap) chunks = cap;
    hipLaunchKernelGGL(cac_moments_kernel, dim3((unsigned)chunks, (unsigned)(batch * channels)),
                       dim3(256), 0, s,
                       (const float2*)spec, channels, (long long)bins, (long long)frames,
                       (long long)stride_b, (long long)stride_c, (long long)stride_f,
                       (long long)stride_t, (double*)sums);
  }
  // tiles above and below one another share the cache lines their 256-byte runs of a spectrum
  // row straddle: xcd_tile() order keeps them behind one L2 (FETCH_SIZE 1.46 x -> 1.00 x the
  // spectrum)
  const unsigned tiles_t = (unsigned)((frames + kCacTT - 1) / kCacTT), tiles_f = (unsigned)((bins + 31) / 32);
  MIA_REQUIRE((unsigned long long)tiles_t * tiles_f < (1ULL << 28), "spectrogram too large for one launch");
  const dim3 grid(xcd_tile_grid(tiles_t, tiles_f), 1, (unsigned)(batch * channels));
  hipLaunchKernelGGL(cac_normalize_kernel, grid, dim3(32, 8), 0, s, (const float2*)spec, channels,
                     (long long)bins, (long long)frames, (long long)stride_b, (long long)stride_c,
                     (long long)stride_f, (long long)stride_t, (const double*)sums, eps, out, stats);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_normalize_cac(const void* spec, int64_t batch, int channels, int64_t bins, int64_t frames,
                      int64_t stride_b, int64_t stride_c, int64_t stride_f, int64_t stride_t,
                      float eps, float* out, float* stats, void* sums, void* stream) {
  return normalize_cac(spec, batch, channels, bins, frames, stride_b, stride_c, stride_f, stride_t, eps,
                       out, stats, sums, 0, stream);
}

int mia_normalize_cac_sums(const void* spec, int64_t batch, int channels, int64_t bins, int64_t frames,
                           int64_t stride_b, int64_t stride_c, int64_t stride_f, int64_t stride_t,
                           float eps, float* out, float* stats, const void* sums, void* stream) {
  return normalize_cac(spec, batch, channels, bins, frames, stride_b, stride_c, stride_f, stride_t, eps,
                       out, stats, const_cast<void*>(sums), 1, stream);
}

int mia_cac_to_spectrum(const float* planes, int64_t rows, int64_t bins, int64_t frames, int pad_bins,
                        int pad_frames, void* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 1 && frames >= 1 && pad_bins >= 0 && pad_frames >= 0,
              "rows, bins, frames, pad_bins, pad_frames must be >= 0, 1, 1, 0, 0");
  if (rows == 0) return MIA_OK;
  MIA_REQUIRE(planes && out, "null pointer");
  const unsigned tiles_t = (unsigned)((frames + 2LL * pad_frames + kCacTT - 1) / kCacTT);
  const unsigned tiles_f = (unsigned)((bins + pad_bins + 31) / 32);
  MIA_REQUIRE((unsigned long long)tiles_t * tiles_f < (1ULL << 28), "spectrogram too large for one launch");
  const size_t in_row = 2 * (size_t)bins * (size_t)frames;
  const size_t out_row = (size_t)(bins + pad_bins) * (size_t)(frames + 2LL * pad_frames);
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {  // a launch takes 65535 rows
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    hipLaunchKernelGGL(cac_to_spectrum_kernel, dim3(xcd_tile_grid(tiles_t, tiles_f), 1, (unsigned)nr),
                       dim3(32, 8), 0, (hipStream_t)stream, planes + (size_t)r0 * in_row, (long long)bins,
                       (long long)frames, pad_bins, pad_frames, (float2*)out + (size_t)r0 * out_row);
  }
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_normalize_rows(const float* x, int64_t rows, int64_t n, float eps, float* out, float* stats,
                       void* sums, void* stream) {
  return normalize_rows(x, rows, n, eps, out, stats, sums, 0, stream);
}

int mia_normalize_rows_sums(const float* x, int64_t rows, int64_t n, float eps, float* out, float* stats,
                            const void* sums, void* stream) {
  return normalize_rows(x, rows, n, eps, out, stats, const_cast<void*>(sums), 1, stream);
}

int mia_pad_rows_moments(const float* x, int64_t rows, int64_t n, int64_t zero_left, int64_t zero_right,
                         int group, float* out, void* sums, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && n >= 1 && zero_left >= 0 && zero_right >= 0, "negative shape");
  MIA_REQUIRE(group >= 1 && group <= 65535 && rows % group == 0,
              "the %lld rows do not split into groups of %d (at most 65535 rows per group)", (long long)rows, group);
  if (rows == 0) return MIA_OK;
  MIA_REQUIRE(x && out && sums, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  MIA_CHECK_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)(rows / group), s));
  const size_t np = (size_t)(zero_left + n + zero_right);
  for (int64_t r0 = 0; r0 < rows; r0 += 65535 / group * group) {  // whole groups per launch
    const int64_t step = 65535 / group * group;
    const int64_t nr = rows - r0 < step ? rows - r0 : step;
    long long chunks = (n + 256 * 16 - 1) / (256 * 16);
    const long long cap = 4096 / nr > 1 ? 4096 / nr : 1;
    if (chunks > cap) chunks = cap;
    hipLaunchKernelGGL(pad_rows_moments_kernel, dim3((unsigned)chunks, (unsigned)nr), dim3(256), 0, s,
                       x + (size_t)r0 * n, (long long)n, (long long)zero_left, (long long)zero_right, group,
                       out + (size_t)r0 * np, (double*)sums + 2 * (size_t)(r0 / group));
  }
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // extern "C"

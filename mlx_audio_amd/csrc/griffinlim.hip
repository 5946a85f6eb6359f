// Elementwise steps of Griffin-Lim phase reconstruction: building a complex spectrum from
// magnitudes and phases, and the per-iteration update that keeps the phase of a freshly
// computed spectrum, imposes the target magnitudes and applies the momentum term.  All arrays
// are walked in memory order (any layout, as long as it is the same for all of them).
#include "mia_internal.h"

namespace mia {

// out = mag * (cos a, sin a)
__global__ __launch_bounds__(256) void polar_kernel(const float* __restrict__ mag,
                                                    const float* __restrict__ angle, long long n,
                                                    float2* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s, c;
  sincosf(angle[i], &s, &c);
  const float m = mag[i];
  out[i] = {m * c, m * s};
}

// u = c_new / |c_new| (1 where c_new is zero, the value exp(i atan2(0, 0)));
// r = mag * u;  rebuilt = r + momentum (r - tprev) (or r when momentum == 0);  tprev = r.
__global__ __launch_bounds__(256) void griffinlim_update_kernel(const float2* __restrict__ c_new,
                                                                const float* __restrict__ mag,
                                                                float2* __restrict__ tprev,
                                                                float2* __restrict__ rebuilt,
                                                                long long n, float momentum) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float2 z = c_new[i];
  const float h = hypotf(z.x, z.y);
  float2 u = {1.0f, 0.0f};
  if (h > 0.0f) u = {z.x / h, z.y / h};
  const float m = mag[i];
  const float2 r = {m * u.x, m * u.y};
  float2 o = r;
  if (momentum > 0.0f) {
    const float2 p = tprev[i];
    o = {r.x + momentum * (r.x - p.x), r.y + momentum * (r.y - p.y)};
  }
  tprev[i] = r;
  rebuilt[i] = o;
}

// |z| or arg z of a spectrum held as (rows, frames, bins) — the order the STFT kernel writes —
// delivered as (rows, bins, frames), the order the reference's functions return: 32 x 32 tiles
// through LDS, read with bins fastest, written with frames fastest.  grid (tiles over frames,
// tiles over bins, rows), block (32, 8).
__global__ __launch_bounds__(256) void spectrum_part_kernel(const float2* __restrict__ z,
                                                            long long F, long long bins, int op,
                                                            float* __restrict__ out) {
  __shared__ float tile[32][33];
  const long long row = blockIdx.z;
  unsigned tile_f, tile_k;
  if (!xcd_tile((unsigned)((F + 31) / 32), (unsigned)((bins + 31) / 32), tile_f, tile_k)) return;
  const long long f0 = (long long)tile_f * 32, k0 = (long long)tile_k * 32;
  const float2* __restrict__ src = z + row * F * bins;
  for (int j = threadIdx.y; j < 32; j += 8) {  // j: frame in the tile, x: bin
    const long long f = f0 + j, k = k0 + threadIdx.x;
    float v = 0.0f;
    if (f < F && k < bins) {
      const float2 c = src[f * bins + k];
      v = op == 0 ? hypotf(c.x, c.y) : atan2f(c.y, c.x);
    }
    tile[j][threadIdx.x] = v;
  }
  __syncthreads();
  float* __restrict__ dst = out + row * bins * F;
  for (int j = threadIdx.y; j < 32; j += 8) {  // j: bin in the tile, x: frame
    const long long k = k0 + j, f = f0 + threadIdx.x;
    if (f < F && k < bins) dst[k * F + f] = tile[threadIdx.x][j];
  }
}

// A complex (rows, bins, frames) array with frames contiguous - the reference's own layout, what
// a caller holds who built the spectrum elsewhere - rewritten as (rows, frames, bins), the order
// the inverse STFT kernels read in whole rows (their strided path is about five times slower).
// Tiles of 32 bins x 64 frames through LDS (512-byte runs in, 256-byte runs out), xcd_tile()
// order, rows along grid.z, block (32, 8).
__global__ __launch_bounds__(256) void spectrum_transpose_kernel(const float2* __restrict__ in,
                                                                 long long bins, long long F,
                                                                 float2* __restrict__ out) {
  __shared__ float2 tile[64][33];
  const long long row = blockIdx.z;
  unsigned tile_f, tile_k;
  if (!xcd_tile((unsigned)((F + 63) / 64), (unsigned)((bins + 31) / 32), tile_f, tile_k)) return;
  const long long f0 = (long long)tile_f * 64, k0 = (long long)tile_k * 32;
  const float2* __restrict__ src = in + row * bins * F;
  const int fx = threadIdx.y % 2 * 32 + threadIdx.x;  // frame within the tile: 64 per bin row
#pragma unroll
  for (int j = threadIdx.y / 2; j < 32; j += 4) {  // j: bin within the tile
    const long long k = k0 + j, f = f0 + fx;
    float2 v = {0.0f, 0.0f};
    if (k < bins && f < F) v = src[k * F + f];
    tile[fx][j] = v;
  }
  __syncthreads();
  float2* __restrict__ dst = out + row * F * bins;
#pragma unroll
  for (int j = threadIdx.y; j < 64; j += 8) {  // j: frame within the tile, x: bin
    const long long f = f0 + j, k = k0 + threadIdx.x;
    if (k < bins && f < F) dst[f * bins + k] = tile[j][threadIdx.x];
  }
}

// Set-up of the Griffin-Lim loop in one pass: the target magnitudes S and the initial phases,
// both (rows, bins, frames) with frames contiguous as the reference holds them, become the loop's
// state in the (rows, frames, bins) order the STFT kernels use - the magnitudes transposed, and
// rebuilt = tprev = S * exp(i angle) (angle == nullptr: phase zero, S itself).  Tiles of 32 bins
// x 64 frames through LDS, xcd_tile() order, rows along grid.z, block (32, 8).
__global__ __launch_bounds__(256) void griffinlim_init_kernel(const float* __restrict__ S,
                                                              const float* __restrict__ angle,
                                                              long long bins, long long F,
                                                              float* __restrict__ mag_t,
                                                              float2* __restrict__ rebuilt,
                                                              float2* __restrict__ tprev) {
  __shared__ float2 tile[64][33];
  __shared__ float mtile[64][33];
  const long long row = blockIdx.z;
  unsigned tile_f, tile_k;
  if (!xcd_tile((unsigned)((F + 63) / 64), (unsigned)((bins + 31) / 32), tile_f, tile_k)) return;
  const long long f0 = (long long)tile_f * 64, k0 = (long long)tile_k * 32;
  const long long base = row * bins * F;
  const int fx = threadIdx.y % 2 * 32 + threadIdx.x;  // frame within the tile: 64 per bin row
#pragma unroll
  for (int j = threadIdx.y / 2; j < 32; j += 4) {  // j: bin within the tile
    const long long k = k0 + j, f = f0 + fx;
    float m = 0.0f;
    float2 z = {0.0f, 0.0f};
    if (k < bins && f < F) {
      m = S[base + k * F + f];
      z = {m, 0.0f};
      if (angle != nullptr) {
        float sn, cs;
        sincosf(angle[base + k * F + f], &sn, &cs);
        z = {m * cs, m * sn};
      }
    }
    mtile[fx][j] = m;
    tile[fx][j] = z;
  }
  __syncthreads();
#pragma unroll
  for (int j = threadIdx.y; j < 64; j += 8) {  // j: frame within the tile, x: bin
    const long long f = f0 + j, k = k0 + threadIdx.x;
    if (k < bins && f < F) {
      const long long o = base + f * bins + k;
      mag_t[o] = mtile[j][threadIdx.x];
      rebuilt[o] = tile[j][threadIdx.x];
      tprev[o] = tile[j][threadIdx.x];
    }
  }
}

}  // namespace mia

extern "C" {

int mia_spectrum_part(const void* z, int64_t rows, int64_t frames, int64_t bins, int op, float* out,
                      void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && frames >= 0 && bins >= 0, "negative shape");
  MIA_REQUIRE(op == 0 || op == 1, "op must be 0 (magnitude) or 1 (phase)");
  if (rows == 0 || frames == 0 || bins == 0) return MIA_OK;
  MIA_REQUIRE(z && out, "null pointer");
  MIA_REQUIRE((bins + 31) / 32 <= 65535, "too many bins for one launch");
  if (rows > 65535) {  // a launch takes 65535 rows: the rest follows in further launches
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
      const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
      const size_t off = (size_t)r0 * frames * bins;
      const int rc = mia_spectrum_part((const float2*)z + off, nr, frames, bins, op, out + off, stream);
      if (rc != MIA_OK) return rc;
    }
    return MIA_OK;
  }
  // tiles of neighbouring bins share the cache lines their 256-byte runs of a spectrum row
  // straddle: xcd_tile() order keeps them behind one L2
  MIA_REQUIRE((unsigned long long)((frames + 31) / 32) * ((bins + 31) / 32) < (1ULL << 28),
              "spectrum too large for one launch");
  const dim3 grid(xcd_tile_grid((unsigned)((frames + 31) / 32), (unsigned)((bins + 31) / 32)), 1,
                  (unsigned)rows);
  hipLaunchKernelGGL(spectrum_part_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream,
                     (const float2*)z, (long long)frames, (long long)bins, op, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_transpose_spectrum(const void* in, int64_t rows, int64_t bins, int64_t frames, void* out,
                           void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 0 && frames >= 0, "negative shape");
  if (rows == 0 || bins == 0 || frames == 0) return MIA_OK;
  MIA_REQUIRE(in && out && in != out, "null or aliased pointers");
  const unsigned tiles_f = (unsigned)((frames + 63) / 64), tiles_k = (unsigned)((bins + 31) / 32);
  MIA_REQUIRE((unsigned long long)tiles_f * tiles_k < (1ULL << 28), "spectrum too large for one launch");
  const size_t per_row = (size_t)bins * (size_t)frames;
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {  // a launch takes 65535 rows
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    hipLaunchKernelGGL(spectrum_transpose_kernel, dim3(xcd_tile_grid(tiles_f, tiles_k), 1, (unsigned)nr),
                       dim3(32, 8), 0, (hipStream_t)stream, (const float2*)in + (size_t)r0 * per_row,
                       (long long)bins, (long long)frames, (float2*)out + (size_t)r0 * per_row);
  }
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_griffinlim_init(const float* S, const float* angle, int64_t rows, int64_t bins, int64_t frames,
                        float* mag_t, void* rebuilt, void* tprev, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && bins >= 0 && frames >= 0, "negative shape");
  if (rows == 0 || bins == 0 || frames == 0) return MIA_OK;
  MIA_REQUIRE(S && mag_t && rebuilt && tprev, "null pointer");
  const unsigned tiles_f = (unsigned)((frames + 63) / 64), tiles_k = (unsigned)((bins + 31) / 32);
  MIA_REQUIRE((unsigned long long)tiles_f * tiles_k < (1ULL << 28), "spectrogram too large for one launch");
  const size_t per_row = (size_t)bins * (size_t)frames;
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {  // a launch takes 65535 rows
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    const size_t off = (size_t)r0 * per_row;
    hipLaunchKernelGGL(griffinlim_init_kernel, dim3(xcd_tile_grid(tiles_f, tiles_k), 1, (unsigned)nr),
                       dim3(32, 8), 0, (hipStream_t)stream, S + off, angle ? angle + off : nullptr,
                       (long long)bins, (long long)frames, mag_t + off, (float2*)rebuilt + off,
                       (float2*)tprev + off);
  }
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_polar(const float* mag, const float* angle, int64_t n, void* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(n >= 0, "n must be >= 0");
  if (n == 0) return MIA_OK;
  MIA_REQUIRE(mag && angle && out, "null pointer");
  hipLaunchKernelGGL(polar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, mag, angle, (long long)n, (float2*)out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_griffinlim_update(const void* c_new, const float* mag, void* tprev, void* rebuilt,
                          int64_t n, float momentum, void* stream) {
  using namespace mia;
  MIA_REQUIRE(n >= 0, "n must be >= 0");
  MIA_REQUIRE(momentum >= 0.0f && momentum < 1.0f, "momentum must be in [0, 1), got %g", momentum);
  if (n == 0) return MIA_OK;
  MIA_REQUIRE(c_new && mag && tprev && rebuilt, "null pointer");
  hipLaunchKernelGGL(griffinlim_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)c_new, mag, (float2*)tprev,
                     (float2*)rebuilt, (long long)n, momentum);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // extern "C"

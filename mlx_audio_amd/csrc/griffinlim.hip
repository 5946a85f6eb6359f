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

}  // namespace mia

extern "C" {

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

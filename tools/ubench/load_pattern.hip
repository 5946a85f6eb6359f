// How fast can a workgroup-per-tile pattern stream the 1024 x 480000 float input?
// Each tile = 10480 contiguous floats starting at tile*10240 - 200 inside a clip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int T = 480000, B = 1024, TILES = 47, SPAN4 = 2620;

template <int WG, int MODE>  // MODE 0: regs only; 1: + LDS write + barrier; 2: nontemporal
__global__ __launch_bounds__(WG) void k(const float* __restrict__ y, float* __restrict__ out,
                                        int tiles_per_wg, int total, int persistent) {
  __shared__ float lds[10640];
  constexpr int NPRE = (SPAN4 + WG - 1) / WG;
  float acc = 0.f;
  int first = persistent ? blockIdx.x * tiles_per_wg : blockIdx.x;
  int last = persistent ? min(first + tiles_per_wg, total) : first + 1;
  for (int idx = first; idx < last; ++idx) {
    int b = idx / TILES, tile = idx - b * TILES;
    long long s0 = (long long)b * T + (long long)tile * 10240 - 200;
    if (s0 < (long long)b * T) s0 = (long long)b * T;               // keep in bounds
    if (s0 + 10480 > (long long)(b + 1) * T) s0 = (long long)(b + 1) * T - 10480;
    const float4* src = reinterpret_cast<const float4*>(y + s0);
    float4 r[NPRE];
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      int q = threadIdx.x + i * WG;
      q = q < SPAN4 ? q : SPAN4 - 1;
      if (MODE == 2) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(src + q));
        r[i] = {t.x, t.y, t.z, t.w};
      } else {
        r[i] = src[q];
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        int q = threadIdx.x + i * WG;
        if (q < SPAN4) {
          float* d = lds + 161 * (q / 40) + 4 * (q % 40);
          d[0] = r[i].x; d[1] = r[i].y; d[2] = r[i].z; d[3] = r[i].w;
        }
      }
      __syncthreads();
      acc += lds[(threadIdx.x * 7) % 10000];
      __syncthreads();
    } else {
#pragma unroll
      for (int i = 0; i < NPRE; ++i) acc += r[i].x + r[i].y + r[i].z + r[i].w;
    }
  }
  if (acc == 12345.678f) out[blockIdx.x] = acc;
}

template <int WG, int MODE>
void run(const char* name, const float* y, float* out, int wgs_per_cu, int persistent) {
  const int total = B * TILES;
  int grid = persistent ? 256 * wgs_per_cu : total;
  int tpw = (total + grid - 1) / grid;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) k<WG, MODE><<<grid, WG>>>(y, out, tpw, total, persistent);
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) k<WG, MODE><<<grid, WG>>>(y, out, tpw, total, persistent);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-44s %7.3f ms  %5.2f TB/s\n", name, ms, 1.97e9 / ms / 1e9);
}

__global__ void fill_rand(float* y, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    y[i] = (float)(h & 0xFFFFFF) / 8388608.0f - 1.0f;
  }
}

int main() {
  float *y, *out;
  hipMalloc(&y, (size_t)B * T * 4); hipMalloc(&out, 1 << 20);
  fill_rand<<<4096, 256>>>(y, (size_t)B * T);
  hipDeviceSynchronize();
  run<1024, 0>("persistent 1/CU, 1024 thr, regs", y, out, 1, 1);
  run<1024, 0>("persistent 2/CU, 1024 thr, regs", y, out, 2, 1);
  run<1024, 1>("persistent 1/CU, 1024 thr, LDS+barrier", y, out, 1, 1);
  run<1024, 1>("persistent 2/CU, 1024 thr, LDS+barrier", y, out, 2, 1);
  run<1024, 2>("persistent 2/CU, 1024 thr, nontemporal", y, out, 2, 1);
  run<512, 0>("persistent 2/CU, 512 thr, regs", y, out, 2, 1);
  run<512, 0>("persistent 4/CU, 512 thr, regs", y, out, 4, 1);
  run<256, 0>("persistent 8/CU, 256 thr, regs", y, out, 8, 1);
  run<1024, 0>("one tile per WG (48128 WGs), 1024 thr", y, out, 1, 0);
  run<512, 0>("one tile per WG, 512 thr", y, out, 1, 0);
  run<256, 0>("one tile per WG, 256 thr", y, out, 1, 0);
  run<256, 1>("one tile per WG, 256 thr, LDS+barrier", y, out, 1, 0);
  return 0;
}

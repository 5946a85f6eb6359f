// Do an LDS-exchange workgroup and a VALU workgroup overlap on one CU?
#include "../../mlx_audio_amd/csrc/fft_engine.h"
#include <cstdio>
using namespace mia;

template <int MODE>  // 0: all WGs exchange; 1: all WGs VALU; 2: even WGs exchange, odd WGs VALU; 3: each WG alternates
__global__ __launch_bounds__(512, 4) void k(float* out, int iters) {
  __shared__ float2 ex[64 * 226 / 2];
  const int lane = threadIdx.x & 63, t = threadIdx.x >> 6;
  float2 y[7], v[16];
  for (int i = 0; i < 7; ++i) y[i] = {threadIdx.x + i * 1.0f, i * 2.0f};
  for (int i = 0; i < 16; ++i) v[i] = {threadIdx.x * 0.002f + i, 1.0f - i};
  float acc = 0.f;
  const bool do_ex = MODE == 0 || MODE == 3 || (MODE == 2 && (blockIdx.x & 1) == 0);
  const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && (blockIdx.x & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (do_ex) {
      const int exw = (226 * lane + 14 * t) / 2;
#pragma unroll
      for (int c = 0; c < 7; ++c) ex[exw + c] = y[c];
      __syncthreads();
      const int exr = (226 * lane) / 2 + (t % 7);
      float2 r[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] = ex[exr + 7 * q];
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += r[q].x - r[q].y;
#pragma unroll
      for (int i = 0; i < 7; ++i) y[i].x += acc * 1e-9f;
      __syncthreads();
    }
    if (do_valu) {
#pragma unroll 1
      for (int rep = 0; rep < 2; ++rep) {
        Dft<16>::run(v);
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc += fmaf(v[i].x, v[i].x, v[i].y * v[i].y); v[i].x *= 0.1f; v[i].y *= 0.1f; }
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + v[3].x + y[2].x;
}

template <int MODE>
void run(const char* name) {
  float* out; hipMalloc(&out, 8 << 20);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<512, 512>>>(out, iters);
  hipEventRecord(e0);
  k<MODE><<<512, 512>>>(out, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-48s %.3f ms\n", name, ms);
}

int main() {
  run<0>("2 WGs/CU, both LDS exchange");
  run<1>("2 WGs/CU, both VALU (2 x dft16)");
  run<2>("2 WGs/CU, one exchange + one VALU");
  run<3>("2 WGs/CU, each alternates exchange and VALU");
  return 0;
}

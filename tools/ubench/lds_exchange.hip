// Cost of an LDS exchange phase: NW b64 writes per thread, barrier, NR b64 reads, barrier.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WAVES, int NWR, int NRD, int ES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, long long* cyc, int iters) {
  extern __shared__ float2 ex[];
  const int lane = threadIdx.x & 63, t = threadIdx.x >> 6;
  float2 y[NWR], v[NRD];
  for (int i = 0; i < NWR; ++i) y[i] = {threadIdx.x + i * 1.0f, i * 2.0f};
  float acc = 0.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int exw = (ES * lane + 2 * NWR * t) / 2;
#pragma unroll
    for (int c = 0; c < NWR; ++c) ex[exw + c] = y[c];
    __syncthreads();
    const int exr = (ES * lane) / 2 + (t % NWR);
#pragma unroll
    for (int q = 0; q < NRD; ++q) v[q] = ex[exr + NWR * (q % WAVES)];
#pragma unroll
    for (int q = 0; q < NRD; ++q) acc += v[q].x - v[q].y;
#pragma unroll
    for (int i = 0; i < NWR; ++i) y[i].x += acc * 1e-9f;
    __syncthreads();
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int WAVES, int NWR, int NRD, int ES>
void run(const char* name, int wgs_per_cu) {
  float* out; long long* cyc;
  hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8);
  const int iters = 2000;
  const size_t lds = (size_t)64 * ES * 4;
  hipFuncSetAttribute((const void*)k<WAVES, NWR, NRD, ES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<WAVES, NWR, NRD, ES><<<256 * wgs_per_cu, WAVES * 64, lds>>>(out, cyc, iters);
  hipEventRecord(e0);
  k<WAVES, NWR, NRD, ES><<<256 * wgs_per_cu, WAVES * 64, lds>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s WGs/CU=%d  %.3f ms  -> %.0f cycles per exchange (2.1 GHz), per CU-exchange %.0f\n", name,
         wgs_per_cu, ms, ms * 1e-3 * 2.1e9 / iters, ms * 1e-3 * 2.1e9 / iters / wgs_per_cu);
}

int main() {
  run<16, 13, 16, 418>("16 waves, 13 wr, 16 rd (107 KB)", 1);
  run<16, 7, 16, 226>("16 waves, 7 wr, 16 rd (58 KB)", 1);
  run<16, 7, 16, 226>("16 waves, 7 wr, 16 rd (58 KB)", 2);
  run<8, 13, 8, 210>("8 waves, 13 wr, 8 rd (54 KB)", 1);
  run<8, 13, 8, 210>("8 waves, 13 wr, 8 rd (54 KB)", 2);
  run<8, 13, 16, 210>("8 waves, 13 wr, 16 rd (54 KB)", 2);
  run<4, 13, 16, 210>("4 waves, 13 wr, 16 rd (54 KB)", 2);
  run<16, 1, 1, 226>("16 waves, 1 wr, 1 rd (barrier cost)", 1);
  run<16, 1, 1, 226>("16 waves, 1 wr, 1 rd (barrier cost)", 2);
  run<8, 1, 1, 226>("8 waves, 1 wr, 1 rd (barrier cost)", 2);
  return 0;
}

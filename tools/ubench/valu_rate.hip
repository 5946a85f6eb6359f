// Issue-rate microbenchmark: cycles per wave-instruction of v_fma_f32 vs
// v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 with 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  v2f a[8];
  float s[16];
  for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x + i, 1.0f + i};
  for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 0.5f + i;
  v2f m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = __builtin_fmaf(s[i], 1.0001f, 0.5f);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = a[i] + c;
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = a[i] + m;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = a[i] * m;
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = a[i] * m;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
  for (int i = 0; i < 16; ++i) r += s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 1024 * 1024 * 4);
  hipMalloc(&cyc, 8);
  const int iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e0);
  k<MODE><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double per_wave_instr = (double)h / (iters * 16.0);
  const int waves_per_simd = threads / 256;
  printf("%-12s waves/SIMD=%d  cycles(counter)/instr per wave=%.2f  per SIMD=%.2f   ms=%.3f\n", name,
         waves_per_simd ? waves_per_simd : 1, per_wave_instr,
         per_wave_instr / (waves_per_simd ? waves_per_simd : 1), ms);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int threads : {256, 512, 1024}) {
    run<0>("v_fma_f32", threads);
    run<1>("v_pk_fma_f32", threads);
    run<2>("v_pk_add_f32", threads);
    run<3>("v_pk_mul_f32", threads);
  }
  return 0;
}

// VALU throughput of the actual butterfly code: rdft25 + 12 twiddle cmuls, and Dft<16> + power,
// on register data only (no LDS, no barriers), at 4 and 8 waves per SIMD.
#include "../../mlx_audio_amd/csrc/fft_engine.h"
#include <cstdio>
using namespace mia;

template <int MODE, int WPS>
__global__ __launch_bounds__(256 * WPS / 1, WPS) void k(float* out, int iters, const float2* __restrict__ tw) {
  float x[25];
  float2 v[16];
  for (int i = 0; i < 25; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 16; ++i) v[i] = {threadIdx.x * 0.002f + i, 1.0f - i};
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      float2 y[13];
      rdft25(x, y);
#pragma unroll
      for (int kk = 1; kk < 13; ++kk) y[kk] = cmul(y[kk], tw[kk]);
#pragma unroll
      for (int i = 0; i < 25; ++i) x[i] = (i & 1 ? y[i / 2].x : y[i / 2].y) * 0.01f;
    } else {
      Dft<16>::run(v);
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc += fmaf(v[i].x, v[i].x, v[i].y * v[i].y); v[i].x *= 0.1f; v[i].y *= 0.1f; }
    }
  }
  float r = acc;
  for (int i = 0; i < 25; ++i) r += x[i];
  for (int i = 0; i < 16; ++i) r += v[i].x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE, int WPS>
void run(const char* name, int valu_per_iter) {
  float* out; float2* tw;
  hipMalloc(&out, 256 * 1024 * 4 * 4);
  hipMalloc(&tw, 16 * 8);
  hipMemset(tw, 0, 128);
  const int iters = 2000, threads = 256 * WPS;  // WPS waves on each of the 4 SIMDs
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, WPS><<<256, threads>>>(out, iters, tw);
  hipEventRecord(e0);
  k<MODE, WPS><<<256, threads>>>(out, iters, tw);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = (double)iters * valu_per_iter * WPS;
  printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f cycles per VALU instr per SIMD at 2.1 GHz (assuming %d VALU/iter)\n",
         name, WPS, ms, ms * 1e-3 * 2.1e9 / instr_per_simd, valu_per_iter);
}

int main() {
  run<0, 1>("rdft25+tw", 255); run<0, 2>("rdft25+tw", 255); run<0, 4>("rdft25+tw", 255);
  run<1, 1>("dft16+pow", 230); run<1, 2>("dft16+pow", 230); run<1, 4>("dft16+pow", 230);
  return 0;
}

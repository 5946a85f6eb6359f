// HBM write rate by bytes per lane and store instruction: every workgroup writes its own
// contiguous 16 KiB rows (one row per iteration, rows dealt round-robin), as a forward-STFT
// workgroup does with its frames' spectra.  MODE: bytes per lane (4, 8, 16).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int BYTES>
__global__ __launch_bounds__(256) void k(float* out, long long rows, int iters_dummy) {
  const int tid = threadIdx.x;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    float* row = out + r * 4096;  // 16 KiB
    if (BYTES == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) row[tid + 256 * i] = (float)i;
    } else if (BYTES == 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) reinterpret_cast<float2*>(row)[tid + 256 * i] = float2{(float)i, 1.0f};
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) reinterpret_cast<float4*>(row)[tid + 256 * i] = float4{(float)i, 1.0f, 2.0f, 3.0f};
    }
  }
}

template <int BYTES>
void run(float* out, long long rows, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<BYTES><<<grid, 256>>>(out, rows, 0);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) k<BYTES><<<grid, 256>>>(out, rows, 0);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  printf("%2d bytes per lane, grid %5d: %.3f ms  %.2f TB/s\n", BYTES, grid, ms, rows * 16384.0 / ms / 1e9);
}

int main() {
  const long long rows = 1 << 20;  // 16 GiB
  float* out;
  if (hipMalloc(&out, rows * 16384) != hipSuccess) return 1;
  for (int grid : {512, 2048, 16384}) {
    run<4>(out, rows, grid);
    run<8>(out, rows, grid);
    run<16>(out, rows, grid);
  }
  hipFree(out);
  return 0;
}

// What a plain copy moves on this box, in the Whisper front end's mix (1.92 MB read and 1.54 MB
// written per clip, 1024 clips): float4 loads and stores, contiguous 16 KiB pieces per workgroup
// and iteration.  Variants: grid size, loads in flight per lane, nontemporal hints, reads only.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float vf4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT, bool NT, bool WRITE>
__global__ __launch_bounds__(256) void k(const vf4* __restrict__ in, vf4* __restrict__ out, long long pieces,
                                         float* sink) {
  const int tid = threadIdx.x;
  float acc = 0.0f;
  for (long long r = blockIdx.x; r < pieces; r += gridDim.x) {
    const vf4* src = in + r * (256 * INFLIGHT);
    vf4* dst = out + (r * (256 * INFLIGHT) * 4) / 5;  // 0.8 bytes written per byte read
    vf4 v[INFLIGHT];
#pragma unroll
    for (int i = 0; i < INFLIGHT; ++i) {
      v[i] = NT ? __builtin_nontemporal_load(src + tid + 256 * i) : src[tid + 256 * i];
    }
#pragma unroll
    for (int i = 0; i < INFLIGHT; ++i) {
      if (WRITE) {
        if (i * 5 < INFLIGHT * 4) {
          if (NT) {
            __builtin_nontemporal_store(v[i], dst + tid + 256 * i);
          } else {
            dst[tid + 256 * i] = v[i];
          }
        } else {
          acc += v[i].x;
        }
      } else {
        acc += v[i].x + v[i].w;
      }
    }
  }
  if (acc == 12345.678f) *sink = acc;
}

template <int INFLIGHT, bool NT, bool WRITE>
void run(const vf4* in, vf4* out, long long bytes_in, int grid, float* sink) {
  const long long pieces = bytes_in / (16 * 256 * INFLIGHT);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<INFLIGHT, NT, WRITE><<<grid, 256>>>(in, out, pieces, sink);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) k<INFLIGHT, NT, WRITE><<<grid, 256>>>(in, out, pieces, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double moved = WRITE ? bytes_in * 1.8 : (double)bytes_in;
  printf("%s %d x float4 in flight%s, grid %5d: %.3f ms  %.2f TB/s\n", WRITE ? "read 1 : write 0.8," : "read only,          ",
         INFLIGHT, NT ? ", nontemporal" : "             ", grid, ms, moved / ms / 1e9);
}

int main() {
  const long long bytes_in = 5LL << 30;  // 5 GiB read, 4 GiB written
  vf4 *in, *out;
  float* sink;
  if (hipMalloc(&in, bytes_in) != hipSuccess || hipMalloc(&out, bytes_in) != hipSuccess ||
      hipMalloc(&sink, 4) != hipSuccess) {
    return 1;
  }
  hipMemset(in, 0, bytes_in);
  for (int grid : {512, 2048, 16384}) {
    run<5, false, true>(in, out, bytes_in, grid, sink);
    run<5, true, true>(in, out, bytes_in, grid, sink);
    run<10, false, true>(in, out, bytes_in, grid, sink);
    run<10, true, true>(in, out, bytes_in, grid, sink);
    run<5, false, false>(in, out, bytes_in, grid, sink);
    run<10, true, false>(in, out, bytes_in, grid, sink);
  }
  hipFree(in);
  hipFree(out);
  return 0;
}

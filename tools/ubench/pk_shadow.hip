// Does scalar / LDS issue hide in the shadow of packed fp32 vector instructions?  A packed
// v_pk_fma_f32 does the work of two v_fma_f32 in one issue slot but holds the vector ALU twice
// as long.  Every wave (4 per SIMD) runs the same stream of independent operations:
//   MODE 0: 64 v_fma_f32                      MODE 1: 32 v_pk_fma_f32 (same flops)
//   MODE 2: 64 v_fma_f32 + 32 s_add_u32       MODE 3: 32 v_pk_fma_f32 + 32 s_add_u32
//   MODE 4: 64 v_fma_f32 + 16 ds_read_b32     MODE 5: 32 v_pk_fma_f32 + 16 ds_read_b32
// interleaved.  If modes 3 / 5 take the time of mode 1 while 2 / 4 take more than mode 0, the
// other instructions issue during the packed operations' second half.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  __shared__ float buf[1024];
  buf[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float s[16];
  v2f p[8];
  for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 0.5f + i;
  for (int i = 0; i < 8; ++i) p[i] = v2f{s[2 * i], s[2 * i + 1]};
  const v2f m2 = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};
  unsigned a0 = 1, a1 = 2, a2 = 3, a3 = 4;
  const unsigned la = (unsigned)(size_t)(buf + (threadIdx.x & 63));
  float t0 = 0, t1 = 0, t2 = 0, t3 = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == 0 || MODE == 2 || MODE == 4) {
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
          if (MODE == 2) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\ns_add_u32 %4, %4, 3\nv_fma_f32 %1, %1, %8, %9\ns_add_u32 %5, %5, 5\n"
                         "v_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9"
                         : "+v"(s[i]), "+v"(s[i + 1]), "+v"(s[i + 2]), "+v"(s[i + 3]), "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3)
                         : "v"(1.0001f), "v"(0.5f));
          } else if (MODE == 4) {
            asm volatile("v_fma_f32 %0, %0, %5, %6\nds_read_b32 %4, %7\nv_fma_f32 %1, %1, %5, %6\n"
                         "v_fma_f32 %2, %2, %5, %6\nv_fma_f32 %3, %3, %5, %6"
                         : "+v"(s[i]), "+v"(s[i + 1]), "+v"(s[i + 2]), "+v"(s[i + 3]), "=v"(t0)
                         : "v"(1.0001f), "v"(0.5f), "v"(la) : "memory");
          } else {
            asm volatile("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5"
                         : "+v"(s[i]), "+v"(s[i + 1]), "+v"(s[i + 2]), "+v"(s[i + 3]) : "v"(1.0001f), "v"(0.5f));
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          if (MODE == 3) {
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\ns_add_u32 %2, %2, 3\nv_pk_fma_f32 %1, %1, %4, %5\ns_add_u32 %3, %3, 5"
                         : "+v"(p[i]), "+v"(p[i + 1]), "+s"(a0), "+s"(a1) : "v"(m2), "v"(c2));
          } else if (MODE == 5) {
            asm volatile("v_pk_fma_f32 %0, %0, %3, %4\nds_read_b32 %2, %5\nv_pk_fma_f32 %1, %1, %3, %4"
                         : "+v"(p[i]), "+v"(p[i + 1]), "=v"(t1) : "v"(m2), "v"(c2), "v"(la) : "memory");
          } else {
            asm volatile("v_pk_fma_f32 %0, %0, %2, %3\nv_pk_fma_f32 %1, %1, %2, %3"
                         : "+v"(p[i]), "+v"(p[i + 1]) : "v"(m2), "v"(c2));
          }
        }
      }
    }
    if (MODE >= 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float r = (float)(a0 ^ a1 ^ a2 ^ a3) + t0 + t1 + t2 + t3;
  for (int i = 0; i < 16; ++i) r += s[i];
  for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name) {
  float* out;
  hipMalloc(&out, 1024 * 1024 * 4);
  const int iters = 8192;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<256, 1024>>>(out, iters);
  hipEventRecord(e0);
  k<MODE><<<256, 1024>>>(out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.3f ms\n", name, ms);
  hipFree(out);
}

int main() {
  run<0>("64 v_fma");
  run<1>("32 v_pk_fma (same flops)");
  run<2>("64 v_fma + 32 s_add");
  run<3>("32 v_pk_fma + 32 s_add");
  run<4>("64 v_fma + 16 ds_read_b32");
  run<5>("32 v_pk_fma + 16 ds_read_b32");
  return 0;
}

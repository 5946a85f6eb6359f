// Issue-rate microbenchmark: cycles per wave-instruction of scalar ALU code, of taken
// scalar branches, and of scalar code running beside vector FMAs of other waves, with
// 1, 2, 4 waves per SIMD (256, 512, 1024 threads per workgroup, one workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: 64 independent s_add per iteration (every wave)
// MODE 1: 16 taken branches per iteration (every wave)
// MODE 2: odd waves 64 s_add, even waves 64 v_fma per iteration
// MODE 3: 64 v_fma per iteration (every wave)
// MODE 4: every wave alternates 1 v_fma + 1 s_add, 64 of each per iteration
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  __shared__ float buf[4096];
  buf[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float s[16];
  for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 0.5f + i;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned acc = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const bool scalar_wave = MODE == 0 || (MODE == 2 && (wave & 1));
    if (MODE == 5 || (MODE == 6 && (wave & 1))) {
      const float* q = buf + (threadIdx.x & 63);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t[i]) : "v"((unsigned)(size_t)q), "n"(256 * 0) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(t[i]));
      }
      continue;
    }
    if (MODE == 1) {
      asm volatile(
          "s_branch 1f\n1:\ns_branch 2f\n2:\ns_branch 3f\n3:\ns_branch 4f\n4:\n"
          "s_branch 5f\n5:\ns_branch 6f\n6:\ns_branch 7f\n7:\ns_branch 8f\n8:\n"
          "s_branch 9f\n9:\ns_branch 10f\n10:\ns_branch 11f\n11:\ns_branch 12f\n12:\n"
          "s_branch 13f\n13:\ns_branch 14f\n14:\ns_branch 15f\n15:\ns_branch 16f\n16:\n" ::: "memory");
    } else if (MODE == 4) {
      unsigned a0 = acc, a1 = acc + 1, a2 = acc + 2, a3 = acc + 3;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
          asm volatile("v_fma_f32 %0, %0, %8, %9\ns_add_u32 %4, %4, 3\nv_fma_f32 %1, %1, %8, %9\ns_add_u32 %5, %5, 5\n"
                       "v_fma_f32 %2, %2, %8, %9\ns_add_u32 %6, %6, 7\nv_fma_f32 %3, %3, %8, %9\ns_add_u32 %7, %7, 9"
                       : "+v"(s[i]), "+v"(s[i + 1]), "+v"(s[i + 2]), "+v"(s[i + 3]), "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3)
                       : "v"(1.0001f), "v"(0.5f));
        }
      }
      acc = a0 ^ a1 ^ a2 ^ a3;
    } else if (scalar_wave) {
      unsigned a0 = acc, a1 = acc + 1, a2 = acc + 2, a3 = acc + 3;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("s_add_u32 %0, %0, 3\ns_add_u32 %1, %1, 5\ns_add_u32 %2, %2, 7\ns_add_u32 %3, %3, 9"
                     : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3));
      }
      acc = a0 ^ a1 ^ a2 ^ a3;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = __builtin_fmaf(s[i], 1.0001f, 0.5f);
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float r = (float)acc;
  for (int i = 0; i < 16; ++i) r += s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, int per_iter) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 1024 * 1024 * 4);
  hipMalloc(&cyc, 8 * 16);
  const int iters = 4096;
  k<MODE><<<256, threads>>>(out, cyc, iters);
  k<MODE><<<256, threads>>>(out, cyc, iters);
  hipDeviceSynchronize();
  long long h[16];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const int waves = threads / 64;
  printf("%-34s waves/SIMD %d:", name, waves / 4 ? waves / 4 : 1);
  for (int w = 0; w < waves; w += (waves > 4 ? waves / 4 : 1))
    printf("  w%-2d %.2f cyc/instr", w, (double)h[w] / ((double)iters * per_iter));
  printf("\n");
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int threads : {256, 512, 1024}) {
    run<0>("s_add x64", threads, 64);
    run<1>("taken s_branch x16", threads, 16);
    run<3>("v_fma x64", threads, 64);
    run<2>("odd waves s_add, even v_fma", threads, 64);
    run<4>("v_fma + s_add interleaved (x64 each)", threads, 64);
    run<5>("ds_read_b32 x64", threads, 64);
    run<6>("odd waves ds_read_b32, even v_fma", threads, 64);
  }
  return 0;
}

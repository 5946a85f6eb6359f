// Memory-bound helper kernels: gather overlap-add, pad, frame, dB conversion,
// floor clamp of log features, DCT-II along a middle axis, |z| and arg z.
#include "mia_internal.h"

namespace mia {

// ---------------------------------------------------------------------------
// Overlap-add, gather form: one thread per output sample, frames visited in
// ascending order, no atomics, so results are deterministic.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ola_kernel(const float* __restrict__ frames,
                                                   const float* __restrict__ win, long long F,
                                                   int n_fft, int hop, long long out_len,
                                                   float scale, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long b = blockIdx.y;
  if (i >= out_len) return;
  long long first = i - n_fft + 1;
  first = first <= 0 ? 0 : (first + hop - 1) / hop;
  long long last = i / hop;
  if (last > F - 1) last = F - 1;
  const float* __restrict__ fb = frames + b * F * (long long)n_fft;
  float sum = 0.0f, wsum = 0.0f;
  for (long long f = first; f <= last; ++f) {
    const int s = (int)(i - f * hop);
    const float w = win[s];
    sum = fmaf(w, fb[f * n_fft + s], sum);  // (spelled out: the fused inverse kernel must
    wsum = fmaf(w, w, wsum);                // round exactly like this one)
  }
  out[b * out_len + i] = sum / fmaxf(wsum, 1e-8f) * scale;
}

int launch_overlap_add(const float* frames, const float* win, long long B, long long F,
                       int n_fft, int hop, long long out_len, float scale, float* out,
                       hipStream_t s) {
  if (B <= 0 || out_len <= 0) return MIA_OK;
  dim3 grid((unsigned)((out_len + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(ola_kernel, grid, dim3(256), 0, s, frames, win, F, n_fft, hop, out_len,
                     scale, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

// ---------------------------------------------------------------------------
// pad / frame
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad_kernel(const float* __restrict__ x, long long T,
                                                   long long pad, int mode,
                                                   float* __restrict__ out) {
  const long long L = T + 2 * pad;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long b = blockIdx.y;
  if (i >= L) return;
  const long long src = i - pad;
  float v;
  if (src >= 0 && src < T) {
    v = x[b * T + src];
  } else if (mode == MIA_PAD_CONSTANT) {
    v = 0.0f;
  } else if (mode == MIA_PAD_EDGE) {
    v = x[b * T + (src < 0 ? 0 : T - 1)];
  } else {
    const long long r = src < 0 ? -src : 2 * (T - 1) - src;
    v = x[b * T + r];
  }
  out[b * L + i] = v;
}

__global__ __launch_bounds__(256) void frame_kernel(const float* __restrict__ x, long long T,
                                                     long long F, int L, int hop,
                                                     float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long b = blockIdx.y;
  if (idx >= F * L) return;
  const long long f = idx / L;
  const int i = (int)(idx - f * L);
  out[b * F * L + idx] = x[b * T + f * hop + i];
}

// ---------------------------------------------------------------------------
// dB conversion of an existing array with a per-row or global top_db floor
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_db_kernel(const float* __restrict__ x, long long n,
                                                     float amin, float inv_ref, float coef,
                                                     float* __restrict__ out,
                                                     int* __restrict__ max_keys, int* __restrict__ min_keys,
                                                     int vec4) {
  const long long b = blockIdx.y;
  int kmax = kKeyMinusInf;  // integer keys: a NaN value wins the maximum, as in the reference
  float vmin = INFINITY;
  auto one = [&](float xv) {
    const float v = coef * log10f((xv < amin ? amin : xv) * inv_ref);  // a NaN stays a NaN
    kmax = max(kmax, float_to_key_nan(v));
    vmin = fminf(vmin, v);
    return v;
  };
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (vec4) {  // rows of whole, 16-byte-aligned quads
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x + b * n);
    float4* __restrict__ o4 = reinterpret_cast<float4*>(out + b * n);
    for (long long i = tid; i < n / 4; i += stride) {
      const float4 q = x4[i];
      o4[i] = {one(q.x), one(q.y), one(q.z), one(q.w)};
    }
  } else {
    for (long long i = tid; i < n; i += stride) out[b * n + i] = one(x[b * n + i]);
  }
  if (max_keys == nullptr) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    kmax = max(kmax, __shfl_xor(kmax, off));
    vmin = fminf(vmin, __shfl_xor(vmin, off));
  }
  __shared__ int red[4];
  __shared__ float redm[4];
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = kmax;
    redm[threadIdx.x >> 6] = vmin;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(max_keys + b, max(max(red[0], red[1]), max(red[2], red[3])));
    atomicMin(min_keys + b, float_to_key(fminf(fminf(redm[0], redm[1]), fminf(redm[2], redm[3]))));
  }
}

// out = ref * 10^(x / div): decibels back to power (div 10) or amplitude (div 20).
__global__ __launch_bounds__(256) void from_db_kernel(const float* __restrict__ x, long long n, float ref,
                                                       float div, float* __restrict__ out, int vec4) {
  auto one = [&](float v) { return ref * exp10f(v / div); };
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (vec4) {
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    float4* __restrict__ o4 = reinterpret_cast<float4*>(out);
    for (long long i = tid; i < n / 4; i += stride) {
      const float4 q = x4[i];
      o4[i] = {one(q.x), one(q.y), one(q.z), one(q.w)};
    }
    for (long long i = (n / 4) * 4 + tid; i < n; i += stride) out[i] = one(x[i]);
  } else {
    for (long long i = tid; i < n; i += stride) out[i] = one(x[i]);
  }
}

// keys[B] -> keys[B] holds max over all rows (one workgroup).
__global__ __launch_bounds__(256) void reduce_keys_kernel(int* __restrict__ keys, long long B) {
  int k = kKeyMinusInf;
  for (long long i = threadIdx.x; i < B; i += 256) k = max(k, keys[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) k = max(k, __shfl_xor(k, off));
  __shared__ int red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = k;
  __syncthreads();
  if (threadIdx.x == 0) keys[B] = max(max(red[0], red[1]), max(red[2], red[3]));
}

int launch_reduce_keys(int* keys, long long B, hipStream_t s) {
  hipLaunchKernelGGL(reduce_keys_kernel, dim3(1), dim3(256), 0, s, keys, B);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

__device__ __forceinline__ float floor_from_key(int key, int epi, float top) {
  const float gmax = key_to_float(key);
  if (epi == MIA_EPI_WHISPER) return ((gmax - top) + 4.0f) * 0.25f;
  return gmax - top;
}

// Both key arrays of a mel call in one launch (two hipMemsetD32Async cost four fill
// dispatches per call).
__global__ __launch_bounds__(256) void init_keys_kernel(int* __restrict__ max_keys,
                                                         int* __restrict__ min_keys, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    max_keys[i] = kKeyMinusInf;
    min_keys[i] = kKeyPlusInf;
  }
}
int launch_init_keys(int* max_keys, int* min_keys, long long n, hipStream_t s) {
  if (n <= 0) return MIA_OK;
  hipLaunchKernelGGL(init_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, max_keys,
                     min_keys, n);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

// Floor clamp of the (B, n) feature tensor.  A clip whose minimum is already above
// its floor is never touched (the common case: the log-mel range of most inputs is
// narrower than the clamp window), so the launch is sized for the few that are: every
// workgroup first derives the same list of clips to clamp (1024 clips per round, flags
// gathered by all threads, compacted in clip order by one wave), then the workgroups
// share the (clip, chunk) items of that list.
constexpr int kFloorChunks = 64;  // pieces a clip is cut into
#ifndef MIA_FLOOR_GRID
#define MIA_FLOOR_GRID 2048
#endif
__global__ __launch_bounds__(256) void floor_clips_kernel(float* __restrict__ x, long long n,
                                                           long long B,
                                                           const int* __restrict__ max_keys,
                                                           long long key_stride,
                                                           const int* __restrict__ min_keys,
                                                           int epi, float top) {
  __shared__ unsigned char need[1024];
  __shared__ short list[1024];
  __shared__ int count;
  const int tid = threadIdx.x, lane = tid & 63;
  const long long chunk_len = (n + kFloorChunks - 1) / kFloorChunks;
  for (long long b0 = 0; b0 < B; b0 += 1024) {
    for (int j = tid; j < 1024; j += 256) {
      bool nd = false;
      if (b0 + j < B) {
        const float thr = floor_from_key(max_keys[(b0 + j) * key_stride], epi, top);
        nd = thr != thr || key_to_float(min_keys[b0 + j]) < thr;  // a NaN maximum floods its clip
      }
      need[j] = nd ? 1 : 0;
    }
    __syncthreads();
    if (tid < 64) {
      int base = 0;
      for (int g = 0; g < 16; ++g) {
        const bool nd = need[g * 64 + lane] != 0;
        const unsigned long long m = __ballot(nd);
        if (nd) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (short)(g * 64 + lane);
        base += __popcll(m);
      }
      if (lane == 0) count = base;
    }
    __syncthreads();
    const int items = count * kFloorChunks;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
      const long long b = b0 + list[item / kFloorChunks];
      const long long lo = (long long)(item % kFloorChunks) * chunk_len;
      const long long hi = lo + chunk_len < n ? lo + chunk_len : n;
      const float thr = floor_from_key(max_keys[b * key_stride], epi, top);
      float* __restrict__ xb = x + b * n;
      for (long long i = lo + tid; i < hi; i += 256) {
        const float v = xb[i];
        if (v < thr || thr != thr) xb[i] = thr;
      }
    }
    __syncthreads();  // the lists are rebuilt for the next 1024 clips
  }
}

int launch_floor_clips(float* x, long long B, long long n, int* max_keys, const int* min_keys,
                       int epi, float top, int per_clip, hipStream_t s);

// Stand-alone dB conversion: one pass that writes the values and takes every row's maximum and
// minimum, then the floor pass of the fused kernels, which touches only rows that reach below
// their floor (most inputs span less than top_db: the values are then read once and written once).
// keys: 2 (B + 1) ints of workspace.
int launch_to_db(const float* x, long long B, long long n, float ref, float amin, float coef,
                 float top_db, int per_clip, float* out, int* keys, hipStream_t s) {
  if (B <= 0 || n <= 0) return MIA_OK;
  const bool clamp = top_db >= 0.0f;
  int* const min_keys = keys ? keys + (B + 1) : nullptr;
  if (clamp) {
    const int rc = launch_init_keys(keys, min_keys, B + 1, s);
    if (rc != MIA_OK) return rc;
  }
  const float refv = fmaxf(ref, amin);
  const int vec4 = n % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  long long bx = (n + 256 * 16 - 1) / (256 * 16);
  if (bx > 4096) bx = 4096;
  if (bx < 1) bx = 1;
  dim3 grid((unsigned)bx, (unsigned)B);
  hipLaunchKernelGGL(to_db_kernel, grid, dim3(256), 0, s, x, n, amin, 1.0f / refv, coef, out,
                     clamp ? keys : nullptr, min_keys, vec4);
  MIA_CHECK_HIP(hipGetLastError());
  if (clamp) return launch_floor_clips(out, B, n, keys, min_keys, MIA_EPI_DB, top_db, per_clip, s);
  return MIA_OK;
}

int launch_from_db(const float* x, long long n, float ref, float div, float* out, hipStream_t s) {
  if (n <= 0) return MIA_OK;
  const int vec4 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  long long bx = (n + 256 * 16 - 1) / (256 * 16);
  if (bx > 65536) bx = 65536;
  hipLaunchKernelGGL(from_db_kernel, dim3((unsigned)bx), dim3(256), 0, s, x, n, ref, div, out, vec4);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int launch_floor_clips(float* x, long long B, long long n, int* max_keys, const int* min_keys,
                       int epi, float top, int per_clip, hipStream_t s) {
  if (B <= 0 || n <= 0) return MIA_OK;
  if (!per_clip) {
    hipLaunchKernelGGL(reduce_keys_kernel, dim3(1), dim3(256), 0, s, max_keys, B);
    MIA_CHECK_HIP(hipGetLastError());
  }
  // Every workgroup first derives the list of clips to clamp (it reads all the keys), so the
  // grid stays small: enough workgroups to stream the clips that do need the floor, few enough
  // that the common launch - nothing to clamp - costs a few microseconds.
  long long grid = B * kFloorChunks;  // never more workgroups than items
  if (grid > MIA_FLOOR_GRID) grid = MIA_FLOOR_GRID;
  hipLaunchKernelGGL(floor_clips_kernel, dim3((unsigned)grid), dim3(256), 0, s, x, n, B,
                     per_clip ? max_keys : max_keys + B, (long long)(per_clip ? 1 : 0),
                     min_keys, epi, top);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

// ---------------------------------------------------------------------------
// DCT-II along the middle axis: out[o,k,i] = sum_n basis[k,n] * max(x[o,n,i], floor_o)
// One workgroup = 64 consecutive `inner` positions of one `o`; the (n_in x 64)
// slab is staged in LDS once and every wave produces a share of the outputs.
// ---------------------------------------------------------------------------
template <int KB>
__global__ __launch_bounds__(256) void dct2_kernel(const float* __restrict__ x, int n_in,
                                                    long long inner,
                                                    const float* __restrict__ basis, int n_out,
                                                    float* __restrict__ out,
                                                    const int* __restrict__ keys,
                                                    long long key_stride, float top) {
  extern __shared__ float slab[];  // n_in x 64
  const long long o = blockIdx.y;
  const long long i0 = (long long)blockIdx.x * 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: basis loads become s_load
  const float thr = keys ? floor_from_key(keys[o * key_stride], MIA_EPI_DB, top) : -INFINITY;
  const bool ok = i0 + lane < inner;
  for (int n = wave; n < n_in; n += 4) {
    float v = ok ? x[(o * n_in + n) * inner + i0 + lane] : 0.0f;
    slab[n * 64 + lane] = floor_nan(v, thr);
  }
  __syncthreads();
  for (int k0 = wave * KB; k0 < n_out; k0 += 4 * KB) {
    float acc[KB];
#pragma unroll
    for (int q = 0; q < KB; ++q) acc[q] = 0.0f;
    // 8 taps at a time: the basis values are wave-uniform, and in runs of 8 they arrive
    // as wide scalar loads instead of one load (and one wait) per value
    int n = 0;
    for (; n + 8 <= n_in; n += 8) {
      float xv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j] = slab[(n + j) * 64 + lane];
#pragma unroll
      for (int q = 0; q < KB; ++q) {
        const int k = k0 + q < n_out ? k0 + q : n_out - 1;
        const float* __restrict__ brow = basis + (long long)k * n_in + n;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[q] = fmaf(brow[j], xv[j], acc[q]);
      }
    }
    for (; n < n_in; ++n) {
      const float xv = slab[n * 64 + lane];
#pragma unroll
      for (int q = 0; q < KB; ++q) {
        const int k = k0 + q < n_out ? k0 + q : n_out - 1;
        acc[q] = fmaf(basis[(long long)k * n_in + n], xv, acc[q]);
      }
    }
    if (ok) {
#pragma unroll
      for (int q = 0; q < KB; ++q) {
        if (k0 + q < n_out) out[(o * n_out + k0 + q) * inner + i0 + lane] = acc[q];
      }
    }
  }
}

// The same product on the matrix cores, for n_in <= 128 and n_out <= 48 (MFCC: 128 mel
// bands -> 40 coefficients): out = basis (n_out x n_in) . max(x, floor) (n_in x 64 positions)
// per tile, v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation).  Persistent
// workgroups: wave w owns positions 16 w .. 16 w + 15 of every tile and keeps the A
// operands of all three 16-row blocks of the basis in registers for its whole run
//   A(mb, ks): lane l holds basis[16 mb + l % 16][4 ks + l / 16]   (0 outside the matrix)
//   B(ks)    : lane l holds x[4 ks + l / 16][16 w + l % 16]         (from the LDS slab)
//   D(mb)    : lane l holds rows 16 mb + 4 (l / 16) + r, r = 0..3, of position 16 w + l % 16
typedef float dct_floatx4 __attribute__((ext_vector_type(4)));
constexpr int kDctMaxMB = 3, kDctKS = 32;  // up to 48 rows (MB blocks of 16), 128 taps
// The kernel for input stored position-major, x[o, i, n] (the layout the generic mel kernel
// writes for the MFCC path: a tile of 64 positions is one contiguous run of 64 * n_in floats).
// Needs n_in % 4 == 0.  The slab is [position][n_in + 4]: whole float4 stores when staging, and
// the B operand reads (16 positions x 4 taps per wave instruction) are 2-way at most.
template <int kDctMB>
__global__ __launch_bounds__(256) void dct2_mfma_fm_kernel(const float* __restrict__ x, int n_in,
                                                            long long inner,
                                                            const float* __restrict__ basis, int n_out,
                                                            float* __restrict__ out,
                                                            const int* __restrict__ keys,
                                                            long long key_stride, float top,
                                                            long long tiles_per_row, long long total,
                                                            int tiles_per_wg) {
  constexpr int PITCH = 4 * kDctKS + 4;
  __shared__ __attribute__((aligned(16))) float slab[64 * PITCH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = lane & 15, kq = lane >> 4;
  float a[kDctMB][kDctKS];
#pragma unroll
  for (int mb = 0; mb < kDctMB; ++mb) {
#pragma unroll
    for (int ks = 0; ks < kDctKS; ++ks) {
      const int m = 16 * mb + row, k = 4 * ks + kq;
      a[mb][ks] = (m < n_out && k < n_in) ? basis[(long long)m * n_in + k] : 0.0f;
    }
  }
  const long long first = (long long)blockIdx.x * tiles_per_wg;
  const long long last = first + tiles_per_wg < total ? first + tiles_per_wg : total;
  const int q4 = n_in / 4;              // float4 pieces per position
  float4 nx[8];                         // 64 positions x <= 32 pieces = 2048 pieces / 256 threads
  int nkey = 0;
  auto fetch = [&](long long idx_n) {
    const long long o = idx_n / tiles_per_row;
    if (keys && idx_n < last) nkey = keys[o * key_stride];
    const long long i0 = (idx_n - o * tiles_per_row) * 64;
    const long long left = idx_n < last ? inner - i0 : 0;   // positions of the tile that exist
    const float4* __restrict__ src = reinterpret_cast<const float4*>(x + (o * inner + i0) * n_in);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = j * 256 + tid;
      nx[j] = (idx < 64 * q4 && idx / q4 < left) ? src[idx] : float4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  // zero the slab once: taps n_in .. 4 kDctKS - 1 stay zero (their A operands are zero too, but
  // 0 * NaN of stale LDS contents would not be)
  for (int i = tid; i < 64 * PITCH; i += 256) slab[i] = 0.0f;
  fetch(first);
  __syncthreads();
  for (long long idx = first; idx < last; ++idx) {
    const long long o = idx / tiles_per_row;
    const long long i0 = (idx - o * tiles_per_row) * 64;
    const float thr = keys ? floor_from_key(nkey, MIA_EPI_DB, top) : -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx4 = j * 256 + tid;
      if (idx4 < 64 * q4) {
        const int pos = idx4 / q4, n4 = idx4 - pos * q4;
        const float4 v = {floor_nan(nx[j].x, thr), floor_nan(nx[j].y, thr), floor_nan(nx[j].z, thr),
                          floor_nan(nx[j].w, thr)};
        *reinterpret_cast<float4*>(slab + pos * PITCH + 4 * n4) = v;
      }
    }
    __syncthreads();
    fetch(idx + 1);
    float bv[kDctKS];
#pragma unroll
    for (int ks = 0; ks < kDctKS; ++ks) bv[ks] = slab[(16 * wave + row) * PITCH + 4 * ks + kq];
    dct_floatx4 acc[kDctMB];
#pragma unroll
    for (int mb = 0; mb < kDctMB; ++mb) acc[mb] = dct_floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ks = 0; ks < kDctKS; ++ks) {
#pragma unroll
      for (int mb = 0; mb < kDctMB; ++mb) {
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][ks], bv[ks], acc[mb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the prefetch, before this tile's stores
    const long long pos = i0 + 16 * wave + row;
    if (pos < inner) {
#pragma unroll
      for (int mb = 0; mb < kDctMB; ++mb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * mb + 4 * kq + r;
          if (m < n_out) out[(o * n_out + m) * inner + pos] = acc[mb][r];
        }
      }
    }
    __syncthreads();  // the next tile overwrites the slab
  }
}

// Workgroups of a matrix-core DCT kernel that one CU holds (by its registers and its LDS slab),
// asked once per instantiation and device.
static int dct_resident_per_cu(const void* kern, int which) {
  static int cached[kDctMaxMB][16] = {{0}};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (cached[which][dev] == 0) {
    int per_cu = 2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) {
      per_cu = 2;
    }
    cached[which][dev] = per_cu;
  }
  return cached[which][dev];
}

// Row-major input without a slab: a wave owns 64 positions of one `o`, lane = position.  Every
// row piece is one coalesced 256-byte load, sixteen of them in flight per wave; the basis values
// are the same in every lane and come through the scalar cache (eight consecutive taps of one
// coefficient per s_load_dwordx8); NK accumulators per lane.  Arithmetic and its order equal the
// matrix-core kernels' (fma chain over ascending taps).  Few registers, no LDS, no barrier: five
// to eight waves per SIMD keep 100+ KB of loads in flight per CU, which the slab kernels with
// their 2 - 3 workgroups per CU cannot.
template <int NK>
__global__ __launch_bounds__(256) void dct2_stream_kernel(const float* __restrict__ x, int n_in, long long inner,
                                                           const float* __restrict__ basis, int n_out,
                                                           float* __restrict__ out, const int* __restrict__ keys,
                                                           long long key_stride, float top, long long tiles_per_row,
                                                           long long total) {
  constexpr int CH = 16;
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long step = (long long)gridDim.x * 4;
  for (long long tile = gw; tile < total; tile += step) {
    const long long o = tile / tiles_per_row;
    const long long pos = (tile - o * tiles_per_row) * 64 + lane;
    const bool ok = pos < inner;
    const float thr = keys ? floor_from_key(keys[o * key_stride], MIA_EPI_DB, top) : -INFINITY;
    const float* __restrict__ xp = x + (o * n_in) * inner + (ok ? pos : 0);
    float acc[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) acc[k] = 0.0f;
    float cur[CH], nxt[CH];
    auto fetch = [&](float (&dst)[CH], int n0) {
#pragma unroll
      for (int j = 0; j < CH; ++j) dst[j] = (ok && n0 + j < n_in) ? xp[(long long)(n0 + j) * inner] : 0.0f;
    };
    fetch(cur, 0);
    for (int n0 = 0; n0 < n_in; n0 += CH) {
      if (n0 + CH < n_in) fetch(nxt, n0 + CH);
      float v[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] = floor_nan(cur[j], thr);
      if (n0 + CH <= n_in) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          if (k < n_out) {
            const float* __restrict__ brow = basis + (long long)k * n_in + n0;
#pragma unroll
            for (int j = 0; j < CH; ++j) acc[k] = fmaf(brow[j], v[j], acc[k]);
          }
        }
      } else {  // the last, partial chunk
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          if (k < n_out) {
            for (int j = 0; n0 + j < n_in; ++j) acc[k] = fmaf(basis[(long long)k * n_in + n0 + j], v[j], acc[k]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) cur[j] = nxt[j];
    }
    if (ok) {
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        if (k < n_out) out[(o * n_out + k) * inner + pos] = acc[k];
      }
    }
  }
}

bool dct2_frame_major_ok(int n_in, int n_out) {
  return n_in <= 4 * kDctKS && n_out <= 16 * kDctMaxMB && n_in >= 32 && n_in % 4 == 0;
}

int launch_dct2_frame_major(const float* x, long long outer, int n_in, long long inner, const float* basis,
                            int n_out, float* out, const int* keys, long long key_stride, float top,
                            hipStream_t s) {
  if (outer <= 0 || inner <= 0 || n_out <= 0) return MIA_OK;
  if (!dct2_frame_major_ok(n_in, n_out)) {
    set_error("dct: position-major input needs 32 <= n_in <= 128, n_in %% 4 == 0, n_out <= 48");
    return MIA_ERR_UNSUPPORTED;
  }
  const long long tiles_per_row = (inner + 63) / 64;
  const long long total = outer * tiles_per_row;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  }
  // one instantiation per number of 16-row blocks of the basis: fewer blocks, fewer resident
  // operand registers and matrix instructions (13 or 20 coefficients instead of 40)
  const int mb = n_out <= 16 ? 1 : (n_out <= 32 ? 2 : 3);
  auto kern = mb == 1 ? dct2_mfma_fm_kernel<1> : (mb == 2 ? dct2_mfma_fm_kernel<2> : dct2_mfma_fm_kernel<3>);
  long long grid = (long long)cus * dct_resident_per_cu(reinterpret_cast<const void*>(kern), mb - 1);
  if (grid > total) grid = total;
  const long long per_wg = (total + grid - 1) / grid;
  if (per_wg > 0x7fffffffLL) {
    set_error("dct: batch too large for one launch");
    return MIA_ERR_UNSUPPORTED;
  }
  grid = (total + per_wg - 1) / per_wg;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, s, x, n_in, inner, basis,
                     n_out, out, keys, key_stride, top, tiles_per_row, total, (int)per_wg);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int launch_dct2(const float* x, long long outer, int n_in, long long inner, const float* basis,
                int n_out, float* out, const int* keys, long long key_stride, float top,
                hipStream_t s) {
  if (outer <= 0 || inner <= 0 || n_out <= 0) return MIA_OK;
  if (n_out <= 48) {
    const long long tiles_per_row = (inner + 63) / 64;
    const long long total = outer * tiles_per_row;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    }
    auto kern = n_out <= 16 ? dct2_stream_kernel<16>
                            : (n_out <= 24 ? dct2_stream_kernel<24>
                                           : (n_out <= 32 ? dct2_stream_kernel<32>
                                                          : (n_out <= 40 ? dct2_stream_kernel<40> : dct2_stream_kernel<48>)));
    long long grid = (long long)cus * 8;  // 32 waves per CU at most; a wave walks tiles grid-strided
    if (grid * 4 > total) grid = (total + 3) / 4;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, s, x, n_in, inner, basis, n_out, out, keys,
                       key_stride, top, tiles_per_row, total);
    MIA_CHECK_HIP(hipGetLastError());
    return MIA_OK;
  }
  const size_t lds = sizeof(float) * 64 * (size_t)n_in;
  if (lds > 160 * 1024) {
    set_error("dct: n_in=%d too large for the LDS slab", n_in);
    return MIA_ERR_UNSUPPORTED;
  }
  auto kern = dct2_kernel<5>;
  if (lds > 64 * 1024) {
    MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  dim3 grid((unsigned)((inner + 63) / 64), (unsigned)outer);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, n_in, inner, basis, n_out, out, keys,
                     key_stride, top);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

// ---------------------------------------------------------------------------
// |z|, arg z
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void magnitude_kernel(const float2* __restrict__ z,
                                                         long long n, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float2 v = z[i];
    out[i] = hypotf(v.x, v.y);
  }
}
__global__ __launch_bounds__(256) void phase_kernel(const float2* __restrict__ z, long long n,
                                                     float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float2 v = z[i];
    out[i] = atan2f(v.y, v.x);
  }
}

static unsigned stream_grid(long long n) {
  long long g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  return (unsigned)(g < 1 ? 1 : g);
}

int launch_pad(const float* x, long long B, long long T, long long pad, int mode, float* out,
               hipStream_t s) {
  const long long L = T + 2 * pad;
  if (B <= 0 || L <= 0) return MIA_OK;
  dim3 grid((unsigned)((L + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(pad_kernel, grid, dim3(256), 0, s, x, T, pad, mode, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int launch_frame(const float* x, long long B, long long T, long long F, int L, int hop,
                 float* out, hipStream_t s) {
  if (B <= 0 || F <= 0) return MIA_OK;
  dim3 grid((unsigned)((F * L + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(frame_kernel, grid, dim3(256), 0, s, x, T, F, L, hop, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int launch_magnitude(const float2* z, long long n, float* out, hipStream_t s) {
  if (n <= 0) return MIA_OK;
  hipLaunchKernelGGL(magnitude_kernel, dim3(stream_grid(n)), dim3(256), 0, s, z, n, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}
int launch_phase(const float2* z, long long n, float* out, hipStream_t s) {
  if (n <= 0) return MIA_OK;
  hipLaunchKernelGGL(phase_kernel, dim3(stream_grid(n)), dim3(256), 0, s, z, n, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // namespace mia

// Whole-signal Fourier resampling: rfft of a row of any length n_in, copy / truncate the
// spectrum to n_out/2+1 bins, irfft of length n_out (what scipy.signal.resample does for real
// input, which is what the reference's resample(res_type="fft") calls).
//
// Two paths.  Even lengths whose halves factor into 2, 3, 5 and 7 (every whole number of
// seconds or milliseconds at the usual sample rates) take the direct one: the row is read in
// place as n_in/2 complex numbers and goes through mixed-radix Stockham passes; the real-input
// split, the spectrum resize and the packing for the half-length inverse are one kernel.
// Every other length is transformed as a chirp-z (Bluestein) convolution:
//   DFT_N(x)[k] = conj(c[k]) * sum_n x[n] conj(c[n]) c[k-n],   c[n] = exp(i pi n^2 / N),
// through power-of-two FFTs of length L >= 2N-1, with the phases reduced in integers
// (n^2 mod 2N) before they become angles, so the chirps are exact to float rounding.
//
// Both use register-resident Stockham passes over global memory: one thread owns one radix-R
// butterfly, reads its R inputs L/R apart (consecutive threads read consecutive elements),
// multiplies by twiddles it derives from one double-precision sincospi and a double recurrence,
// and writes the R outputs of the autosort step; no LDS, no transposes, no twiddle tables.  An
// inverse transform is the same machine applied to the conjugate spectrum.  Long rows of the
// direct path whose length splits into two composite radices of at most 1024 take two
// "super-passes" instead (stockham_super_kernel: the same pass to memory, its radix-R transform
// done in LDS by up to three small butterflies), which halves the trips through memory.
#include "fft_engine.h"
#include "mia_internal.h"

// Butterfly sizes tried, largest first, when a length is split into passes.
#ifndef MIA_RADIX_CANDIDATES
#define MIA_RADIX_CANDIDATES 36, 35, 32, 30, 28, 27, 25, 24, 21, 20, 18, 16, 15, 14, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2
#endif

namespace mia {

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
  return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// One Stockham pass of radix R over `rows` rows of length L = 1 << log2L; Ns = product of the
// radices of the passes before it.
template <int R, int LOG2R>
__global__ __launch_bounds__(256) void stockham_pass_kernel(const float2* __restrict__ in,
                                                            float2* __restrict__ out, int log2L,
                                                            long long rows, int Ns) {
  const int log2Lr = log2L - LOG2R;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (rows << log2Lr)) return;
  const long long row = g >> log2Lr;
  const long long Lr = 1LL << log2Lr;
  const long long j = g & (Lr - 1);
  const float2* __restrict__ src = in + (row << log2L) + j;
  float2 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = src[(long long)r * Lr];
  const long long m = j & (long long)(Ns - 1);
  if (Ns > 1) {
    double s1, c1;
    sincospi(-2.0 * (double)m / ((double)Ns * (double)R), &s1, &c1);
    double wr = c1, wi = s1;
#pragma unroll
    for (int r = 1; r < R; ++r) {
      v[r] = cmulf(v[r], float2{(float)wr, (float)wi});
      const double nr = wr * c1 - wi * s1;
      wi = wr * s1 + wi * c1;
      wr = nr;
    }
  }
  Dft<R>::run(v);
  float2* __restrict__ dst = out + (row << log2L) + (j - m) * R + m;
#pragma unroll
  for (int k = 0; k < R; ++k) dst[(long long)k * Ns] = v[k];
}

// c[n] = exp(i pi n^2 / N), n < N.
__global__ void chirp_kernel(long long N, float2* __restrict__ c) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const long long q = (n * n) % (2 * N);
  double s, co;
  sincospi((double)q / (double)N, &s, &co);
  c[n] = {(float)co, (float)s};
}

// The convolution kernel of length L: b[m] = b[L-m] = c[m] for m < N, zero between.
__global__ void chirp_filter_kernel(const float2* __restrict__ c, long long N, long long L,
                                    float2* __restrict__ b) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  float2 v = {0.0f, 0.0f};
  if (i < N) v = c[i];
  else if (L - i < N) v = c[L - i];
  b[i] = v;
}

// a[row, n] = x[row, n] * conj(c[n]) for n < N, zero up to L.
__global__ void chirp_in_real_kernel(const float* __restrict__ x, const float2* __restrict__ c,
                                     long long N, int log2L, long long rows,
                                     float2* __restrict__ a) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (rows << log2L)) return;
  const long long row = g >> log2L, n = g & ((1LL << log2L) - 1);
  float2 v = {0.0f, 0.0f};
  if (n < N) {
    const float xv = x[row * N + n];
    const float2 cn = c[n];
    v = {xv * cn.x, -xv * cn.y};
  }
  a[g] = v;
}

// a[row, i] = conj(a[row, i] * B[i]): the spectrum product, conjugated so that a forward
// transform of it yields the conjugate of the (unnormalised) inverse transform.
__global__ void spectrum_product_kernel(float2* __restrict__ a, const float2* __restrict__ B,
                                        int log2L, long long rows) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (rows << log2L)) return;
  const float2 p = cmulf(a[g], B[g & ((1LL << log2L) - 1)]);
  a[g] = {p.x, -p.y};
}

// From the first convolution to the input of the second.  r1[row, k] holds the conjugate of
// L1 * (convolution), so X[k] = conj(cN[k] * r1[k]) / L1 is bin k of the length-N transform.
// The output spectrum keeps bins below nyq, scales the shared Nyquist bin, is extended to all
// M bins by Hermitian symmetry (imaginary parts of DC and of an even M's Nyquist bin are
// dropped, as a complex-to-real transform ignores them), conjugated (inverse transform as a
// forward one) and multiplied by conj(cM[k]): the chirp-premultiplied input of the second
// convolution, zero up to L2.
__global__ void resize_spectrum_kernel(const float2* __restrict__ r1, const float2* __restrict__ cN,
                                       const float2* __restrict__ cM, long long N, long long M,
                                       int log2L1, int log2L2, long long rows, long long nyq,
                                       long long shared_bin, float shared_scale, float inv_L1,
                                       float2* __restrict__ a2) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (rows << log2L2)) return;
  const long long row = g >> log2L2, k = g & ((1LL << log2L2) - 1);
  float2 v = {0.0f, 0.0f};
  if (k < M) {
    const bool neg = 2 * k > M;
    const long long kk = neg ? M - k : k;
    if (kk < nyq) {
      const float2 p = cmulf(cN[kk], r1[(row << log2L1) + kk]);
      float2 X = {p.x * inv_L1, -p.y * inv_L1};
      if (kk == shared_bin) X = {X.x * shared_scale, X.y * shared_scale};
      if (kk == 0 || 2 * kk == M) X.y = 0.0f;
      // Yfull = neg ? conj(X) : X;  Z = conj(Yfull)
      const float2 Z = {X.x, neg ? X.y : -X.y};
      const float2 cm = cM[k];
      v = cmulf(Z, float2{cm.x, -cm.y});
    }
  }
  a2[g] = v;
}

// y[row, n] = Re(cM[n] * r2[row, n]) * scale, n < M.
__global__ void chirp_out_real_kernel(const float2* __restrict__ r2, const float2* __restrict__ cM,
                                      long long M, int log2L2, long long rows, float scale,
                                      float* __restrict__ y) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * M) return;
  const long long row = g / M, n = g - row * M;
  const float2 r = r2[(row << log2L2) + n];
  const float2 cm = cM[n];
  y[g] = (cm.x * r.x - cm.y * r.y) * scale;
}

// ---------------------------------------------------------------------------
// Direct path for even lengths whose halves factor into 2, 3, 5 and 7: the row, read as
// N/2 complex numbers z[n] = x[2n] + i x[2n+1], goes through Stockham passes of mixed radix
// and the real-input split is folded into the kernel that resizes the spectrum.
// ---------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void stockham_any_kernel(const float2* __restrict__ in,
                                                           float2* __restrict__ out, unsigned L,
                                                           unsigned Ns, long long rows) {
  const unsigned Lr = L / R;
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Lr) return;
  const unsigned m = j % Ns;
  double s1 = 0.0, c1 = 1.0;
  if (Ns > 1) sincospi(-2.0 * (double)m / ((double)Ns * (double)R), &s1, &c1);
  const size_t base = (size_t)(j - m) * R + m;
  for (long long row = blockIdx.y; row < rows; row += gridDim.y) {
    const float2* __restrict__ src = in + (size_t)row * L + j;
    float2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = src[(size_t)r * Lr];
    if (Ns > 1) {
      double wr = c1, wi = s1;
#pragma unroll
      for (int r = 1; r < R; ++r) {
        v[r] = cmulf(v[r], float2{(float)wr, (float)wi});
        const double nr = wr * c1 - wi * s1;
        wi = wr * s1 + wi * c1;
        wr = nr;
      }
    }
    Dft<R>::run(v);
    float2* __restrict__ dst = out + (size_t)row * L + base;
#pragma unroll
    for (int k = 0; k < R; ++k) dst[(size_t)k * Ns] = v[k];
  }
}

// ---------------------------------------------------------------------------
// Super-passes: one Stockham pass of a composite radix R = r0 r1 r2 (each factor a butterfly
// of at most 16 points, R <= 1024) whose inner structure runs in LDS, so that a row of
// 661 500 points takes two trips through memory instead of four.  To the outside a super-pass
// is exactly the radix-R pass of stockham_any_kernel: super-butterfly u < D = L / R reads
// in[u + c D], c < R, multiplies by exp(-2 pi i m c / (Ns R)), m = u mod Ns, and writes
// X[k] to out[(u - m) R + m + k Ns].  A workgroup takes kSuperW adjacent super-butterflies
// (their elements are adjacent in memory on the way in and, past the first pass, on the way
// out) and keeps them interleaved in LDS, element c of sequence w at [c kSuperW + w]; the
// R-point transforms are in-place decimation steps (each butterfly reads and writes its own
// slots, a barrier between steps), which leave X[k0 + r0 k1 + r0 r1 k2] at slot
// k0 r1 r2 + k1 r2 + k2; the store phase reads the slots in that order.  Inner twiddles come
// from an LDS table of exp(-2 pi i j / R) the workgroup builds once (double-precision
// sincospi); the outer ones from a double recurrence per thread, as in the single passes.
// ---------------------------------------------------------------------------
constexpr int kSuperW = 16;          // adjacent super-butterflies per workgroup (128-byte runs)
constexpr int kSuperMaxR = 1024;     // LDS: R (kSuperW + 1) complex numbers
constexpr int kSuperThreads = 512;
constexpr int kSuperBatch = 8;       // loads a thread has in flight before it uses the first
constexpr unsigned long long kSuperGrid = 1024;  // workgroups per launch, about: each walks rows

struct SuperPass {
  int R;      // r[0] * r[1] * r[2]
  int r[3];   // unused factors are 1
  // ceil(2^32 / d) for d = r[0], r[1], r[1] * r[2], r[2], R: __umulhi(x, magic) == x / d for
  // x < 65536 and those d (x d < 2^32), which covers every index in the kernel
  unsigned magic_r0, magic_r1, magic_s0, magic_s1, magic_R;
};
static unsigned super_magic(int d) { return (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }

// One in-place decimation step of radix R_ over the interleaved sequences: butterflies j <
// nbf per sequence, elements S apart (magic_S divides by S); outputs k >= 1 are multiplied by
// table[lo * k * mul].
template <int R_>
__device__ __forceinline__ void super_step(float2* __restrict__ buf, const float2* __restrict__ table,
                                           int nbf, int S, unsigned magic_S, int mul) {
  const int w = threadIdx.x % kSuperW;
  for (int j = threadIdx.x / kSuperW; j < nbf; j += kSuperThreads / kSuperW) {
    const int hi = S > 1 ? (int)__umulhi((unsigned)j, magic_S) : j;
    const int lo = j - hi * S;
    float2* __restrict__ e = buf + (size_t)(hi * (R_ * S) + lo) * kSuperW + w;
    float2 v[R_];
#pragma unroll
    for (int q = 0; q < R_; ++q) v[q] = e[(size_t)q * S * kSuperW];
    Dft<R_>::run(v);
    if (S > 1) {
      const int step = lo * mul;
#pragma unroll
      for (int k = 1; k < R_; ++k) v[k] = cmulf(v[k], table[step * k]);
    }
#pragma unroll
    for (int q = 0; q < R_; ++q) e[(size_t)q * S * kSuperW] = v[q];
  }
}

__device__ __forceinline__ void super_step_any(int radix, float2* __restrict__ buf,
                                               const float2* __restrict__ table, int nbf, int S,
                                               unsigned magic_S, int mul) {
  switch (radix) {
#define MIA_STEP(RR) case RR: super_step<RR>(buf, table, nbf, S, magic_S, mul); break;
    MIA_STEP(16) MIA_STEP(15) MIA_STEP(14) MIA_STEP(12) MIA_STEP(10) MIA_STEP(9) MIA_STEP(8)
    MIA_STEP(7) MIA_STEP(6) MIA_STEP(5) MIA_STEP(4) MIA_STEP(3) MIA_STEP(2)
#undef MIA_STEP
    default: break;
  }
}

// Bin kk (0 <= kk <= Nh) of the length-N real transform from the length-Nh transform Z of the
// packed row: X = E + w^kk O, E = (Z[kk] + conj Z[Nh-kk])/2, O = -i (Z[kk] - conj Z[Nh-kk])/2.
__device__ __forceinline__ float2 real_bin(const float2* __restrict__ Z, long long Nh, long long kk) {
  const float2 a = Z[kk == Nh ? 0 : kk];
  const float2 bz = Z[kk == 0 ? 0 : Nh - kk];
  const float2 E = {0.5f * (a.x + bz.x), 0.5f * (a.y - bz.y)};
  const float2 O = {0.5f * (a.y + bz.y), -0.5f * (a.x - bz.x)};
  // exp(-2 pi i kk / N), N = 2 Nh: the quotient in double, its sine and cosine in float (the
  // quotient's rounding to float moves the angle by at most 1e-7 rad, one float rounding of the
  // twiddle; the double-precision sincospi was most of this kernel's time)
  float s, c;
  sincospif((float)(-(double)kk / (double)Nh), &s, &c);
  const float2 wO = cmulf(O, float2{c, s});
  return {E.x + wO.x, E.y + wO.y};
}

// The spectrum of a resampled row as input of its packed inverse transform.  Element k < Mh: the
// two output-spectrum bins k and Mh-k (copied from the input spectrum below nyq, the shared
// Nyquist bin scaled, zero above), combined into bin k of the packed inverse transform,
// Z'[k] = (A + conj B) + i exp(2 pi i k / M) (A - conj B), conjugated (the inverse transform runs
// as a forward one).
struct ResizeIn {
  const float2* Z;  // (rows, Nh): forward transform of the packed input rows
  long long Nh, Mh, nyq, shared_bin;
  float shared_scale;
};
__device__ __forceinline__ float2 resized_element(const ResizeIn& r, long long row, long long k) {
  const float2* __restrict__ Z = r.Z + row * r.Nh;
  auto bin = [&](long long kk) {
    float2 X = {0.0f, 0.0f};
    if (kk < r.nyq) {
      X = real_bin(Z, r.Nh, kk);
      if (kk == r.shared_bin) X = {X.x * r.shared_scale, X.y * r.shared_scale};
      if (kk == 0 || kk == r.Mh) X.y = 0.0f;
    }
    return X;
  };
  const float2 A = bin(k);
  const float2 B = bin(r.Mh - k);
  const float2 S = {A.x + B.x, A.y - B.y};  // A + conj B
  const float2 D = {A.x - B.x, A.y + B.y};  // A - conj B
  float s, c;
  sincospif((float)((double)k / (double)r.Mh), &s, &c);  // exp(+2 pi i k / M), M = 2 Mh
  const float2 wD = cmulf(D, float2{c, s});
  // Z' = S + i wD;  conj(Z')
  return {S.x - wD.y, -(S.y + wD.x)};
}

// The elements c0, c0 + CSTEP, ... < R of one sequence, D apart, into registers.
template <int PER, int CSTEP>
__device__ __forceinline__ void super_fetch(float2 (&v)[PER], const float2* __restrict__ src, bool live,
                                            int c0, int R, unsigned D) {
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = c0 + i * CSTEP;
    v[i] = (live && c < R) ? src[(size_t)c * D] : float2{0.0f, 0.0f};
  }
}

// out_scale != 0: the pass (the last of an inverse transform) stores conj(result) * out_scale - the
// real output row, two samples per element.
__global__ __launch_bounds__(kSuperThreads, 1) void stockham_super_kernel(
    const float2* __restrict__ in, float2* __restrict__ out, unsigned L, unsigned Ns, long long rows,
    SuperPass sp, unsigned blocks_u, unsigned per_xcd, unsigned row_slots, float out_scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char super_smem[];
  float2* buf = reinterpret_cast<float2*>(super_smem);   // [R][kSuperW]
  float2* table = buf + (size_t)sp.R * kSuperW;          // [R]
  const int R = sp.R, r0 = sp.r[0], r1 = sp.r[1], r2 = sp.r[2];
  const unsigned D = L / (unsigned)R;
  // Workgroups are dealt to the eight XCDs in turn, each with its own L2.  The 128-byte runs
  // of neighbouring blocks of super-butterflies share cache lines (rows and strides are not
  // multiples of a line), so XCD x takes the contiguous blocks [x per_xcd, (x + 1) per_xcd) of
  // a row, one after the other: the shared lines are fetched, and the partial lines written,
  // within one L2.
  const unsigned xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;
  const unsigned row_slot = seq / per_xcd;
  const unsigned block_u = xcd * per_xcd + (seq - row_slot * per_xcd);
  if (block_u >= blocks_u) return;
  const unsigned u0 = block_u * kSuperW;
  const int tid = threadIdx.x;
  for (int j = tid; j < R; j += kSuperThreads) {
    double sn, cs;
    sincospi(-2.0 * (double)j / (double)R, &sn, &cs);
    table[j] = {(float)cs, (float)sn};
  }
  // load phase: thread = (sequence w, elements c = c0, c0 + cstep, ...)
  const int w = tid % kSuperW, c0 = tid / kSuperW;
  constexpr int cstep = kSuperThreads / kSuperW;
  constexpr int kPer = kSuperMaxR / cstep;  // elements per thread at most
  const unsigned u = u0 + (unsigned)w;
  const bool live = u < D;
  const unsigned m = live ? u % Ns : 0u;
  double wr0 = 1.0, wi0 = 0.0, sr = 1.0, si = 0.0;  // exp(-2 pi i m c0 / (Ns R)) and the step by cstep
  if (Ns > 1) {
    const double den = (double)Ns * (double)R;
    sincospi(-2.0 * ((double)m * (double)c0) / den, &wi0, &wr0);
    sincospi(-2.0 * ((double)m * (double)cstep) / den, &si, &sr);
  }
  // store phase: past the first pass the kSuperW sequences are adjacent in memory for every k;
  // in the first pass (Ns == 1) a sequence's R outputs are, so there the lanes run along k
  const bool along_k = Ns < (unsigned)kSuperW;
  for (long long row = row_slot; row < rows; row += row_slots) {
    float2* __restrict__ dst = out + (size_t)row * L;
    __syncthreads();  // the previous row's store phase is done with buf (and the table is built)
    {
      const float2* __restrict__ src = in + (size_t)row * L + u;
      double wr = wr0, wi = wi0;
      constexpr int kBatch = kPer < kSuperBatch ? kPer : kSuperBatch;
      for (int cb = c0; cb < R; cb += cstep * kBatch) {
        float2 v[kBatch];
        super_fetch<kBatch, cstep>(v, src, live, cb, R, D);
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
          const int c = cb + i * cstep;
          if (c < R) {
            float2 x = v[i];
            if (Ns > 1) x = cmulf(x, float2{(float)wr, (float)wi});
            buf[(size_t)c * kSuperW + w] = x;
          }
          const double nr = wr * sr - wi * si;
          wi = wr * si + wi * sr;
          wr = nr;
        }
      }
    }
    __syncthreads();
    super_step_any(r0, buf, table, R / r0, r1 * r2, sp.magic_s0, 1);
    if (r1 > 1) {
      __syncthreads();
      super_step_any(r1, buf, table, R / r1, r2, sp.magic_s1, r0);
    }
    if (r2 > 1) {
      __syncthreads();
      super_step_any(r2, buf, table, R / r2, 1, 0u, 0);
    }
    __syncthreads();
    const int total = R * kSuperW;
    for (int idx = tid; idx < total; idx += kSuperThreads) {
      int ws, k;
      if (along_k) {
        ws = (int)__umulhi((unsigned)idx, sp.magic_R);
        k = idx - ws * R;
      } else {
        k = idx / kSuperW;
        ws = idx - k * kSuperW;
      }
      const unsigned us = u0 + (unsigned)ws;
      if (us >= D) continue;
      const int t = (int)__umulhi((unsigned)k, sp.magic_r0), k0 = k - t * r0;
      const int k2 = (int)__umulhi((unsigned)t, sp.magic_r1), k1 = t - k2 * r1;
      const int slot = (k0 * r1 + k1) * r2 + k2;
      // not along k: ws is this thread's own w
      const unsigned ms = along_k ? (Ns == 1 ? 0u : us % Ns) : m;
      float2 res = buf[(size_t)slot * kSuperW + ws];
      if (out_scale != 0.0f) res = {res.x * out_scale, -res.y * out_scale};
      dst[(size_t)(us - ms) * R + ms + (size_t)k * Ns] = res;
    }
  }
}

// One thread per element of the packed inverse transform's input (the plans whose first inverse
// pass does not build its input itself).
__global__ void resize_packed_kernel(ResizeIn r, long long rows, float2* __restrict__ Zout) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * r.Mh) return;
  const long long row = g / r.Mh, k = g - row * r.Mh;
  Zout[g] = resized_element(r, row, k);
}

// y[2n] + i y[2n+1] = conj(F[n]) * scale.
__global__ void unpack_real_kernel(const float2* __restrict__ F, long long total, float scale,
                                   float2* __restrict__ y) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const float2 f = F[g];
  y[g] = {f.x * scale, -f.y * scale};
}

// Radices of a mixed-radix plan for length L: the fewest passes the available butterflies
// allow and, among those, the smallest butterflies (registers); largest first.  Returns the
// number of passes, 0 if L has a prime factor other than 2, 3, 5, 7.
static int plan_search(long long L, const int* cand, int n_cand, int first, int depth, int limit,
                       int* cur, int* best, int* best_sum, int sum) {
  if (L == 1) {
    if (depth < limit || (depth == limit && sum < *best_sum)) {
      for (int i = 0; i < depth; ++i) best[i] = cur[i];
      *best_sum = sum;
      return depth;
    }
    return limit;
  }
  if (depth >= limit) return limit;  // cannot beat the best plan found so far
  for (int i = first; i < n_cand; ++i) {
    if (L % cand[i]) continue;
    cur[depth] = cand[i];
    limit = plan_search(L / cand[i], cand, n_cand, i, depth + 1, limit, cur, best, best_sum,
                        sum + cand[i]);
  }
  return limit;
}
static int plan_radices(long long L, int* radices) {
  static const int cand[] = {MIA_RADIX_CANDIDATES};
  const int n_cand = (int)(sizeof(cand) / sizeof(cand[0]));
  long long rest = L;
  for (int p : {2, 3, 5, 7}) {
    while (rest % p == 0) rest /= p;
  }
  if (rest != 1 || L < 2) return 0;
  int cur[32], best_sum = 1 << 30;
  int limit = 28;  // 2^27 = 28 passes of radix 2 at worst
  for (int i = 0; i < 32; ++i) radices[i] = 0;
  limit = plan_search(L, cand, n_cand, 0, 0, limit + 1, cur, radices, &best_sum, 0);
  return radices[0] ? limit : 0;
}
// Super-pass plan of length L: the fewest radices R_i <= kSuperMaxR, each a product of at most
// three butterflies of at most 16 points and leaving at least kSuperW super-butterflies per row
// (L / R_i >= kSuperW), that multiply to L; among those the plan whose largest radix is
// smallest (LDS per workgroup).  Returns the number of passes, 0 if there is none.
#ifndef MIA_FFT_SUPERPASS
#define MIA_FFT_SUPERPASS 1
#endif
static const int kSuperSub[] = {16, 15, 14, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2};
static bool super_split(int R, SuperPass* sp) {
  const int n = (int)(sizeof(kSuperSub) / sizeof(kSuperSub[0]));
  auto is_sub = [&](int v) {
    for (int i = 0; i < n; ++i) {
      if (kSuperSub[i] == v) return true;
    }
    return false;
  };
  sp->R = R;
  if (is_sub(R)) {
    sp->r[0] = R;
    sp->r[1] = sp->r[2] = 1;
    return true;
  }
  for (int i = 0; i < n; ++i) {  // two factors, the larger first
    const int a = kSuperSub[i];
    if (R % a == 0 && R / a <= a && is_sub(R / a)) {
      sp->r[0] = a;
      sp->r[1] = R / a;
      sp->r[2] = 1;
      return true;
    }
  }
  for (int i = 0; i < n; ++i) {
    for (int j = i; j < n; ++j) {
      const int a = kSuperSub[i], b = kSuperSub[j];
      if (R % (a * b) == 0 && R / (a * b) <= b && is_sub(R / (a * b))) {
        sp->r[0] = a;
        sp->r[1] = b;
        sp->r[2] = R / (a * b);
        return true;
      }
    }
  }
  return false;
}
static void super_search(long long rest, const int* cand, int n_cand, int first, int depth, int* cur,
                         int* best, int* best_n, int* best_max, int cur_max) {
  if (rest == 1) {
    if (depth < *best_n || (depth == *best_n && cur_max < *best_max)) {
      for (int i = 0; i < depth; ++i) best[i] = cur[i];
      *best_n = depth;
      *best_max = cur_max;
    }
    return;
  }
  if (depth + 1 > *best_n) return;
  for (int i = first; i < n_cand; ++i) {
    if (rest % cand[i]) continue;
    cur[depth] = cand[i];
    super_search(rest / cand[i], cand, n_cand, i, depth + 1, cur, best, best_n, best_max,
                 cand[i] > cur_max ? cand[i] : cur_max);
  }
}
static int plan_super(long long L, SuperPass* passes, int max_passes) {
  if (!MIA_FFT_SUPERPASS || L < 2LL * kSuperW) return 0;
  int cand[512], n_cand = 0;
  const long long cap = L / kSuperW < kSuperMaxR ? L / kSuperW : kSuperMaxR;
  for (int R = (int)cap; R >= 2 && n_cand < 512; --R) {
    SuperPass sp;
    if (L % R == 0 && super_split(R, &sp)) cand[n_cand++] = R;
  }
  int cur[8], best[8], best_n = max_passes + 1, best_max = 1 << 30;
  super_search(L, cand, n_cand, 0, 0, cur, best, &best_n, &best_max, 0);
  if (best_n > max_passes) return 0;
  for (int i = 0; i < best_n; ++i) {
    SuperPass& q = passes[i];
    super_split(best[i], &q);
    q.magic_r0 = super_magic(q.r[0]);
    q.magic_r1 = super_magic(q.r[1]);
    q.magic_s0 = super_magic(q.r[1] * q.r[2]);
    q.magic_s1 = super_magic(q.r[2]);
    q.magic_R = super_magic(q.R);
  }
  return best_n;
}
static bool direct_ok(long long n_in, long long n_out) {
  int r[32];
  if (n_in % 2 || n_out % 2) return false;
  if (n_in / 2 > 1 && plan_radices(n_in / 2, r) == 0) return false;
  if (n_out / 2 > 1 && plan_radices(n_out / 2, r) == 0) return false;
  return true;
}

// Forward FFT of `rows` rows of length L from `in` through the two buffers; `in` may be
// bufB but not bufA.  *result = where the transform ends up (`in` itself when L == 1).
// With super-passes the last step of the Fourier resampler rides along: the last pass writes
// conj(result) * final_scale to `final_out` (*folded tells whether it did; the single-pass plans
// leave that to unpack_real_kernel).
static int fft_any(const float2* in, float2* bufA, float2* bufB, long long L, long long rows,
                   hipStream_t s, const float2** result, float2* final_out = nullptr,
                   float final_scale = 0.0f, bool* folded = nullptr) {
  if (folded) *folded = false;
  int radices[32];
  const int n = L > 1 ? plan_radices(L, radices) : 0;
  if (L > 1 && n == 0) {
    set_error("internal: length %lld is not 2-3-5-7 smooth", L);
    return MIA_ERR_UNSUPPORTED;
  }
  const float2* src = in;
  float2* dst = bufA;
  unsigned Ns = 1;
  const unsigned gy = (unsigned)(rows < 32768 ? rows : 32768);
  // Two super-passes instead of four or more single ones, for rows long enough that both
  // composite radices are large: measured on 64 x 30 s rows, 661 500 -> 240 000 points
  // 1.13 -> 0.98 ms and half the HBM traffic.  Shorter rows (any-length STFT frames, a 10 s
  // clip) and lengths that need three super-passes run faster on the single passes: few
  // super-butterflies per row leave the 16-sequence workgroups part empty, and a third trip
  // through LDS costs more than the pass it saves.
  SuperPass sp[4];
  const int n_super = (n >= 4 && L >= 200000) ? plan_super(L, sp, 2) : 0;
  if (n_super > 0) {
    const bool fold = final_out != nullptr;
    for (int i = 0; i < n_super; ++i) {
      const bool last_final = fold && i == n_super - 1;
      auto kern = stockham_super_kernel;
      const size_t lds = sizeof(float2) * (size_t)sp[i].R * (kSuperW + 1);
      if (lds > 64 * 1024) {
        MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      }
      if (last_final) dst = final_out;
      const unsigned D = (unsigned)(L / sp[i].R);
      // keep the grid around the size of the device when there are many rows: a workgroup
      // builds its twiddle table once and walks rows
      const unsigned gx = (D + kSuperW - 1) / kSuperW;
      unsigned gys = gy;
      while (gys > 1 && (unsigned long long)gx * gys > kSuperGrid) gys = (gys + 1) / 2;
      const unsigned per_xcd = (gx + 7) / 8;
      hipLaunchKernelGGL(kern, dim3(8 * per_xcd * gys), dim3(kSuperThreads), lds, s, src, dst,
                         (unsigned)L, Ns, rows, sp[i], gx, per_xcd, gys, last_final ? final_scale : 0.0f);
      MIA_CHECK_HIP(hipGetLastError());
      Ns *= (unsigned)sp[i].R;
      src = dst;
      dst = dst == bufA ? bufB : bufA;
    }
    *result = src;
    if (folded) *folded = fold;
    return MIA_OK;
  }
  for (int i = 0; i < n; ++i) {
    const int R = radices[i];
    const dim3 grid((unsigned)((L / R + 255) / 256), gy);
#define MIA_PASS(RR) \
  case RR: hipLaunchKernelGGL((stockham_any_kernel<RR>), grid, dim3(256), 0, s, src, dst, (unsigned)L, Ns, rows); break;
    switch (R) {
      MIA_PASS(36) MIA_PASS(35) MIA_PASS(32) MIA_PASS(30) MIA_PASS(28) MIA_PASS(27) MIA_PASS(25)
      MIA_PASS(24) MIA_PASS(21) MIA_PASS(20) MIA_PASS(18)
      MIA_PASS(16) MIA_PASS(15) MIA_PASS(14) MIA_PASS(12) MIA_PASS(10) MIA_PASS(9) MIA_PASS(8)
      MIA_PASS(7) MIA_PASS(6) MIA_PASS(5) MIA_PASS(4) MIA_PASS(3) MIA_PASS(2)
      default: break;
    }
#undef MIA_PASS
    MIA_CHECK_HIP(hipGetLastError());
    Ns *= (unsigned)R;
    src = dst;
    dst = dst == bufA ? bufB : bufA;
  }
  *result = src;
  return MIA_OK;
}

static int ceil_log2(long long v) {
  int p = 0;
  while ((1LL << p) < v) ++p;
  return p;
}
static int conv_log2(long long N) {  // log2 of the convolution length for a length-N transform
  const int p = ceil_log2(2 * N - 1);
  return p < 1 ? 1 : p;
}

static inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

// Forward FFT of `rows` rows of length 1 << log2L held in `a`; `b` is scratch of the same
// size.  Returns (through *result) which of the two buffers holds the transform.
static int fft_pow2(float2* a, float2* b, int log2L, long long rows, hipStream_t s,
                    float2** result) {
  if ((rows << log2L) / 2 >= (1LL << 31) * 256) {
    set_error("resample: %lld rows of 2^%d points do not fit one launch", rows, log2L);
    return MIA_ERR_UNSUPPORTED;
  }
  int left = log2L, Ns = 1;
  float2 *src = a, *dst = b;
  while (left > 0) {
    int lg = left >= 4 ? 4 : left;
    if (left == 5) lg = 3;  // 8 * 4 rather than 16 * 2
    const long long threads = rows << (log2L - lg);
    const unsigned grid = blocks_for(threads);
    switch (lg) {
      case 4: hipLaunchKernelGGL((stockham_pass_kernel<16, 4>), dim3(grid), dim3(256), 0, s, src, dst, log2L, rows, Ns); break;
      case 3: hipLaunchKernelGGL((stockham_pass_kernel<8, 3>), dim3(grid), dim3(256), 0, s, src, dst, log2L, rows, Ns); break;
      case 2: hipLaunchKernelGGL((stockham_pass_kernel<4, 2>), dim3(grid), dim3(256), 0, s, src, dst, log2L, rows, Ns); break;
      default: hipLaunchKernelGGL((stockham_pass_kernel<2, 1>), dim3(grid), dim3(256), 0, s, src, dst, log2L, rows, Ns); break;
    }
    MIA_CHECK_HIP(hipGetLastError());
    Ns <<= lg;
    left -= lg;
    float2* t = src;
    src = dst;
    dst = t;
  }
  *result = src;
  return MIA_OK;
}

struct RfLayout {
  long long N, M;
  int p1, p2;
  long long L1, L2, Lmax;
  size_t off_cN, off_cM, off_B1, off_B2, table_elems;
};
static RfLayout rf_layout(long long n_in, long long n_out) {
  RfLayout l;
  l.N = n_in;
  l.M = n_out;
  l.p1 = conv_log2(n_in);
  l.p2 = conv_log2(n_out);
  l.L1 = 1LL << l.p1;
  l.L2 = 1LL << l.p2;
  l.Lmax = l.L1 > l.L2 ? l.L1 : l.L2;
  l.off_cN = 0;
  l.off_cM = (size_t)n_in;
  l.off_B1 = l.off_cM + (size_t)n_out;
  l.off_B2 = l.off_B1 + (size_t)l.L1;
  l.table_elems = l.off_B2 + (size_t)l.L2;
  return l;
}


// ---------------------------------------------------------------------------
// Any-length frames: the plan of an n_fft that has no compiled LDS kernel.  Frames are
// materialised as complex rows in a caller workspace and transformed by the whole-row engine
// above - mixed-radix passes when n_fft is 2-3-5-7-smooth, else a chirp-z convolution over a
// power-of-two length - in chunks of rows that bound the workspace.  A fallback that accepts
// what the reference accepts (any n_fft, odd ones included); the compiled plans are the fast
// path.
// ---------------------------------------------------------------------------
constexpr size_t kGenericChunkBytes = 256u << 20;  // per buffer

static bool smooth2357(long long n) {
  int r[32];
  return n == 1 || plan_radices(n, r) != 0;
}
long long generic_row_len(int n_fft) {  // complex elements per row in the workspace buffers
  return smooth2357(n_fft) ? (long long)n_fft : (1LL << conv_log2(n_fft));
}
long long generic_chunk_rows(int n_fft, long long rows) {
  const long long per = (long long)(kGenericChunkBytes / (sizeof(float2) * (size_t)generic_row_len(n_fft)));
  const long long c = per < 1 ? 1 : per;
  return rows < c ? (rows < 1 ? 1 : rows) : c;
}
size_t generic_fft_bytes(int n_fft, long long rows) {  // the two row buffers
  return 2 * sizeof(float2) * (size_t)generic_chunk_rows(n_fft, rows) * (size_t)generic_row_len(n_fft);
}

// Chirp and filter spectrum of the chirp-z path (nothing to build for smooth lengths).
int generic_build_tables(int n_fft, float2** d_chirp, float2** d_bspec) {
  *d_chirp = nullptr;
  *d_bspec = nullptr;
  if (smooth2357(n_fft)) return MIA_OK;
  const int p = conv_log2(n_fft);
  const long long L = 1LL << p;
  float2 *w0 = nullptr, *w1 = nullptr;
  MIA_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(d_chirp), sizeof(float2) * (size_t)n_fft));
  MIA_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(d_bspec), sizeof(float2) * (size_t)L));
  MIA_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&w0), sizeof(float2) * (size_t)L));
  MIA_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&w1), sizeof(float2) * (size_t)L));
  hipLaunchKernelGGL(chirp_kernel, dim3(blocks_for(n_fft)), dim3(256), 0, 0, (long long)n_fft, *d_chirp);
  hipLaunchKernelGGL(chirp_filter_kernel, dim3(blocks_for(L)), dim3(256), 0, 0, *d_chirp,
                     (long long)n_fft, L, w0);
  float2* res = nullptr;
  int rc = fft_pow2(w0, w1, p, 1, 0, &res);
  if (rc == MIA_OK) {
    rc = hipMemcpy(*d_bspec, res, sizeof(float2) * (size_t)L, hipMemcpyDeviceToDevice) == hipSuccess
             ? MIA_OK : MIA_ERR_HIP;
  }
  (void)hipDeviceSynchronize();
  (void)hipFree(w0);
  (void)hipFree(w1);
  return rc;
}

// a[row, m] = window[m] * y[clip, frame * hop - pad + m] (times conj(chirp[m]) on the chirp-z
// path) for m < N, zero up to L; rows are frames row0 .. row0 + nrows - 1 of the (B, F) grid.
__global__ __launch_bounds__(256) void generic_frames_kernel(
    const float* __restrict__ y, long long T, long long F, int N, int hop, int pad, int pad_mode,
    const float* __restrict__ win, const float2* __restrict__ chirp, long long L, long long row0,
    long long nrows, float2* __restrict__ a) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nrows * L) return;
  const long long row = g / L, m = g - row * L;
  float2 v = {0.0f, 0.0f};
  if (m < N) {
    const long long fr = row0 + row, b = fr / F, f = fr - b * F;
    const float xv = fetch_sample(y + b * T, T, f * hop - pad + m, pad_mode) * win[m];
    if (chirp != nullptr) {
      const float2 c = chirp[m];
      v = {xv * c.x, -xv * c.y};
    } else {
      v = {xv, 0.0f};
    }
  }
  a[g] = v;
}

// Bin k of row `row` from the transformed buffer: r[k] itself on the smooth path,
// conj(chirp[k] * r[k]) / L on the chirp-z path.
__device__ __forceinline__ float2 generic_bin(const float2* __restrict__ r, const float2* __restrict__ chirp,
                                              long long L, float inv_L, long long row, long long k) {
  const float2 v = r[row * L + k];
  if (chirp == nullptr) return v;
  const float2 q = cmulf(chirp[k], v);
  return {q.x * inv_L, -q.y * inv_L};
}

__global__ __launch_bounds__(256) void generic_store_bins_kernel(
    const float2* __restrict__ r, const float2* __restrict__ chirp, long long L, float inv_L, int bins,
    long long row0, long long nrows, float scale, float2* __restrict__ out) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nrows * bins) return;
  const long long row = g / bins, k = g - row * bins;
  const float2 X = generic_bin(r, chirp, L, inv_L, row, k);
  out[(row0 + row) * bins + k] = {X.x * scale, X.y * scale};
}

// Mel projection + epilogue of the rows of one chunk: one thread per (filter, frame), frames
// fastest (coalesced stores); the per-clip extremes go straight to the keys.
__global__ __launch_bounds__(256) void generic_mel_kernel(
    const float2* __restrict__ r, const float2* __restrict__ chirp, long long L, float inv_L,
    long long F, long long row0, long long nrows, FwdParams p) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nrows * p.n_mels) return;
  const int m = (int)(g / nrows);
  const long long row = g - (long long)m * nrows;
  const long long fr = row0 + row, b = fr / F, f = fr - b * F;
  const int lo = p.row_lo[m], len = p.row_len[m];
  const float* __restrict__ w = p.wts + p.row_off[m];
  float acc = 0.0f;
  for (int j = 0; j < len; ++j) {
    const float2 X = generic_bin(r, chirp, L, inv_L, row, lo + j);
    float pw = X.x * X.x + X.y * X.y;
    if (p.power_mode == 1) pw = sqrtf(pw);
    else if (p.power_mode == 2) pw = powf(sqrtf(pw), p.power);
    acc = fmaf(w[j], pw, acc);
  }
  float outv = acc;
  if (p.epi == MIA_EPI_WHISPER) {
    const float yv = acc <= 1e-10f ? -10.0f : __log10f(acc);
    outv = (yv + 4.0f) * 0.25f;
    if (p.clip_max_key != nullptr) {
      atomicMax(p.clip_max_key + b, float_to_key_nan(yv));
      atomicMin(p.clip_min_key + b, float_to_key(outv));
    }
  } else if (p.epi == MIA_EPI_DB) {
    outv = acc <= p.amin ? p.db_floor : p.coef * __log10f(acc * p.inv_ref);
    if (p.clip_max_key != nullptr) {
      atomicMax(p.clip_max_key + b, float_to_key_nan(outv));
      atomicMin(p.clip_min_key + b, float_to_key(outv));
    }
  }
  p.mout[(b * p.n_mels + m) * F + f] = outv;
}

// Forward transform of the frames of chunk [row0, row0 + nrows); *res = the buffer holding it.
static int generic_forward_chunk(const mia_plan* plan, const FwdParams& p, long long row0, long long nrows,
                                 float2* bufA, float2* bufB, hipStream_t s, const float2** res) {
  const int N = plan->n_fft;
  const bool cz = plan->d_chirp != nullptr;
  const long long L = generic_row_len(N);
  hipLaunchKernelGGL(generic_frames_kernel, dim3(blocks_for(nrows * L)), dim3(256), 0, s, p.y, p.T, p.F,
                     N, p.hop, p.pad, p.pad_mode, p.win, cz ? plan->d_chirp : nullptr, L, row0, nrows,
                     bufA);
  MIA_CHECK_HIP(hipGetLastError());
  if (!cz) {
    if (N == 1) {
      *res = bufA;
      return MIA_OK;
    }
    return fft_any(bufA, bufB, bufA, L, nrows, s, res);  // in = its second buffer: allowed
  }
  const int lg = conv_log2(N);
  float2* r0 = nullptr;
  int rc = fft_pow2(bufA, bufB, lg, nrows, s, &r0);
  if (rc != MIA_OK) return rc;
  float2* other = r0 == bufA ? bufB : bufA;
  hipLaunchKernelGGL(spectrum_product_kernel, dim3(blocks_for(nrows << lg)), dim3(256), 0, s, r0,
                     plan->d_bspec, lg, nrows);
  MIA_CHECK_HIP(hipGetLastError());
  float2* r1 = nullptr;
  rc = fft_pow2(r0, other, lg, nrows, s, &r1);
  if (rc != MIA_OK) return rc;
  *res = r1;
  return MIA_OK;
}

int launch_forward_generic(const mia_plan* plan, const FwdParams& p, int mode, void* workspace,
                           size_t workspace_bytes, hipStream_t s) {
  const long long rows = p.B * p.F;
  if (rows <= 0) return MIA_OK;
  const int N = plan->n_fft, bins = N / 2 + 1;
  MIA_REQUIRE(workspace != nullptr && workspace_bytes >= generic_fft_bytes(N, rows),
              "workspace too small for n_fft=%d: %zu bytes, need %zu", N, workspace_bytes,
              generic_fft_bytes(N, rows));
  const long long L = generic_row_len(N), chunk = generic_chunk_rows(N, rows);
  float2* bufA = reinterpret_cast<float2*>(workspace);
  float2* bufB = bufA + chunk * L;
  const bool cz = plan->d_chirp != nullptr;
  const float inv_L = 1.0f / (float)L;
  for (long long row0 = 0; row0 < rows; row0 += chunk) {
    const long long n = rows - row0 < chunk ? rows - row0 : chunk;
    const float2* res = nullptr;
    const int rc = generic_forward_chunk(plan, p, row0, n, bufA, bufB, s, &res);
    if (rc != MIA_OK) return rc;
    if (mode == 0) {
      hipLaunchKernelGGL(generic_store_bins_kernel, dim3(blocks_for(n * bins)), dim3(256), 0, s, res,
                         cz ? plan->d_chirp : nullptr, L, inv_L, bins, row0, n, p.scale, p.cout);
    } else {
      hipLaunchKernelGGL(generic_mel_kernel, dim3(blocks_for(n * p.n_mels)), dim3(256), 0, s, res,
                         cz ? plan->d_chirp : nullptr, L, inv_L, p.F, row0, n, p);
    }
    MIA_CHECK_HIP(hipGetLastError());
  }
  return MIA_OK;
}

// Inverse: conj of the Hermitian extension of the bins of frames [row0, row0 + nrows)
// (imaginary parts of DC and of an even length's Nyquist bin dropped, as a complex-to-real
// transform ignores them), chirp-premultiplied on the chirp-z path, zero up to L.
__global__ __launch_bounds__(256) void generic_spectrum_rows_kernel(
    const float2* __restrict__ spec, long long sb, long long sk, long long sf, long long F, int N,
    const float2* __restrict__ chirp, long long L, long long row0, long long nrows,
    float2* __restrict__ a) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nrows * L) return;
  const long long row = g / L, k = g - row * L;
  float2 v = {0.0f, 0.0f};
  if (k < N) {
    const long long fr = row0 + row, b = fr / F, f = fr - b * F;
    const bool neg = 2 * k > N;
    const long long kk = neg ? N - k : k;
    float2 X = spec[b * sb + kk * sk + f * sf];
    if (kk == 0 || 2 * kk == N) X.y = 0.0f;
    // full spectrum Z[k] = neg ? conj(X) : X; the transform input is conj(Z)
    const float2 Zc = {X.x, neg ? X.y : -X.y};
    if (chirp != nullptr) {
      const float2 c = chirp[k];
      v = cmulf(Zc, float2{c.x, -c.y});
    } else {
      v = Zc;
    }
  }
  a[g] = v;
}

__global__ __launch_bounds__(256) void generic_store_frames_kernel(
    const float2* __restrict__ r, const float2* __restrict__ chirp, long long L, float inv_L, int N,
    long long row0, long long nrows, float* __restrict__ frames) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nrows * N) return;
  const long long row = g / N, n = g - row * N;
  // x[n] = Re(conj(DFT(conj Z))[n]) / N = Re(DFT(conj Z)[n]) / N
  const float2 X = generic_bin(r, chirp, L, inv_L, row, n);
  frames[(row0 + row) * N + n] = X.x / (float)N;
}

int launch_inverse_generic(const mia_plan* plan, const InvParams& p, void* workspace,
                           size_t workspace_bytes, hipStream_t s) {
  const long long rows = p.B * p.F;
  if (rows <= 0) return MIA_OK;
  const int N = plan->n_fft;
  MIA_REQUIRE(workspace != nullptr && workspace_bytes >= generic_fft_bytes(N, rows),
              "workspace too small for n_fft=%d: %zu bytes, need %zu", N, workspace_bytes,
              generic_fft_bytes(N, rows));
  const long long L = generic_row_len(N), chunk = generic_chunk_rows(N, rows);
  float2* bufA = reinterpret_cast<float2*>(workspace);
  float2* bufB = bufA + chunk * L;
  const bool cz = plan->d_chirp != nullptr;
  const float inv_L = 1.0f / (float)L;
  for (long long row0 = 0; row0 < rows; row0 += chunk) {
    const long long n = rows - row0 < chunk ? rows - row0 : chunk;
    hipLaunchKernelGGL(generic_spectrum_rows_kernel, dim3(blocks_for(n * L)), dim3(256), 0, s, p.spec,
                       p.sb, p.sk, p.sf, p.F, N, cz ? plan->d_chirp : nullptr, L, row0, n, bufA);
    MIA_CHECK_HIP(hipGetLastError());
    const float2* res = bufA;
    if (!cz) {
      if (N > 1) {
        const int rc = fft_any(bufA, bufB, bufA, L, n, s, &res);
        if (rc != MIA_OK) return rc;
      }
    } else {
      const int lg = conv_log2(N);
      float2* r0 = nullptr;
      int rc = fft_pow2(bufA, bufB, lg, n, s, &r0);
      if (rc != MIA_OK) return rc;
      float2* other = r0 == bufA ? bufB : bufA;
      hipLaunchKernelGGL(spectrum_product_kernel, dim3(blocks_for(n << lg)), dim3(256), 0, s, r0,
                         plan->d_bspec, lg, n);
      MIA_CHECK_HIP(hipGetLastError());
      float2* r1 = nullptr;
      rc = fft_pow2(r0, other, lg, n, s, &r1);
      if (rc != MIA_OK) return rc;
      res = r1;
    }
    hipLaunchKernelGGL(generic_store_frames_kernel, dim3(blocks_for(n * N)), dim3(256), 0, s, res,
                       cz ? plan->d_chirp : nullptr, L, inv_L, N, row0, n, p.frames);
    MIA_CHECK_HIP(hipGetLastError());
  }
  return MIA_OK;
}

}  // namespace mia

extern "C" {

size_t mia_resample_fft_table_bytes(int64_t n_in, int64_t n_out) {
  if (n_in < 1 || n_out < 1 || n_in > (1LL << 27) || n_out > (1LL << 27)) return 0;
  if (mia::direct_ok(n_in, n_out)) return sizeof(float2);  // no tables on the direct path
  return mia::rf_layout(n_in, n_out).table_elems * sizeof(float2);
}

size_t mia_resample_fft_workspace_bytes(int64_t rows, int64_t n_in, int64_t n_out) {
  if (rows < 1 || n_in < 1 || n_out < 1 || n_in > (1LL << 27) || n_out > (1LL << 27)) return 0;
  if (mia::direct_ok(n_in, n_out)) {
    const int64_t h = (n_in > n_out ? n_in : n_out) / 2;
    return 2 * (size_t)rows * (size_t)h * sizeof(float2);
  }
  return 2 * (size_t)rows * (size_t)mia::rf_layout(n_in, n_out).Lmax * sizeof(float2);
}

int mia_resample_fft_tables(int64_t n_in, int64_t n_out, void* tables, void* workspace,
                            size_t workspace_bytes, void* stream) {
  using namespace mia;
  MIA_REQUIRE(n_in >= 1 && n_out >= 1, "n_in and n_out must be >= 1");
  if (n_in > (1LL << 27) || n_out > (1LL << 27)) {
    set_error("resample: rows longer than 2^27 samples are not supported");
    return MIA_ERR_UNSUPPORTED;
  }
  MIA_REQUIRE(tables && workspace, "null pointer");
  if (direct_ok(n_in, n_out)) return MIA_OK;
  const RfLayout l = rf_layout(n_in, n_out);
  MIA_REQUIRE(workspace_bytes >= 2 * (size_t)l.Lmax * sizeof(float2),
              "workspace too small: %zu bytes, need %zu", workspace_bytes,
              2 * (size_t)l.Lmax * sizeof(float2));
  hipStream_t s = (hipStream_t)stream;
  float2* T = (float2*)tables;
  float2* w0 = (float2*)workspace;
  float2* w1 = w0 + l.Lmax;
  hipLaunchKernelGGL(chirp_kernel, dim3(blocks_for(l.N)), dim3(256), 0, s, l.N, T + l.off_cN);
  hipLaunchKernelGGL(chirp_kernel, dim3(blocks_for(l.M)), dim3(256), 0, s, l.M, T + l.off_cM);
  MIA_CHECK_HIP(hipGetLastError());
  const long long NN[2] = {l.N, l.M}, LL[2] = {l.L1, l.L2};
  const int PP[2] = {l.p1, l.p2};
  const size_t offc[2] = {l.off_cN, l.off_cM}, offB[2] = {l.off_B1, l.off_B2};
  for (int i = 0; i < 2; ++i) {
    hipLaunchKernelGGL(chirp_filter_kernel, dim3(blocks_for(LL[i])), dim3(256), 0, s, T + offc[i],
                       NN[i], LL[i], w0);
    MIA_CHECK_HIP(hipGetLastError());
    float2* res = nullptr;
    const int rc = fft_pow2(w0, w1, PP[i], 1, s, &res);
    if (rc != MIA_OK) return rc;
    MIA_CHECK_HIP(hipMemcpyAsync(T + offB[i], res, (size_t)LL[i] * sizeof(float2),
                                 hipMemcpyDeviceToDevice, s));
  }
  return MIA_OK;
}

int mia_resample_fft(const float* x, int64_t rows, int64_t n_in, int64_t n_out, float gain,
                     const void* tables, float* out, void* workspace, size_t workspace_bytes,
                     void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && n_in >= 1 && n_out >= 1, "rows, n_in, n_out must be >= 0, 1, 1");
  if (n_in > (1LL << 27) || n_out > (1LL << 27)) {
    set_error("resample: rows longer than 2^27 samples are not supported");
    return MIA_ERR_UNSUPPORTED;
  }
  if (rows == 0) return MIA_OK;
  MIA_REQUIRE(x && tables && out && workspace, "null pointer");
  if (direct_ok(n_in, n_out)) {
    // the rows are read and written as pairs of samples (both lengths are even)
    MIA_REQUIRE((((uintptr_t)x | (uintptr_t)out) & 7) == 0, "x and out must be 8-byte aligned");
    const long long Nh = n_in / 2, Mh = n_out / 2, H = Nh > Mh ? Nh : Mh;
    const size_t need_d = 2 * (size_t)rows * (size_t)H * sizeof(float2);
    MIA_REQUIRE(workspace_bytes >= need_d, "workspace too small: %zu bytes, need %zu",
                workspace_bytes, need_d);
    hipStream_t sd = (hipStream_t)stream;
    float2* w0 = (float2*)workspace;
    float2* w1 = w0 + (size_t)rows * H;
    const float2* Z = nullptr;
    int rcd = fft_any(reinterpret_cast<const float2*>(x), w0, w1, Nh, rows, sd, &Z);
    if (rcd != MIA_OK) return rcd;
    float2* Zp = Z == w0 ? w1 : w0;
    const long long Nmin = n_in < n_out ? n_in : n_out;
    long long shared_bin = -1;
    float shared_scale = 1.0f;
    if (n_in != n_out) {  // both even here, so the shorter length always has a Nyquist bin
      shared_bin = Nmin / 2;
      shared_scale = n_out < n_in ? 2.0f : 0.5f;
    }
    const ResizeIn rz{Z, Nh, Mh, Nmin / 2 + 1, shared_bin, shared_scale};
    const float out_scale = (float)((double)gain / (double)n_in);
    hipLaunchKernelGGL(resize_packed_kernel, dim3(blocks_for(rows * Mh)), dim3(256), 0, sd, rz,
                       (long long)rows, Zp);
    MIA_CHECK_HIP(hipGetLastError());
    // Long rows: the last inverse super-pass writes the samples itself (no unpacking pass through
    // memory).  Building the first pass's input from Z inside the pass as well was slower than
    // this kernel plus a plain pass: 1.01 against 0.94 ms for 64 x 30 s 44.1 -> 16 kHz - four
    // scattered loads and two sincos per element in a one-workgroup-per-CU kernel.
    const float2* F = nullptr;
    bool folded = false;
    rcd = fft_any(Zp, Zp == w0 ? w1 : w0, Zp, Mh, rows, sd, &F, reinterpret_cast<float2*>(out), out_scale,
                  &folded);
    if (rcd != MIA_OK) return rcd;
    if (folded) return MIA_OK;
    hipLaunchKernelGGL(unpack_real_kernel, dim3(blocks_for(rows * Mh)), dim3(256), 0, sd, F,
                       (long long)rows * Mh, out_scale, reinterpret_cast<float2*>(out));
    MIA_CHECK_HIP(hipGetLastError());
    return MIA_OK;
  }
  const RfLayout l = rf_layout(n_in, n_out);
  const size_t need = 2 * (size_t)rows * (size_t)l.Lmax * sizeof(float2);
  MIA_REQUIRE(workspace_bytes >= need, "workspace too small: %zu bytes, need %zu", workspace_bytes,
              need);
  if ((rows << (l.p1 > l.p2 ? l.p1 : l.p2)) >= (1LL << 31) * 256) {
    set_error("resample: %lld rows of %lld points do not fit one launch (split the batch)",
              (long long)rows, (long long)l.Lmax);
    return MIA_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  const float2* T = (const float2*)tables;
  float2* w0 = (float2*)workspace;
  float2* w1 = w0 + (size_t)rows * l.Lmax;

  // length-N transform of every row
  hipLaunchKernelGGL(chirp_in_real_kernel, dim3(blocks_for(rows << l.p1)), dim3(256), 0, s, x,
                     T + l.off_cN, l.N, l.p1, (long long)rows, w0);
  MIA_CHECK_HIP(hipGetLastError());
  float2 *A = nullptr, *R1 = nullptr, *A2 = nullptr, *R2 = nullptr;
  int rc = fft_pow2(w0, w1, l.p1, rows, s, &A);
  if (rc != MIA_OK) return rc;
  hipLaunchKernelGGL(spectrum_product_kernel, dim3(blocks_for(rows << l.p1)), dim3(256), 0, s, A,
                     T + l.off_B1, l.p1, (long long)rows);
  MIA_CHECK_HIP(hipGetLastError());
  rc = fft_pow2(A, A == w0 ? w1 : w0, l.p1, rows, s, &R1);
  if (rc != MIA_OK) return rc;

  // spectrum of the output, then the length-M inverse transform
  const long long Nmin = l.N < l.M ? l.N : l.M;
  const long long nyq = Nmin / 2 + 1;
  long long shared_bin = -1;
  float shared_scale = 1.0f;
  if (Nmin % 2 == 0 && l.N != l.M) {
    shared_bin = Nmin / 2;
    shared_scale = l.M < l.N ? 2.0f : 0.5f;
  }
  A2 = R1 == w0 ? w1 : w0;
  hipLaunchKernelGGL(resize_spectrum_kernel, dim3(blocks_for(rows << l.p2)), dim3(256), 0, s, R1,
                     T + l.off_cN, T + l.off_cM, l.N, l.M, l.p1, l.p2, (long long)rows, nyq,
                     shared_bin, shared_scale, (float)(1.0 / (double)l.L1), A2);
  MIA_CHECK_HIP(hipGetLastError());
  float2* other = A2 == w0 ? w1 : w0;
  float2* A2f = nullptr;
  rc = fft_pow2(A2, other, l.p2, rows, s, &A2f);
  if (rc != MIA_OK) return rc;
  hipLaunchKernelGGL(spectrum_product_kernel, dim3(blocks_for(rows << l.p2)), dim3(256), 0, s, A2f,
                     T + l.off_B2, l.p2, (long long)rows);
  MIA_CHECK_HIP(hipGetLastError());
  rc = fft_pow2(A2f, A2f == w0 ? w1 : w0, l.p2, rows, s, &R2);
  if (rc != MIA_OK) return rc;
  const float scale = (float)((double)gain / ((double)l.L2 * (double)l.N));
  hipLaunchKernelGGL(chirp_out_real_kernel, dim3(blocks_for(rows * l.M)), dim3(256), 0, s, R2,
                     T + l.off_cM, l.M, l.p2, (long long)rows, scale, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // extern "C"

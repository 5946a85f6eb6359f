// Sample-rate conversion kernels: the polyphase FIR resampler (upsample by `up`, filter,
// keep every `down`-th sample) and linear interpolation.
//
// Both reproduce the host libraries the reference delegates to bit for bit: every output
// of the polyphase filter is a float32 sum, oldest input sample first, of float32
// products rounded separately (multiply, then add: never fused), and the interpolation
// runs in float64 with the same operation order as the NumPy expression it replaces.
//
// Polyphase data flow.  Output o of a row needs the P = taps-per-phase input samples that
// end at x[((o + skip) * down) / up], weighted by row ((o + skip) * down) % up of the phase
// table.  A workgroup produces a run of consecutive outputs of one row: it stages the span
// of input samples the run touches in LDS with coalesced loads (signal extension folded
// in), keeps the phase table in LDS too when it fits, and every thread then walks its
// outputs with independent accumulators.  HBM sees each input sample about once per
// workgroup that touches it and each output once.
#include "mia_internal.h"

// Products and sums in this file are rounded one by one, as the host code they stand in
// for does it: the compiler must not contract a*b+c into a fused multiply-add here.  The
// arithmetic is written with plain operators for that reason: the __fmul_rn / __dadd_rn
// wrappers are header functions compiled with contraction allowed, and still fuse.
#pragma clang fp contract(off)

namespace mia {

constexpr int kRsOPT = 8;           // outputs per thread and pass, at most
constexpr int kRsMaxSpan = 16384;   // staged input samples per workgroup, at most (64 KiB)
constexpr int kRsMaxTable = 12288;  // phase-table entries kept in LDS, at most (48 KiB)

// Sample i of the row extended beyond its ends the way the polyphase filter's padding
// modes define it (the extension is periodic for every mode but "edge").
__device__ __forceinline__ float extended_sample(const float* __restrict__ x, long long n,
                                                 long long i, int mode) {
  if (i >= 0 && i < n) return x[i];
  switch (mode) {
    case MIA_RS_PAD_EDGE:
      return x[i < 0 ? 0 : n - 1];
    case MIA_RS_PAD_REFLECT: {  // ... x2 x1 | x0 x1 ... : the edge sample is not repeated
      if (n == 1) return x[0];
      const long long p = 2 * (n - 1);
      long long j = i % p;
      if (j < 0) j += p;
      return x[j < n ? j : p - j];
    }
    case MIA_RS_PAD_SYMMETRIC: {  // ... x1 x0 | x0 x1 ...
      const long long p = 2 * n;
      long long j = i % p;
      if (j < 0) j += p;
      return x[j < n ? j : p - 1 - j];
    }
    case MIA_RS_PAD_WRAP: {
      long long j = i % n;
      if (j < 0) j += n;
      return x[j];
    }
    default:
      return 0.0f;
  }
}

template <bool TABLE_IN_LDS>
__global__ __launch_bounds__(1024) void upfirdn_kernel(
    const float* __restrict__ x, long long n_in, const float* __restrict__ table, int up, int down,
    int P, long long skip, long long n_out, int mode, float* __restrict__ out, int S, int kcount,
    int span_cap, long long runs_per_row, long long total_runs, int runs_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float rs_smem[];
  float* xs = rs_smem;
  float* ht = rs_smem + span_cap;  // span_cap includes the slack of the 16-byte copy
  // rows of the LDS copy of the phase table have an odd pitch: lanes read different phases
  const int PL = TABLE_IN_LDS ? (P | 1) : P;
  const int run = S * kcount;
  const int dS = (S / up) * down;  // input samples between outputs S apart (used when up | S)

  const int tid = threadIdx.x, nthr = blockDim.x;
  if constexpr (TABLE_IN_LDS) {
    for (int i = tid; i < up * P; i += nthr) {
      const int t = i / P;
      ht[t * PL + (i - t * P)] = table[i];
    }
  }
  // persistent workgroup: a contiguous block of output runs
  const long long first = (long long)blockIdx.x * runs_per_wg;
  const long long last = first + runs_per_wg < total_runs ? first + runs_per_wg : total_runs;
  for (long long item = first; item < last; ++item) {
    const long long row = item / runs_per_row;
    const long long o0 = (item - row * runs_per_row) * run;
    const long long rem = n_out - o0;
    const int cnt = rem < run ? (int)rem : run;
    const float* __restrict__ xr = x + row * n_in;

    const long long q0 = (o0 + skip) * down;  // position of the run's first output on the
    const long long xq0 = q0 / up;            // upsampled grid, split into sample and phase
    const unsigned ph0 = (unsigned)(q0 - xq0 * up);
    const long long x_first = xq0 - P + 1;
    const unsigned q_last = ph0 + (unsigned)(cnt - 1) * (unsigned)down;
    const int span = (int)(q_last / (unsigned)up) + P;  // <= span_cap by the launcher's choice of run

    // Staging.  One decision per run: a span that lies inside the row is copied with
    // independent 16-byte loads (the copy starts up to 3 samples early, on a 16-byte
    // boundary, and `lead` remembers by how much); only runs that touch an end of the row
    // take the per-sample path that knows about the signal extension.
    const int lead = (int)(((uintptr_t)(xr + x_first) >> 2) & 3);
    const int quads = (span + lead + 3) >> 2;
    const bool inside = x_first - lead >= 0 && x_first - lead + 4LL * quads <= n_in &&
                        (((uintptr_t)x) & 3) == 0;
    int shift = 0;
    if (inside) {
      shift = lead;
      const float4* __restrict__ src = reinterpret_cast<const float4*>(xr + x_first - lead);
      float4* dst = reinterpret_cast<float4*>(xs);
      int i = tid;
      for (; i + 3 * nthr < quads; i += 4 * nthr) {
        const float4 v0 = src[i], v1 = src[i + nthr], v2 = src[i + 2 * nthr], v3 = src[i + 3 * nthr];
        dst[i] = v0;
        dst[i + nthr] = v1;
        dst[i + 2 * nthr] = v2;
        dst[i + 3 * nthr] = v3;
      }
      for (; i < quads; i += nthr) dst[i] = src[i];
    } else {
      for (int i = tid; i < span; i += nthr) xs[i] = extended_sample(xr, n_in, x_first + i, mode);
    }
    __syncthreads();

    // Thread w produces outputs w, w + S, w + 2S, ...: S is a multiple of `up`, so they
    // share one phase and every tap is read once for all of them.
    float* __restrict__ orow = out + row * n_out + o0;
    for (int w = tid; w < S && w < cnt; w += nthr) {
      const unsigned q = ph0 + (unsigned)w * (unsigned)down;
      const unsigned xi = q / (unsigned)up;
      const unsigned t = q - xi * (unsigned)up;
      const float* __restrict__ hh = (TABLE_IN_LDS ? ht : table) + (size_t)t * PL;
      const float* xk[kRsOPT];
      float acc[kRsOPT];
#pragma unroll
      for (int k = 0; k < kRsOPT; ++k) {
        const bool live = k < kcount && w + k * S < cnt;
        xk[k] = xs + shift + xi + (live ? k * dS : 0);  // a dead slot re-reads slot 0 and is not stored
        acc[k] = 0.0f;
      }
      // taps in blocks of 8: all of a block's LDS reads are issued (at constant offsets
      // from one address per stream) before its 64 multiply-adds, oldest sample first
      int j = 0;
      for (; j + 8 <= P; j += 8) {
        float hv[8], xv[kRsOPT][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) hv[u] = hh[j + u];
#pragma unroll
        for (int k = 0; k < kRsOPT; ++k) {
#pragma unroll
          for (int u = 0; u < 8; ++u) xv[k][u] = xk[k][j + u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
          for (int k = 0; k < kRsOPT; ++k) acc[k] = acc[k] + xv[k][u] * hv[u];
        }
      }
      for (; j < P; ++j) {
        const float hv = hh[j];
#pragma unroll
        for (int k = 0; k < kRsOPT; ++k) acc[k] = acc[k] + xk[k][j] * hv;
      }
#pragma unroll
      for (int k = 0; k < kRsOPT; ++k) {
        if (k < kcount && w + k * S < cnt) orow[w + k * S] = acc[k];
      }
    }
    __syncthreads();  // the next run overwrites the staged samples
  }
}

__global__ __launch_bounds__(256) void lerp_kernel(const float* __restrict__ x, long long rows,
                                                   long long n_in, long long n_out, double step,
                                                   double gain, float* __restrict__ out) {
  const long long total = rows * n_out;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long r = idx / n_out;
    const long long i = idx - r * n_out;
    // the evenly spaced positions 0 .. n_in-1: i*step, the last one exactly n_in-1
    const double pos = (i == n_out - 1 && n_out > 1) ? (double)(n_in - 1) : (double)i * step;
    const long long lo = (long long)floor(pos);
    const long long hi = lo + 1 < n_in ? lo + 1 : n_in - 1;
    const double frac = pos - (double)lo;
    const float* __restrict__ xr = x + r * n_in;
    const double a = (double)xr[lo], b = (double)xr[hi];
    double v = (1.0 - frac) * a + frac * b;
    v = v * gain;
    out[idx] = (float)v;
  }
}

}  // namespace mia

extern "C" {

int mia_upfirdn(const float* x, int64_t rows, int64_t n_in, const float* phase_table, int up,
                int down, int taps_per_phase, int64_t skip, int64_t n_out, int pad_mode, float* out,
                void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && n_in >= 1 && n_out >= 0, "rows, n_in, n_out must be >= 0, 1, 0");
  MIA_REQUIRE(up >= 1 && down >= 1 && taps_per_phase >= 1 && skip >= 0,
              "up, down, taps_per_phase must be >= 1 and skip >= 0");
  MIA_REQUIRE(pad_mode >= MIA_RS_PAD_CONSTANT && pad_mode <= MIA_RS_PAD_WRAP,
              "unknown padding mode %d", pad_mode);
  if (rows == 0 || n_out == 0) return MIA_OK;
  MIA_REQUIRE(x && phase_table && out, "null pointer");
  const int P = taps_per_phase;
  if (P + 16 > kRsMaxSpan || (long long)up > 0x3fffffff || (long long)down > 0x3fffffff) {
    set_error("polyphase filter too long for one workgroup: %d taps per phase (limit %d)", P,
              kRsMaxSpan - 16);
    return MIA_ERR_UNSUPPORTED;
  }
  // Outputs S apart share a filter phase when `up` divides S.  S is also the number of
  // threads with work, so it is picked among the multiples of `up` near 256..1024 for the
  // fewest idle lanes; a thread then produces up to kRsOPT outputs per staged span, as many
  // as keep the span within its LDS budget and the upsampled-grid positions within 32 bits.
  const long long by_span = ((long long)(kRsMaxSpan - P - 14) * up) / down;
  const long long by_bits = (0x7fffffffLL - up) / down;
  const long long max_run = by_span < by_bits ? by_span : by_bits;
  long long S = 0;
  int kcount = 1, threads = 64;
  {
    const int target = 320;  // measured best among 320..960 for 160/441 (more workgroups per CU)
    double best = 2.0;
    for (long long m = 1; m * up <= 1024 || m == 1; ++m) {
      const long long cand = m * up;
      if (cand > max_run && S != 0) break;
      const long long bd = cand >= 1024 ? 1024 : (cand + 63) / 64 * 64;
      double idle = cand >= 1024 ? 0.0 : double(bd - cand) / double(bd);
      idle += 0.002 * double(cand > target ? cand - target : target - cand) / 64.0;  // prefer ~target
      if (idle < best) {
        best = idle;
        S = cand;
        threads = (int)bd;
      }
    }
    if (S > max_run) {  // not even one output per thread fits: a short run, phases not shared
      S = max_run < 1 ? 1 : max_run;
      threads = (int)(S >= 1024 ? 1024 : (S + 63) / 64 * 64);
      kcount = 1;
    } else {
      long long k = max_run / S;
      kcount = (int)(k > kRsOPT ? kRsOPT : k);
      if (kcount < 1) kcount = 1;
    }
  }
  const long long run = S * kcount;
  const int span_cap = ((int)(((run - 1) * down + up - 1) / up) + P + 1 + 8 + 3) & ~3;
  const bool table_lds = (long long)up * (P | 1) <= kRsMaxTable;
  const long long runs_per_row = (n_out + run - 1) / run;
  const long long total_runs = rows * runs_per_row;
  if ((long long)(skip + n_out) > (0x7fffffffffffffffLL / down)) {
    set_error("resample: row of %lld outputs is too long", (long long)n_out);
    return MIA_ERR_UNSUPPORTED;
  }
  const size_t lds = sizeof(float) * ((size_t)span_cap + (table_lds ? (size_t)up * (P | 1) : 0));
  // as many workgroups as the device holds at once (the phase table is staged once per
  // workgroup), each walking a contiguous block of runs
  int dev = 0, cus = 256;
  MIA_CHECK_HIP(hipGetDevice(&dev));
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  long long per_cu = (160 * 1024) / (long long)(lds + 256);
  if (per_cu > 2048 / threads) per_cu = 2048 / threads;
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  long long grid = (long long)cus * per_cu;
  if (grid > total_runs) grid = total_runs;
  const long long per_wg = (total_runs + grid - 1) / grid;
  if (per_wg > 0x7fffffffLL) {
    set_error("resample: %lld output runs do not fit one launch (split the batch)", total_runs);
    return MIA_ERR_UNSUPPORTED;
  }
  grid = (total_runs + per_wg - 1) / per_wg;
  const int runs_per_wg = (int)per_wg;
  hipStream_t s = (hipStream_t)stream;
  if (table_lds) {
    auto kern = upfirdn_kernel<true>;
    if (lds > 64 * 1024) {
      MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, s, x, (long long)n_in,
                       phase_table, up, down, P, (long long)skip, (long long)n_out, pad_mode, out,
                       (int)S, kcount, span_cap, runs_per_row, total_runs, runs_per_wg);
  } else {
    auto kern = upfirdn_kernel<false>;
    if (lds > 64 * 1024) {
      MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, s, x, (long long)n_in,
                       phase_table, up, down, P, (long long)skip, (long long)n_out, pad_mode, out,
                       (int)S, kcount, span_cap, runs_per_row, total_runs, runs_per_wg);
  }
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int mia_resample_linear(const float* x, int64_t rows, int64_t n_in, int64_t n_out, double step,
                        double gain, float* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && n_in >= 1 && n_out >= 0, "rows, n_in, n_out must be >= 0, 1, 0");
  if (rows == 0 || n_out == 0) return MIA_OK;
  MIA_REQUIRE(x && out, "null pointer");
  MIA_REQUIRE(step >= 0.0 && (n_out == 1 || step * (double)(n_out - 1) <= (double)(n_in - 1) * (1.0 + 1e-12)),
              "step %g walks past the end of a row of %lld samples", step, (long long)n_in);
  const long long total = (long long)rows * n_out;
  long long grid = (total + 255) / 256;
  if (grid > 65536) grid = 65536;
  hipLaunchKernelGGL(lerp_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)rows, (long long)n_in, (long long)n_out, step, gain, out);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // extern "C"

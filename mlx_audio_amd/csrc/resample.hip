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
// workgroup that touches it and each output once.  (upfirdn_kernel: any ratio, any row length.)
// A second kernel, upfirdn_shared_kernel further down, gives the same bits faster for the ratios
// and row lengths it takes: lanes a whole number of filter periods apart, taps through the scalar
// cache, the samples of adjacent outputs read from LDS once.
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
    // The four modes below extend by arithmetic on the samples; every operation is a single
    // float32 one in SciPy's order (scipy/signal/_upfirdn_apply.pyx, _extend_left / _extend_right),
    // so the results are the same bits.  A row of one sample has no slope or mirror there
    // (SciPy reads past the row): it extends by its sample here.
    case MIA_RS_PAD_SMOOTH: {  // the end segments continued as straight lines
      if (n == 1) return x[0];
      if (i < 0) return x[0] + (float)i * (x[1] - x[0]);
      return x[n - 1] + (float)(i - n + 1) * (x[n - 1] - x[n - 2]);
    }
    case MIA_RS_PAD_LINE: {  // the line through the first and the last sample
      if (n == 1) return x[0];
      const float slope = (x[n - 1] - x[0]) / (float)(n - 1);
      if (i < 0) return x[0] + (float)i * slope;
      return x[n - 1] + (float)(i - n + 1) * slope;
    }
    case MIA_RS_PAD_ANTISYMMETRIC: {  // ... -x1 -x0 | x0 x1 ... : period 2 n, odd about the ends
      const long long p = 2 * n;
      if (i < 0) {
        const long long j = (-i - 1) % p;
        return j < n ? -x[j] : x[p - 1 - j];
      }
      const long long j = i % p;
      return j < n ? x[j] : -x[p - 1 - j];
    }
    case MIA_RS_PAD_ANTIREFLECT: {  // point reflection about the end samples, repeated
      if (n == 1) return x[0];
      const long long p = 2 * (n - 1);
      if (i < 0) {
        if (-i < n) return x[0] - (x[-i] - x[0]);
        const float le = x[0] + (x[0] - x[n - 1]) * (float)((-i - 1) / (n - 1));
        const long long j = (-i - 1) % p;
        if (j < n - 1) return le - (x[j + 1] - x[0]);
        return le - (x[n - 1] - x[p - 1 - j]);
      }
      if (i < 2 * n - 1) return x[n - 1] - (x[2 * n - 2 - i] - x[n - 1]);
      const float re = x[n - 1] + (x[n - 1] - x[0]) * (float)(i / (n - 1) - 1);
      const long long j = i % p;
      if (j < n - 1) return re + (x[j] - x[0]);
      return re + (x[n - 1] - x[p - j]);
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

// ---- Polyphase filter with the sample window shared in registers ---------------------------
//
// Outputs `up` apart use the same filter phase, and their sample windows lie `down` apart.
// Here a lane takes a sub-run of U = c * up consecutive outputs and the 64 lanes of a wave take
// 64 consecutive sub-runs, so at any moment all lanes of a wave work on the same phase: the
// taps are wave-uniform and come through the scalar path, and the lanes' sample addresses are
// c * down apart (odd, or 2 mod 4: no or two-way bank conflicts).  A wave-step produces R (4 or
// 5) adjacent outputs per lane: their windows overlap almost completely, so the lane reads the
// union once (16 samples at a time) and every sample feeds all R accumulators — a quarter of
// the LDS reads of one read per tap, none of them for taps.  Output j of the R starts d_j
// samples later than output 0; its taps are addressed as h[s - d_j] in a table whose rows carry
// `margin` zeros on both sides, so positions outside an output's own window add x * 0 = +-0 to
// a sum that is never -0: the result is the same float32 sum, oldest sample first, as long as
// the samples are finite.  Runs whose staged span holds an infinity or a NaN (where x * 0
// would differ from skipping the product) take a plain per-output loop instead.
//
// While a run is computed the next run's span is already on its way into registers.

constexpr int kShChunk = 16;  // samples per chunk of a wave-step (two half-chunks of eight)
constexpr int kShPre = 8;    // 16-byte loads per thread that hold the next run's span
constexpr int kShBt = 12;    // ints per wave-step in the table of step constants

// Where run k of a row lies.  A run is 64 U outputs = a whole number of filter periods, so run k
// starts 64 c * down samples after run 0 and on the same phase: no division per run (every
// wave works this out for itself, on the scalar unit: two 64-bit divisions per run and wave
// were a quarter of all instructions the kernel issued).
struct ShGeom {
  long long xq_skip;  // sample index of output 0's newest sample: (skip * down) / up
  unsigned ph0;       // and its phase: (skip * down) % up
  int span_full;      // samples a full run stages
  int run, stride, P, over, up, down;
};

struct ShRun {
  long long row, o0, x_first;
  int k, cnt, span, lead, quads;
};

__device__ __forceinline__ ShRun sh_run(const ShGeom& g, long long row, int k, long long n_out,
                                        const float* __restrict__ x, long long n_in) {
  ShRun r;
  r.row = row;
  r.k = k;
  r.o0 = (long long)k * g.run;
  const long long rem = n_out - r.o0;
  r.cnt = rem < g.run ? (int)rem : g.run;
  r.x_first = (long long)k * g.stride + g.xq_skip - g.P + 1;
  // `over` samples behind the last window are read too (with zero taps): they must be staged,
  // so that they are finite or seen not to be
  r.span = g.span_full;
  if (r.cnt != g.run) {
    const unsigned q_last = g.ph0 + (unsigned)(r.cnt - 1) * (unsigned)g.down;
    r.span = (int)(q_last / (unsigned)g.up) + g.P + g.over;
  }
  const float* xr = x + row * n_in;
  r.lead = (int)(((uintptr_t)(xr + r.x_first) >> 2) & 3);
  r.quads = (r.span + r.lead + 3) >> 2;
  return r;
}

typedef float tap8 __attribute__((ext_vector_type(8)));
typedef float smp2 __attribute__((ext_vector_type(2)));

// Eight consecutive floats at base + byte_off, both wave-uniform, through the scalar cache.
__device__ __forceinline__ void tap_load(tap8& dst, const float* base, unsigned byte_off) {
  asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(dst) : "s"(base), "s"(byte_off));
}

// Eight consecutive floats of LDS at a per-lane byte address (4-byte aligned).
__device__ __forceinline__ void smp_load(smp2 (&d)[4], unsigned lds_byte_addr) {
  asm volatile(
      "ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %4 offset0:2 offset1:3\n\t"
      "ds_read2_b32 %2, %4 offset0:4 offset1:5\n\tds_read2_b32 %3, %4 offset0:6 offset1:7"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
      : "v"(lds_byte_addr));
}

template <int R>
__device__ __forceinline__ void tap_wait(tap8 (&t)[R], smp2 (&d)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t[0]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
#pragma unroll
  for (int j = 1; j < R; ++j) asm volatile("" : "+s"(t[j]));  // stays behind the wait
}

__device__ __forceinline__ bool not_finite4(const float4 v) {
  return !(__builtin_isfinite(v.x) && __builtin_isfinite(v.y) && __builtin_isfinite(v.z) &&
           __builtin_isfinite(v.w));
}

// ROWS: the layout for lane spacings that are a multiple of four samples (all lanes would read
// one LDS bank): every lane has a row of its own, `rp` floats apart (odd), holding the `rw`
// samples from its sub-run's first window to its last — the overlap with the next lane's row
// is stored twice.  A wave stages rows wave, wave + 16, ... with lanes along the row, dword by
// dword (rows start at any sample); needs 16 waves.
template <int R, bool ROWS>
__global__ __launch_bounds__(1024) void upfirdn_shared_kernel(
    const float* __restrict__ x, long long n_in, const float* __restrict__ table, int up, int down,
    int P, int margin, long long skip, long long n_out, int mode, float* __restrict__ out, int U,
    int cd, int span_cap, long long runs_per_row, long long total_runs, int runs_per_wg, int rp,
    int rw) {
  extern __shared__ __attribute__((aligned(16))) float rs_smem[];
  float* xs = rs_smem;
  int* bt = reinterpret_cast<int*>(rs_smem + span_cap);  // per wave-step constants, kShBt ints each
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int PP = P + 2 * margin;
  const int run = 64 * U;
  const int nblocks = (U + R - 1) / R;
  volatile int* flag = bt + nblocks * kShBt;
  if (tid == 0) *flag = -1;  // runs are numbered from 0
  // A run is a multiple of `up` outputs, so every run starts on the same filter phase, and the
  // phases and sample offsets of a wave-step's R outputs are the same in every run: worked out
  // once.  Entry jj < R: byte offset into the table of the taps that go with the step's sample
  // position 0 for output jj (row * pitch + margin - d_jj); entry 8: the step's first sample,
  // relative to the lane's; entry 9: its number of half-chunks.
  {
    const unsigned ph0 = (unsigned)((skip * down) % up);
    const unsigned ddiv = (unsigned)down / (unsigned)up, dmod = (unsigned)down % (unsigned)up;
    for (int jb = tid; jb < nblocks; jb += nthr) {
      const unsigned qj = ph0 + (unsigned)(jb * R) * (unsigned)down;
      unsigned xi = qj / (unsigned)up;
      unsigned t = qj - xi * (unsigned)up;
      const unsigned xi0 = xi;
      int dlast = 0;
      for (int jj = 0; jj < R; ++jj) {
        dlast = (int)(xi - xi0);
        bt[jb * kShBt + jj] = 4 * ((int)t * PP + margin - dlast);
        t += dmod;
        xi += ddiv;
        if (t >= (unsigned)up) {
          t -= (unsigned)up;
          ++xi;
        }
      }
      bt[jb * kShBt + 8] = (int)xi0;
      bt[jb * kShBt + 9] = (P + dlast + kShChunk / 2 - 1) / (kShChunk / 2);
    }
  }
  __syncthreads();

  const long long first = (long long)blockIdx.x * runs_per_wg;
  const long long last = first + runs_per_wg < total_runs ? first + runs_per_wg : total_runs;
  ShGeom g;
  {
    const long long qs = skip * down;
    g.xq_skip = qs / up;
    g.ph0 = (unsigned)(qs - g.xq_skip * up);
    g.run = run, g.stride = 64 * cd, g.P = P, g.over = margin, g.up = up, g.down = down;
    g.span_full = (int)((g.ph0 + (unsigned)(run - 1) * (unsigned)down) / (unsigned)up) + P + margin;
  }
  long long row = first / runs_per_row;
  int k = (int)(first - row * runs_per_row);
  ShRun r = sh_run(g, row, k, n_out, x, n_in);
  float4 pre[kShPre];
#pragma unroll
  for (int k = 0; k < kShPre; ++k) pre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  bool have_pre = false;
  for (long long item = first; item < last; ++item) {
    const int id = (int)(item - first);
    const float* __restrict__ xr = x + r.row * n_in;

    // the run's span of samples into LDS, 16 bytes at a time from a 16-byte boundary of the
    // row (`lead` samples early): from the registers filled during the previous run, else
    // fetched here; quads that leave the row are built sample by sample with the signal
    // extension; a thread that meets a non-finite sample marks the run
    bool bad = false;
    const int shift = ROWS ? 0 : r.lead;
    if constexpr (ROWS) {
      // element (a, m) of a thread: row wave + 16 a, position lane + 64 m
      float* pf = reinterpret_cast<float*>(pre);
      const bool edge = r.x_first < 0 || r.x_first + 63LL * cd + rw > n_in;  // wave-uniform
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int l = wave + 16 * a;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int j = lane + 64 * m;
          const long long sidx = r.x_first + (long long)l * cd + j;
          if (j < rw && sidx >= 0 && sidx < n_in) {
            const float v = have_pre ? pf[a * 8 + m] : xr[sidx];
            xs[l * rp + j] = v;
            bad |= !__builtin_isfinite(v);
          }
        }
      }
      if (edge) {
#pragma unroll 1
        for (int a = 0; a < 4; ++a) {
          const int l = wave + 16 * a;
#pragma unroll 1
          for (int j = lane; j < rw; j += 64) {
            const long long sidx = r.x_first + (long long)l * cd + j;
            if (sidx >= 0 && sidx < n_in) continue;
            const float v = extended_sample(xr, n_in, sidx, mode);
            xs[l * rp + j] = v;
            bad |= !__builtin_isfinite(v);
          }
        }
      }
    } else
    {
      float4* dst = reinterpret_cast<float4*>(xs);
      const long long a0 = r.x_first - r.lead;  // sample index of quad 0 (may lie before the row)
      if (have_pre) {
#pragma unroll
        for (int kk = 0; kk < kShPre; ++kk) {
          const int i = tid + kk * nthr;
          const long long a = a0 + 4LL * i;
          if (i < r.quads && a >= 0 && a + 4 <= n_in) {
            dst[i] = pre[kk];
            bad |= not_finite4(pre[kk]);
          }
        }
      } else {
        for (int i = tid; i < r.quads; i += nthr) {
          const long long a = a0 + 4LL * i;
          if (a >= 0 && a + 4 <= n_in) {
            const float4 v = *reinterpret_cast<const float4*>(xr + a);
            dst[i] = v;
            bad |= not_finite4(v);
          }
        }
      }
      if (a0 < 0 || a0 + 4LL * r.quads > n_in) {  // the run touches an end of the row (wave-uniform)
        for (int i = tid; i < r.quads; i += nthr) {
          const long long a = a0 + 4LL * i;
          if (a >= 0 && a + 4 <= n_in) continue;
#pragma unroll 1
          for (int e = 0; e < 4; ++e) {
            const float v = extended_sample(xr, n_in, a + e, mode);
            xs[4 * i + e] = v;
            bad |= !__builtin_isfinite(v);
          }
        }
      }
    }
    if (bad) *flag = id;
    __syncthreads();
    const bool plain = __builtin_amdgcn_readfirstlane(*flag) == id;

    have_pre = false;
    ShRun nx = r;
    if (item + 1 < last) {
      if (++k == (int)runs_per_row) {
        k = 0;
        ++row;
      }
      nx = sh_run(g, row, k, n_out, x, n_in);
      if constexpr (ROWS) {
        const float* __restrict__ nr = x + nx.row * n_in;
        float* pf = reinterpret_cast<float*>(pre);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const int j = lane + 64 * m;
            const long long sidx = nx.x_first + (long long)(wave + 16 * a) * cd + j;
            if (j < rw && sidx >= 0 && sidx < n_in) pf[a * 8 + m] = nr[sidx];
          }
        }
        have_pre = true;
      } else if (nx.quads <= kShPre * nthr) {
        const float* __restrict__ nr = x + nx.row * n_in;
        const long long a0 = nx.x_first - nx.lead;
#pragma unroll
        for (int kk = 0; kk < kShPre; ++kk) {
          const int i = tid + kk * nthr;
          const long long a = a0 + 4LL * i;  // quads that leave the row are filled when staged
          if (i < nx.quads && a >= 0 && a + 4 <= n_in)
            pre[kk] = *reinterpret_cast<const float4*>(nr + a);
        }
        have_pre = true;
      }
    }

    float* __restrict__ orow = out + r.row * n_out + r.o0;
    if (!plain) {
      const float* xl = ROWS ? xs + lane * rp : xs + shift + lane * cd;
      for (int jb = wave; jb < nblocks; jb += nw) {
        // the step's constants (wave-uniform: through the scalar registers)
        const int* bc = bt + jb * kShBt;
        const int xi0 = __builtin_amdgcn_readfirstlane(bc[8]);
        const int nch = __builtin_amdgcn_readfirstlane(bc[9]);
        float acc[R];
#pragma unroll
        for (int jj = 0; jj < R; ++jj) acc[jj] = 0.0f;
        // Taps (scalar loads, eight per output) and samples (LDS reads, eight per lane) come in
        // two sets that take turns, each requested one half-chunk (8 R multiply-adds) before
        // its use.  Scalar loads return out of order, so the only wait there is is "all of
        // them", placed after the arithmetic that covers it.  Loads and waits are invisible
        // to the compiler (it would wait right behind each load); a wait names the registers
        // it makes valid, so that no use can move in front of it, and every request names a
        // register of the arithmetic that has to stay behind it.
        unsigned xa = (unsigned)(unsigned long long)(xl + xi0);  // LDS byte address of the window
        const float* pa = table;
        const float* pb = table + 8;
        tap8 ta[R], tb[R];
        smp2 va[4], vb[4];
        unsigned hb[R];
#pragma unroll
        for (int jj = 0; jj < R; ++jj) hb[jj] = (unsigned)__builtin_amdgcn_readfirstlane(bc[jj]);
#pragma unroll
        for (int jj = 0; jj < R; ++jj) tap_load(ta[jj], pa, hb[jj]);
        smp_load(va, xa);
        for (int c = 0; 2 * c < nch; ++c) {  // nch counts half-chunks
          tap_wait(ta, va);
          pa += kShChunk;
#pragma unroll
          for (int jj = 0; jj < R; ++jj) tap_load(tb[jj], pb, hb[jj]);
          smp_load(vb, xa + 32u);
          pb += kShChunk;
          xa += 64u;
          asm volatile("" : "+v"(va[0]));
#pragma unroll
          for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int jj = 0; jj < R; ++jj) acc[jj] = acc[jj] + va[u >> 1][u & 1] * ta[jj][u];
          }
          tap_wait(tb, vb);
          // behind the last half-chunk these fetch margin zeros and the samples behind the window; unused
#pragma unroll
          for (int jj = 0; jj < R; ++jj) tap_load(ta[jj], pa, hb[jj]);
          smp_load(va, xa);
          asm volatile("" : "+v"(vb[0]));
          if (2 * c + 1 < nch) {  // an odd count ends behind the first half (all taps zero beyond)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
              for (int jj = 0; jj < R; ++jj) acc[jj] = acc[jj] + vb[u >> 1][u & 1] * tb[jj][u];
            }
          }
        }
        tap_wait(ta, va);  // nothing may stay in flight into registers the compiler reuses
        float* __restrict__ ol = orow + lane * U + jb * R;
        if (r.cnt == run && jb * R + R <= U) {  // wave-uniform: a whole run, a whole step
#pragma unroll
          for (int jj = 0; jj < R; ++jj) ol[jj] = acc[jj];
        } else {
#pragma unroll
          for (int jj = 0; jj < R; ++jj) {
            const int j = jb * R + jj;
            if (j < U && lane * U + j < r.cnt) ol[jj] = acc[jj];
          }
        }
      }
    } else {
      for (int w = tid; w < r.cnt; w += nthr) {
        const unsigned q = g.ph0 + (unsigned)w * (unsigned)down;
        const unsigned xi = q / (unsigned)up;
        const unsigned t = q - xi * (unsigned)up;
        const float* __restrict__ hh = table + (size_t)t * PP + margin;
        const float* xk = xs + shift + xi;
        if constexpr (ROWS) {
          const unsigned sub = (unsigned)w / (unsigned)U;  // the sub-run, i.e. the row
          xk = xs + sub * rp + (xi - sub * (unsigned)cd);
        }
        float acc = 0.0f;
        for (int j = 0; j < P; ++j) acc = acc + xk[j] * hh[j];
        orow[w] = acc;
      }
    }
    __syncthreads();  // the next run overwrites the staged samples
    r = nx;
  }
}

// Launch shape of the shared-window kernel for one (up, down, taps-per-phase), or false when
// it does not apply: lanes c * down samples apart must not meet in one LDS bank more than
// two at a time, and 64 sub-runs' samples must fit the workgroup's LDS.
struct ShPlan {
  int c, U, cd, R, span_cap, margin, threads;
  int rp, rw;  // row pitch and length of the row-per-lane layout; 0 = one flat span
};

static bool sh_plan(int up, int down, int P, ShPlan* pl) {
  if (up < 1 || down < 1 || P < 1) return false;
  const long long dmax5 = ((long long)up - 1 + 4LL * down) / up;  // for the most outputs per step
  if (dmax5 > 2048) return false;
  const long long extra5 = (long long)P + dmax5 + 40 + 4;
  // floats of LDS for the staged span AND the table of step constants (kShBt ints per wave-step,
  // ceil(c * up / r) steps with r >= 4): two workgroups per CU, else one
  const long long caps[2] = {19000, 38000};
  const long long per_c = 64LL * down + ((long long)kShBt * up + 3) / 4;
  for (int ci = 0; ci < 2; ++ci) {
    const long long cmax = (caps[ci] - extra5 - kShBt) / per_c;
    if (cmax < 1) continue;
    long long best_c = 0;
    double best = 1e9;
    for (long long c = cmax; c >= 1 && c > cmax - 32; --c) {
      const long long cd = c * down, U = c * up;
      if (cd % 4 == 0) continue;
      if ((64 * U + 8) * (long long)down + up >= 0x7fffffffLL) continue;
      const double score = (cd % 2 == 0 ? 0.5 : 0.0) + 0.001 * double(cmax - c);
      if (score < best) {
        best = score;
        best_c = c;
      }
    }
    if (best_c == 0) continue;
    const long long U = best_c * up;
    if (U < 32 && ci == 0) continue;  // few outputs per sub-run: try the larger span
    if (U < 8) return false;
    // outputs per lane and wave-step (two sets of 8 taps each have to fit the scalar registers):
    // the one with the least work on the busiest wave (steps per wave x chunks per step x
    // outputs per step, zero-tap positions included)
    int best_r = 0;
    long long best_work = 0;
    for (int r = 5; r >= 4; --r) {
      const long long nblocks = (U + r - 1) / r;
      const long long nw = nblocks < 16 ? nblocks : 16;
      const long long d = ((long long)up - 1 + (long long)(r - 1) * down) / up;
      const long long work = ((nblocks + nw - 1) / nw) * ((P + d + kShChunk - 1) / kShChunk) * r;
      if (best_r == 0 || work < best_work) {
        best_r = r;
        best_work = work;
      }
    }
    const long long dmax = ((long long)up - 1 + (long long)(best_r - 1) * down) / up;
    pl->c = (int)best_c;
    pl->U = (int)U;
    pl->cd = (int)(best_c * down);
    pl->R = best_r;
    pl->span_cap = (int)((64LL * pl->cd + P + dmax + 40 + 4 + 3) & ~3LL);
    pl->margin = (int)dmax + 24;
    const long long nblocks = (U + best_r - 1) / best_r;
    if (pl->span_cap + kShBt * nblocks > caps[ci]) continue;  // (the estimate above is exact or high)
    pl->threads = (int)(64 * (nblocks < 16 ? nblocks : 16));
    pl->rp = pl->rw = 0;
    return true;
  }
  // No flat span: every lane spacing c * down is a multiple of four samples.  Row per lane, with
  // the fewest outputs per lane that still give all 16 waves a step (16 steps of 5 outputs).
  // Worth it for long filters only (decimation by 4, 8 ...): with some twenty taps per output the
  // dword-wise staging of overlapping rows costs more than the shared window saves (48 -> 44.1
  // kHz, 23 taps: 1.13 ms against 0.77 for the per-output kernel; 64 -> 16 kHz, 85 taps: 0.97
  // against 1.37).
  if (P < 48) return false;
  {
    const int r = 5;
    const long long c = (16 * r + up - 1) / up;
    const long long U = c * up, cd = c * down;
    const long long dmax = ((long long)up - 1 + (long long)(r - 1) * down) / up;
    const long long rw = cd + P + dmax + 48;
    const long long rp = rw | 1;
    // 32 samples per thread; the CU's LDS (span and the step constants of 16 steps)
    if (rw > 512 || 64 * rp + 3 + kShBt * ((U + r - 1) / r) > 38000) return false;
    if ((64 * U + 8) * (long long)down + up >= 0x7fffffffLL) return false;
    pl->c = (int)c;
    pl->U = (int)U;
    pl->cd = (int)cd;
    pl->R = r;
    pl->span_cap = (int)((64 * rp + 3) & ~3LL);
    pl->margin = (int)dmax + 24;
    pl->threads = 1024;
    pl->rp = (int)rp;
    pl->rw = (int)rw;
    return true;
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
  MIA_REQUIRE(pad_mode >= MIA_RS_PAD_CONSTANT && pad_mode <= MIA_RS_PAD_ANTIREFLECT,
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
  if (lds > 160 * 1024) {  // (sh_plan keeps its plans below this; checked where the launch is)
    set_error("resample: the shared-window plan for up %d, down %d needs %zu bytes of LDS", up, down, lds);
    return MIA_ERR_UNSUPPORTED;
  }
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

int mia_upfirdn_margin(int up, int down, int taps_per_phase, int64_t n_out) {
  mia::ShPlan pl;
  if (!mia::sh_plan(up, down, taps_per_phase, &pl)) return 0;
  if (n_out < 64LL * pl.U) return 0;  // not one full run per row: the per-output kernel
  return pl.margin;
}

int64_t mia_upfirdn_shared_lds_bytes(int up, int down, int taps_per_phase) {
  mia::ShPlan pl;
  if (!mia::sh_plan(up, down, taps_per_phase, &pl)) return 0;
  return (int64_t)(sizeof(float) * ((size_t)pl.span_cap +
                                    (size_t)mia::kShBt * ((pl.U + pl.R - 1) / pl.R)) + 16);
}

int mia_upfirdn_shared(const float* x, int64_t rows, int64_t n_in, const float* padded_table,
                       int margin, int up, int down, int taps_per_phase, int64_t skip,
                       int64_t n_out, int pad_mode, float* out, void* stream) {
  using namespace mia;
  MIA_REQUIRE(rows >= 0 && n_in >= 1 && n_out >= 0, "rows, n_in, n_out must be >= 0, 1, 0");
  MIA_REQUIRE(up >= 1 && down >= 1 && taps_per_phase >= 1 && skip >= 0,
              "up, down, taps_per_phase must be >= 1 and skip >= 0");
  MIA_REQUIRE(pad_mode >= MIA_RS_PAD_CONSTANT && pad_mode <= MIA_RS_PAD_ANTIREFLECT,
              "unknown padding mode %d", pad_mode);
  ShPlan pl;
  if (!sh_plan(up, down, taps_per_phase, &pl)) {
    set_error("no shared-window plan for up %d, down %d, %d taps per phase (use mia_upfirdn)", up,
              down, taps_per_phase);
    return MIA_ERR_UNSUPPORTED;
  }
  MIA_REQUIRE(margin == pl.margin, "table margin is %d, mia_upfirdn_margin says %d", margin,
              pl.margin);
  if (rows == 0 || n_out == 0) return MIA_OK;
  MIA_REQUIRE(x && padded_table && out, "null pointer");
  if ((long long)(skip + n_out) > (0x7fffffffffffffffLL / down)) {
    set_error("resample: row of %lld outputs is too long", (long long)n_out);
    return MIA_ERR_UNSUPPORTED;
  }
  const long long run = 64LL * pl.U;
  const long long runs_per_row = (n_out + run - 1) / run;
  const long long total_runs = rows * runs_per_row;
  const size_t lds = (size_t)mia_upfirdn_shared_lds_bytes(up, down, taps_per_phase);
  if (lds > 160 * 1024) {  // (sh_plan keeps its plans below this; checked where the launch is)
    set_error("resample: the shared-window plan for up %d, down %d needs %zu bytes of LDS", up, down, lds);
    return MIA_ERR_UNSUPPORTED;
  }
  int dev = 0, cus = 256;
  MIA_CHECK_HIP(hipGetDevice(&dev));
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  long long per_cu = (160 * 1024) / (long long)(lds + 256);
  if (per_cu > 2048 / pl.threads) per_cu = 2048 / pl.threads;
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  long long grid = (long long)cus * per_cu;
  if (grid > total_runs) grid = total_runs;
  const long long per_wg = (total_runs + grid - 1) / grid;
  if (per_wg > 0x7fffffffLL || runs_per_row > 0x7fffffffLL) {
    set_error("resample: %lld output runs do not fit one launch (split the batch)", total_runs);
    return MIA_ERR_UNSUPPORTED;
  }
  grid = (total_runs + per_wg - 1) / per_wg;
  void (*kern)(const float*, long long, const float*, int, int, int, int, long long, long long, int,
               float*, int, int, int, long long, long long, int, int, int) = nullptr;
  if (pl.rp) {
    kern = upfirdn_shared_kernel<5, true>;
  } else {
    kern = pl.R == 4 ? upfirdn_shared_kernel<4, false> : upfirdn_shared_kernel<5, false>;
  }
  if (lds > 64 * 1024) {
    MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(pl.threads), lds, (hipStream_t)stream, x,
                     (long long)n_in, padded_table, up, down, taps_per_phase, pl.margin,
                     (long long)skip, (long long)n_out, pad_mode, out, pl.U, pl.cd, pl.span_cap,
                     runs_per_row, total_runs, (int)per_wg, pl.rp, pl.rw);
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

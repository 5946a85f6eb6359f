// Forward transform for the n_fft that have no compiled plan: any even n_fft up to 2048 whose
// half length factors into butterflies of at most 16 points (2^a 3^b 5^c 7^d: 320, 480, 640,
// 882, 960, 1200, 1536, 1764, ...), and any such odd n_fft up to 1023 (441, 675, 945: transformed
// unpacked, with zero imaginary parts).  The reference takes any n_fft at one speed (mx.fft.rfft,
// primitives/stft.py:109-133); the whole-row engine of bigfft.hip, which served these sizes
// alone before, gathers the frames into a workspace and makes several trips through memory
// (64 x 30 s at n_fft 960: 3.9 ms against 0.15 ms for the compiled 1024-point plan).
//
// Here a frame belongs to ONE wave: its n_fft / 2 complex points sit in the wave's own two LDS
// buffers and go through Stockham passes whose radices are run-time data (a switch over the
// butterfly templates; lanes take butterflies lane, lane + L, ...), twiddles from a table of
// exp(-2 pi i k / (n_fft / 2)) in LDS.  No workgroup barrier stands in the frame loop - LDS
// operations of one wave complete in order.  Short transforms share a wave: a frame takes
// L = 1, 2, ... 64 lanes (the power of two that covers its widest pass), 64 / L frames run side
// by side in a wave, each in its own buffers - for n_fft 16 or 32 a lane holds a whole frame's
// butterfly.  Split step, spectrum rows / power tile, band projection and epilogues are those of
// the compiled plans (spectral_fwd.hip); a workgroup is four waves.  Per n log n it runs at a
// third to a half of the compiled plans' rate, which is what run-time radices and 4 - 12 waves
// per CU cost.
#include "fft_engine.h"
#include "mia_internal.h"

namespace mia {
namespace wavefft {

constexpr int WG = 256, WAVES = WG / 64, MAXFPW = 2;  // frames per wave and tile: 1 or 2
constexpr int MAXP = 6, MAXM = 1024;

struct Passes {
  int n;
  int radix[MAXP];
};

// LDS slot of logical element i: one pad slot per 16 elements, so that the strided writes of a
// pass (lane j writes j R + r, ...) spread over the banks.
__host__ __device__ __forceinline__ int pos(int i) { return i + (i >> 4); }
__host__ __device__ inline int buf_len(int M) { return M + (M >> 4) + 2; }

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// One Stockham pass of radix R after passes whose radices multiply to NS (inv_ns = 1 / NS), by
// the L lanes of one frame (li = lane within the frame).
template <int R>
__device__ __forceinline__ void pass(const float2* __restrict__ src, float2* __restrict__ dst,
                                     const float2* __restrict__ twM, int M, int NS, float inv_ns,
                                     int li, int L) {
  const int S = M / R;
  const int tstep = S / NS;  // step in the table of exp(-2 pi i k / M)
  for (int j = li; j < S; j += L) {
    // j / NS and j mod NS without an integer division: exact for j < 2^20
    const int q = NS == 1 ? j : (int)(((float)j + 0.5f) * inv_ns);
    const int jm = j - q * NS;
    float2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float2 x = src[pos(j + r * S)];
      if (r > 0 && NS > 1) x = cmul(x, twM[jm * r * tstep]);
      v[r] = x;
    }
    Dft<R>::run(v);
    const int base = q * NS * R + jm;
#pragma unroll
    for (int r = 0; r < R; ++r) dst[pos(base + r * NS)] = v[r];
  }
}

// All passes of one frame, ping-pong between its two buffers; returns the buffer that holds the
// transform in natural order.  Every lane of the wave calls it (frames side by side).
__device__ __forceinline__ const float2* transform(float2* bufA, float2* bufB, const float2* __restrict__ twM,
                                                   const Passes& ps, int M, int li, int L) {
  const float2* src = bufA;
  float2* dst = bufB;
  int NS = 1;
  for (int pi = 0; pi < ps.n; ++pi) {
    const int R = ps.radix[pi];
    const float inv_ns = 1.0f / (float)NS;
    switch (R) {
      case 2: pass<2>(src, dst, twM, M, NS, inv_ns, li, L); break;
      case 3: pass<3>(src, dst, twM, M, NS, inv_ns, li, L); break;
      case 4: pass<4>(src, dst, twM, M, NS, inv_ns, li, L); break;
      case 5: pass<5>(src, dst, twM, M, NS, inv_ns, li, L); break;
      case 6: pass<6>(src, dst, twM, M, NS, inv_ns, li, L); break;
      case 7: pass<7>(src, dst, twM, M, NS, inv_ns, li, L); break;
      case 8: pass<8>(src, dst, twM, M, NS, inv_ns, li, L); break;
      case 10: pass<10>(src, dst, twM, M, NS, inv_ns, li, L); break;
      case 12: pass<12>(src, dst, twM, M, NS, inv_ns, li, L); break;
      case 15: pass<15>(src, dst, twM, M, NS, inv_ns, li, L); break;
      default: pass<16>(src, dst, twM, M, NS, inv_ns, li, L); break;
    }
    wave_sync();
    float2* const t = const_cast<float2*>(src);
    src = dst;
    dst = t;
    NS *= R;
  }
  return src;
}

#ifndef MIA_WAVE_OCC
#define MIA_WAVE_OCC 3
#endif
// Plans compiled in: the framings of the reference's speech-enhancement model
// (models/enhance/config.py:28-31, 63-68: n_fft 1920 and 640) and common 44.1 / 48 kHz framings.  With the length, the radices and
// the lanes per frame known to the compiler the strides, twiddle steps and loop counts of
// the passes are constants and a lane's loads of a frame are issued together; the run-time form
// spends two thirds of its vector instructions on that arithmetic.  RuntimePlan: everything
// from the kernel's arguments.
struct RuntimePlan {
  static constexpr int M = 0, L = 0, OCC = MIA_WAVE_OCC;
  static constexpr bool PACKED = true;
};
// M points (n_fft / 2 packed, n_fft itself for an odd n_fft), L lanes per frame, radices.
template <int M_, int L_, bool PACKED_, int R0_, int R1_, int R2_ = 1, int R3_ = 1, int OCC_ = MIA_WAVE_OCC>
struct FixedPlan {
  static constexpr int M = M_, L = L_, R0 = R0_, R1 = R1_, R2 = R2_, R3 = R3_;
  static constexpr int OCC = OCC_;  // waves per SIMD the register budget is for
  static constexpr bool PACKED = PACKED_;
  static_assert(R0_ * R1_ * R2_ * R3_ == M_, "radices multiply to the length");
  static bool matches(const Passes& ps, int M, int L, int packed) {
    const int want[4] = {R0_, R1_, R2_, R3_};
    int n = 0;
    while (n < 4 && want[n] > 1) ++n;
    if (M != M_ || L != L_ || (packed != 0) != PACKED_ || ps.n != n) return false;
    for (int i = 0; i < n; ++i) {
      if (ps.radix[i] != want[i]) return false;
    }
    return true;
  }
};
using Plan960 = FixedPlan<960, 64, true, 16, 15, 4, 1, 2>;  // n_fft 1920 (spills under the budget of three waves)
using Plan320 = FixedPlan<320, 64, true, 8, 8, 5>;     // n_fft 640
using Plan480 = FixedPlan<480, 64, true, 12, 10, 4>;   // n_fft 960: 20 ms at 48 kHz
using Plan240 = FixedPlan<240, 64, true, 5, 4, 4, 3>;  // n_fft 480: 10 ms at 48 kHz, 30 ms at 16 kHz
using Plan441 = FixedPlan<441, 64, true, 7, 7, 3, 3>;  // n_fft 882: 20 ms at 44.1 kHz
using Plan882 = FixedPlan<882, 64, true, 7, 7, 6, 3, 2>;  // n_fft 1764: 40 ms at 44.1 kHz (0.45 -> 0.34 ms with the budget of two waves)
using Plan600 = FixedPlan<600, 64, true, 12, 10, 5>;   // n_fft 1200: 25 ms at 48 kHz
using Plan768 = FixedPlan<768, 64, true, 16, 12, 4>;   // n_fft 1536: 32 ms at 48 kHz
using Plan500 = FixedPlan<500, 64, true, 10, 10, 5>;   // n_fft 1000
using Plan640 = FixedPlan<640, 64, true, 16, 10, 4>;   // n_fft 1280: 80 ms at 16 kHz
using Plan800 = FixedPlan<800, 64, true, 16, 10, 5, 1, 2>;  // n_fft 1600: 100 ms at 16 kHz
using Plan1000 = FixedPlan<1000, 64, true, 10, 10, 10, 1, 2>;  // n_fft 2000
using Plan160 = FixedPlan<160, 16, true, 16, 10>;      // n_fft 320: 20 ms at 16 kHz, four frames side by side in a wave
using Plan441u = FixedPlan<441, 64, false, 7, 7, 3, 3>;  // n_fft 441 (odd: unpacked): 10 ms at 44.1 kHz
// The small sizes with a compiled multi-wave forward plan have no fused inverse of their own
// (inverse transform, frame workspace, overlap-add kernel: 1.1 - 1.7 ms for 256 x 30 s); their
// istft runs on these one-wave-per-frame plans instead (always 64 lanes per frame: the fused
// kernel's layout).
using Inv128 = FixedPlan<128, 64, true, 4, 4, 8>;    // n_fft 256
using Inv200 = FixedPlan<200, 64, true, 5, 5, 8>;    // n_fft 400
using Inv256w = FixedPlan<256, 64, true, 4, 8, 8>;   // n_fft 512
using Inv400w = FixedPlan<400, 64, true, 5, 8, 10>;  // n_fft 800
// (1024 and 2048 points at a hop other than n_fft / 4, which their own fused kernel does not take)
using Inv512w = FixedPlan<512, 64, true, 8, 8, 8>;           // n_fft 1024
using Inv1024w = FixedPlan<1024, 64, true, 16, 16, 4, 1, 2>;  // n_fft 2048
#define MIA_WAVE_INVERSE_ONLY_PLANS(X) X(Inv128) X(Inv200) X(Inv256w) X(Inv400w) X(Inv512w) X(Inv1024w)
#define MIA_WAVE_FIXED_PLANS(X) \
  X(Plan960) X(Plan320) X(Plan480) X(Plan240) X(Plan441) X(Plan882) X(Plan600) X(Plan768) X(Plan500) X(Plan640) \
  X(Plan800) X(Plan1000) X(Plan160) X(Plan441u)

// One pass of a compiled-in plan IN PLACE: a lane first reads the points of all its butterflies
// (ceil(M / R / L) of them, L lanes per frame), then writes their results to the natural-order slots of the same
// buffer.  The frame belongs to one wave, whose LDS operations complete in order, so every read
// of the pass precedes every write: half the LDS of a ping-pong pair.
template <int R, int M, int NS, int L>
__device__ __forceinline__ void pass_inplace(float2* __restrict__ buf, const float2* __restrict__ twM, int li) {
  constexpr int S = M / R, NB = (S + L - 1) / L, tstep = S / NS;
  float2 v[NB][R];
  int base[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int j = li + b * L;
    if (j < S) {
      const int q = j / NS, jm = j - q * NS;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float2 x = buf[pos(j + r * S)];
        if (r > 0 && NS > 1) x = cmul(x, twM[jm * r * tstep]);
        v[b][r] = x;
      }
      base[b] = q * NS * R + jm;
    }
  }
  wave_sync();  // keeps the compiler from moving a write above another butterfly's reads
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (li + b * L < S) {
      Dft<R>::run(v[b]);
#pragma unroll
      for (int r = 0; r < R; ++r) buf[pos(base[b] + r * NS)] = v[b][r];
    }
  }
  wave_sync();
}
template <class PL>
__device__ __forceinline__ void transform_inplace(float2* buf, const float2* __restrict__ twM, int li) {
  pass_inplace<PL::R0, PL::M, 1, PL::L>(buf, twM, li);
  pass_inplace<PL::R1, PL::M, PL::R0, PL::L>(buf, twM, li);
  if constexpr (PL::R2 > 1) pass_inplace<PL::R2, PL::M, PL::R0 * PL::R1, PL::L>(buf, twM, li);
  if constexpr (PL::R3 > 1) pass_inplace<PL::R3, PL::M, PL::R0 * PL::R1 * PL::R2, PL::L>(buf, twM, li);
}

// The passes of a frame: a compiled-in plan in place in bufA, the run-time form between bufA and
// bufB; returns the buffer that holds the transform.
template <class PL>
__device__ __forceinline__ const float2* transform_plan(float2* bufA, float2* bufB,
                                                        const float2* __restrict__ twM, const Passes& ps,
                                                        int M, int li, int L) {
  if constexpr (PL::M == 0) {
    return transform(bufA, bufB, twM, ps, M, li, L);
  } else {
    transform_inplace<PL>(bufA, twM, li);
    return bufA;
  }
}

template <int MODE, class PL>  // 0: spectrum rows, 1: mel (B, n_mels, F), 3: mel (B, F, n_mels)
__global__ __launch_bounds__(WG, PL::OCC) void fwd_wave_kernel(FwdParams p, Passes ps, int M_arg, int FPW,
                                                                    int packed_arg, int L_arg) {
  const int M = PL::M ? PL::M : M_arg, packed = PL::M ? (PL::PACKED ? 1 : 0) : packed_arg, L = PL::M ? PL::L : L_arg;
  // packed: n_fft is even, M = n_fft / 2 points z[n] = x[2 n] + i x[2 n + 1] and a split step;
  // else n_fft is odd, M = n_fft points with zero imaginary parts, bins 0 .. (M - 1) / 2 as they come.
  // L lanes per frame, GF = 64 / L frames side by side in a wave, FPW such rounds per tile.
  constexpr bool MEL = MODE != 0;
  const int GF = 64 / L;
  const int FT = WAVES * FPW * GF;  // frames per tile (a power of two)
  const int N = packed ? 2 * M : M;
  const int BINS = N / 2 + 1;
  const int LB = buf_len(M);
  const int PS = BINS | 1;  // odd row stride of the power tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int NBUF = PL::M ? 1 : 2;                             // a compiled-in plan transforms in place
  float2* const bufs = reinterpret_cast<float2*>(smem_raw);      // [WAVES][GF][NBUF][LB]
  float2* const twM = bufs + WAVES * GF * NBUF * LB;              // [M]
  float* const P = reinterpret_cast<float*>(twM + M);             // [FT][PS] (mel modes)
  float* const lds_wts = P + (MEL ? FT * PS : 0);
  int* const lds_rows = reinterpret_cast<int*>(lds_wts + (MEL ? p.nnz : 0));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & (L - 1), g = lane / L;  // lane within the frame, frame of the wave's round
  for (int k = tid; k < M; k += WG) twM[k] = p.tw[packed ? 2 * k : k];
  if constexpr (MEL) {
    if (p.lds_tables) {
      for (int i = tid; i < p.nnz; i += WG) lds_wts[i] = p.wts[i];
      for (int i = tid; i < p.n_mels; i += WG) {
        lds_rows[i] = p.row_lo[i];
        lds_rows[p.n_mels + i] = p.row_len[i];
        lds_rows[2 * p.n_mels + i] = p.row_off[i];
      }
    }
  }
  __syncthreads();

  float2* const bufA = bufs + (wave * GF + g) * NBUF * LB;
  float2* const bufB = bufA + (NBUF - 1) * LB;
  const float2* __restrict__ win2 = reinterpret_cast<const float2*>(p.win);
  const float half_scale = 0.5f * p.scale;

  const long long total = p.B * p.tiles;
  const long long first = (long long)blockIdx.x * p.tiles_per_wg;
  const long long last = first + p.tiles_per_wg < total ? first + p.tiles_per_wg : total;
  for (long long idx = first; idx < last; ++idx) {
    const long long b = (long long)((unsigned)idx / (unsigned)p.tiles);  // the launcher keeps the tile count below 2^32
    const int tile = (int)(idx - b * p.tiles);
    const long long f0 = (long long)tile * FT;
    const float* __restrict__ yb = p.y + b * p.T;

    for (int i = 0; i < FPW; ++i) {
      const int fl = (wave * FPW + i) * GF + g;
      const long long f = f0 + fl;
      if (f0 + (long long)(wave * FPW + i) * GF >= p.F) break;  // wave-uniform: the round has no frame
      const bool valid = f < p.F;  // per frame (lanes of a frame past the clip's end idle along)
      // ---- windowed sample pairs z[n] = x[2 n] + i x[2 n + 1] --------------------------------
      const long long s0 = f * p.hop - p.pad;
      const bool interior = s0 >= 0 && s0 + N <= p.T;
      if (!valid) {
        // nothing to load: the passes run on whatever the buffers hold, nothing is stored
      } else if (!packed) {
        for (int n = li; n < M; n += L) {
          const float x = interior ? yb[s0 + n] : fetch_sample(yb, p.T, s0 + n, p.pad_mode);
          bufA[pos(n)] = {x * p.win[n], 0.0f};
        }
      } else if (interior && (reinterpret_cast<uintptr_t>(yb + s0) & 7) == 0) {
        const float2* __restrict__ src = reinterpret_cast<const float2*>(yb + s0);
#pragma unroll 4
        for (int n = li; n < M; n += L) {
          const float2 x = src[n], w = win2[n];
          bufA[pos(n)] = {x.x * w.x, x.y * w.y};
        }
      } else if (interior) {
#pragma unroll 4
        for (int n = li; n < M; n += L) {
          const float2 w = win2[n];
          bufA[pos(n)] = {yb[s0 + 2 * n] * w.x, yb[s0 + 2 * n + 1] * w.y};
        }
      } else {
        for (int n = li; n < M; n += L) {
          const float2 w = win2[n];
          bufA[pos(n)] = {fetch_sample(yb, p.T, s0 + 2 * n, p.pad_mode) * w.x,
                          fetch_sample(yb, p.T, s0 + 2 * n + 1, p.pad_mode) * w.y};
        }
      }
      wave_sync();
      const float2* src = transform_plan<PL>(bufA, bufB, twM, ps, M, li, L);
      // ---- split step: bins k and M - k from Z[k] and Z[M - k] --------------------------------
      // E = U + conj(V);  O = -i (U - conj(V));  X[k] = (E + w O) / 2,  X[M - k] = conj(E - w O) / 2
      float2* __restrict__ crow = MODE == 0 ? p.cout + (b * p.F + f) * (long long)BINS : nullptr;
      float* __restrict__ prow = P + fl * PS;
      if (valid && !packed) {  // the transform of the real frame itself: its first half is the spectrum
        for (int k = li; k < BINS; k += L) {
          const float2 X = src[pos(k)];
          if constexpr (MODE == 0) {
            crow[k] = {X.x * p.scale, X.y * p.scale};
          } else {
            prow[k] = X.x * X.x + X.y * X.y;
          }
        }
      }
#pragma unroll 4
      for (int k = li; valid && packed && 2 * k <= M; k += L) {
        const float2 U = src[pos(k)];
        const float2 V = src[pos(k == 0 ? 0 : M - k)];
        const float2 w = p.tw[k];
        const float2 E = {U.x + V.x, U.y - V.y};
        const float2 O = {U.y + V.y, V.x - U.x};
        const float2 WO = cmul(O, w);
        const float2 X1 = {E.x + WO.x, E.y + WO.y};
        const float2 X2 = {E.x - WO.x, WO.y - E.y};
        if constexpr (MODE == 0) {
          crow[k] = {X1.x * half_scale, X1.y * half_scale};
          if (2 * k != M) crow[M - k] = {X2.x * half_scale, X2.y * half_scale};
        } else {
          prow[k] = 0.25f * (X1.x * X1.x + X1.y * X1.y);
          if (2 * k != M) prow[M - k] = 0.25f * (X2.x * X2.x + X2.y * X2.y);
        }
      }
      wave_sync();  // the next round overwrites the buffers
    }

    if constexpr (MEL) {
      __syncthreads();
      const long long frem = p.F - f0;
      const int fcount = frem < FT ? (int)frem : FT;
      if (p.power_mode != 0) {  // magnitudes or |X|^p instead of powers
        for (int i = tid; i < fcount * PS; i += WG) {
          const int k = i % PS;
          if (k < BINS) {
            const float mag = sqrtf(P[i]);
            P[i] = p.power_mode == 1 ? mag : powf(mag, p.power);
          }
        }
        __syncthreads();
      }
      const int ft_shift = 31 - __builtin_clz((unsigned)FT);
      int kmax = kKeyMinusInf;  // integer keys: a NaN value wins the maximum, as in the reference
      float vmin = INFINITY;
      float* const stage = reinterpret_cast<float*>(bufs);  // frame-major rows leave through the idle FFT buffers
      auto project = [&](const int* row_lo, const int* row_len, const int* row_off, const float* wts) {
        for (int idx2 = tid; idx2 < p.n_mels * FT; idx2 += WG) {
          const int m = idx2 >> ft_shift;  // FT is a power of two
          const int fl = idx2 & (FT - 1);
          if (fl >= fcount) continue;
          const int lo = row_lo[m], len = row_len[m];
          const float* w = wts + row_off[m];
          const float* __restrict__ pr = P + fl * PS + lo;
          float acc = 0.0f;
#pragma unroll 4
          for (int j = 0; j < len; ++j) acc = fmaf(w[j], pr[j], acc);
          float outv = acc;
          if (p.epi == MIA_EPI_WHISPER) {
            // values at or below the 1e-10 floor give exactly log10(1e-10) = -10
            const float yv = acc <= 1e-10f ? -10.0f : __log10f(acc);
            kmax = max(kmax, float_to_key_nan(yv));
            outv = (yv + 4.0f) * 0.25f;
            vmin = fminf(vmin, outv);
          } else if (p.epi == MIA_EPI_DB) {
            outv = acc <= p.amin ? p.db_floor : p.coef * __log10f(acc * p.inv_ref);
            kmax = max(kmax, float_to_key_nan(outv));
            vmin = fminf(vmin, outv);
          }
          if (MODE == 3) {
            stage[fl * p.n_mels + m] = outv;
          } else {
            p.mout[(b * p.n_mels + m) * p.F + f0 + fl] = outv;
          }
        }
      };
      if (p.lds_tables) {
        project(lds_rows, lds_rows + p.n_mels, lds_rows + 2 * p.n_mels, lds_wts);
      } else {
        project(p.row_lo, p.row_len, p.row_off, p.wts);
      }
      if (MODE == 3) {
        __syncthreads();
        float* __restrict__ dstp = p.mout + (b * p.F + f0) * p.n_mels;  // 16-byte aligned: n_mels % 4 == 0
        const int n4 = fcount * p.n_mels / 4;
        for (int i = tid; i < n4; i += WG) {
          reinterpret_cast<float4*>(dstp)[i] = reinterpret_cast<const float4*>(stage)[i];
        }
      }
      if (p.epi != MIA_EPI_NONE && p.clip_max_key != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          kmax = max(kmax, __shfl_xor(kmax, off));
          vmin = fminf(vmin, __shfl_xor(vmin, off));
        }
        __shared__ int red_max[WAVES];
        __shared__ float red_min[WAVES];
        if (lane == 0) {
          red_max[wave] = kmax;
          red_min[wave] = vmin;
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
          for (int w = 1; w < WAVES; ++w) {
            kmax = max(kmax, red_max[w]);
            vmin = fminf(vmin, red_min[w]);
          }
          atomicMax(p.clip_max_key + b, kmax);
          atomicMin(p.clip_min_key + b, float_to_key(vmin));
        }
      }
      __syncthreads();  // the next tile overwrites the power tile, the buffers and the reduction slots
    }
  }
}

// The inverse: bins of one frame -> n_fft real samples (scaled by 1 / n_fft) in the frame
// workspace the overlap-add kernel reads, a frame per wave.  The inverse transform runs as the
// forward one on conjugated input (sum_k Z[k] e^{+2 pi i k n / M} = conj(FFT(conj Z))[n]); for
// an even n_fft the bins are first folded into the half-length sequence Z[k] = E + i O,
// E = (X[k] + conj X[M - k]) / 2, O = conj(w_k) (X[k] - conj X[M - k]) / 2, the inverse of the
// forward kernel's split step; the imaginary parts of X[0] and X[n_fft / 2] are ignored, as
// they are by the compiled plans.
template <class PL>
__global__ __launch_bounds__(WG, PL::OCC) void inv_wave_kernel(InvParams p, Passes ps, int M_arg,
                                                                    int packed_arg, int L_arg,
                                                                    long long frames_per_wg) {
  const int M = PL::M ? PL::M : M_arg, packed = PL::M ? (PL::PACKED ? 1 : 0) : packed_arg, L = PL::M ? PL::L : L_arg;
  const int GF = 64 / L;
  const int N = packed ? 2 * M : M;
  const int LB = buf_len(M);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int NBUF = PL::M ? 1 : 2;                         // a compiled-in plan transforms in place
  float2* const bufs = reinterpret_cast<float2*>(smem_raw);  // [WAVES][GF][NBUF][LB]
  float2* const twM = bufs + WAVES * GF * NBUF * LB;          // [M]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & (L - 1), g = lane / L;
  for (int k = tid; k < M; k += WG) twM[k] = p.tw[packed ? 2 * k : k];
  __syncthreads();
  float2* const bufA = bufs + (wave * GF + g) * NBUF * LB;
  float2* const bufB = bufA + (NBUF - 1) * LB;
  const long long total = p.B * p.F;
  const long long first = (long long)blockIdx.x * frames_per_wg;
  const long long last = first + frames_per_wg < total ? first + frames_per_wg : total;
  const float inv_n = 1.0f / (float)N;
  for (long long base = first + (long long)wave * GF; base < last; base += WAVES * GF) {
    const long long fi = base + g;
    const bool valid = fi < last;
    const long long b = valid ? fi / p.F : 0, f = valid ? fi - b * p.F : 0;
    const float2* __restrict__ xs = p.spec + b * p.sb + f * p.sf;
    if (valid && packed) {
#pragma unroll 4
      for (int k = li; 2 * k <= M; k += L) {
        float2 xk = xs[(long long)k * p.sk], xm = xs[(long long)(M - k) * p.sk];
        if (k == 0) {
          xk.y = 0.0f;
          xm.y = 0.0f;
        }
        const float2 w = p.tw[k];
        const float2 E = {xk.x + xm.x, xk.y - xm.y};
        const float2 D = {xk.x - xm.x, xk.y + xm.y};
        const float2 O = {D.x * w.x + D.y * w.y, D.y * w.x - D.x * w.y};  // D conj(w)
        // twice Z[k] and Z[M - k]; stored conjugated for the forward passes
        bufA[pos(k)] = {E.x - O.y, -(E.y + O.x)};
        if (k != 0 && 2 * k != M) bufA[pos(M - k)] = {E.x + O.y, E.y - O.x};
      }
    } else if (valid) {
      for (int k = li; 2 * k <= M; k += L) {  // Hermitian extension of bins 0 .. (M - 1) / 2
        const float2 x = xs[(long long)k * p.sk];
        if (k == 0) {
          bufA[pos(0)] = {x.x, 0.0f};
        } else {
          bufA[pos(k)] = {x.x, -x.y};
          bufA[pos(M - k)] = {x.x, x.y};
        }
      }
    }
    wave_sync();
    const float2* src = transform_plan<PL>(bufA, bufB, twM, ps, M, li, L);
    float* __restrict__ orow = p.frames + fi * (long long)N;
    if (valid && packed) {  // (x[2 n], x[2 n + 1]) = conj(y[n]) / (2 M): the sequence above was twice Z
      for (int n = li; n < M; n += L) {
        const float2 y = src[pos(n)];
        reinterpret_cast<float2*>(orow)[n] = {y.x * inv_n, -y.y * inv_n};
      }
    } else if (valid) {
      for (int n = li; n < M; n += L) orow[n] = src[pos(n)].x * inv_n;
    }
    wave_sync();  // the next round overwrites the buffers
  }
}

// istft in one launch for the compiled-in plans (one frame per wave): inverse transform and
// windowed overlap-add without the frame workspace (n_fft / hop times the signal, written and
// read once by the two-kernel form).  A workgroup walks runs of consecutive frames of one row,
// four frames (one per wave) a round.  After a round's transforms every thread takes samples of
// the span the four frames cover, adds their windowed values to the sample's slot of an LDS ring
// in ascending frame order - the order and the fused multiply-adds of ola_kernel, so the result
// is bit-identical to the two-kernel form - and samples no later frame reaches are divided by
// their window envelope and leave.  A run starts ceil(n_fft / hop) - 1 frames early (their
// transforms are repeated) so that its first samples see all their frames.
template <class PL>
__global__ __launch_bounds__(WG, 2) void istft_wave_fused_kernel(InvParams p, int frames_per_run,
                                                                 long long runs_per_row, long long total_runs,
                                                                 long long runs_per_wg) {
  constexpr int M = PL::M, N = 2 * M, LB = M + (M >> 4) + 2;  // LB = buf_len(M)
  const int hop = p.hop;
  const int RING = N + (WAVES - 1) * hop;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2* const bufs = reinterpret_cast<float2*>(smem_raw);  // [WAVES][LB]
  float2* const twM = bufs + WAVES * LB;                      // [M]
  float2* const twH = twM + M;                                // [M / 2 + 1]
  float* const win = reinterpret_cast<float*>(twH + M / 2 + 1);  // [N]
  float* const acc = win + N;                                 // [RING]
  float* const env = acc + RING;                              // [hop]: window envelope of a sample all of whose frames exist
  const int tid = threadIdx.x, li = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int k = tid; k < M; k += WG) twM[k] = p.tw[2 * k];
  for (int k = tid; k <= M / 2; k += WG) twH[k] = p.tw[k];
  for (int k = tid; k < N; k += WG) win[k] = p.win[k];
  // sum of w^2 over the frames that cover position r of a hop, in ascending frame order
  // (descending position in the frame), as ola_kernel accumulates it
  for (int r = tid; r < hop; r += WG) {
    float wsum = 0.0f;
    for (int n = r + ((N - 1 - r) / hop) * hop; n >= 0; n -= hop) {
      const float wv = p.win[n];
      wsum = fmaf(wv, wv, wsum);
    }
    env[r] = wsum;
  }
  float2* const buf = bufs + wave * LB;
  const float inv_n = 1.0f / (float)N;
  const int reach = (N + hop - 1) / hop - 1;  // earlier frames that reach into a frame's first samples
  const long long run0 = (long long)blockIdx.x * runs_per_wg;
  const long long run1 = run0 + runs_per_wg < total_runs ? run0 + runs_per_wg : total_runs;
  for (long long run = run0; run < run1; ++run) {
    const long long b = run / runs_per_row;
    const long long fa = (run - b * runs_per_row) * frames_per_run;
    const long long fb = fa + frames_per_run < p.F ? fa + frames_per_run : p.F;
    const long long own_lo = fa * hop;
    const long long own_hi = fb == p.F ? p.out_len : (fb * hop < p.out_len ? fb * hop : p.out_len);
    const long long fstart = fa > reach ? fa - reach : 0;
    __syncthreads();  // the previous run is done with the ring (and the tables are loaded)
    for (int q = tid; q < RING; q += WG) acc[q] = 0.0f;
    int origin = 0;  // ring slot of the round's first sample
    float* __restrict__ orow = p.out + b * p.out_len;
    for (long long f0 = fstart; f0 < fb; f0 += WAVES) {
      const long long f = f0 + wave;
      if (f < fb) {  // wave-uniform
        const float2* __restrict__ xs = p.spec + b * p.sb + f * p.sf;
#pragma unroll 4
        for (int k = li; 2 * k <= M; k += 64) {
          float2 xk = xs[(long long)k * p.sk], xm = xs[(long long)(M - k) * p.sk];
          if (k == 0) {
            xk.y = 0.0f;
            xm.y = 0.0f;
          }
          const float2 w = twH[k];
          const float2 E = {xk.x + xm.x, xk.y - xm.y};
          const float2 D = {xk.x - xm.x, xk.y + xm.y};
          const float2 O = {D.x * w.x + D.y * w.y, D.y * w.x - D.x * w.y};  // D conj(w)
          buf[pos(k)] = {E.x - O.y, -(E.y + O.x)};
          if (k != 0 && 2 * k != M) buf[pos(M - k)] = {E.x + O.y, E.y - O.x};
        }
        wave_sync();
        transform_inplace<PL>(buf, twM, li);
      }
      __syncthreads();
      const int nfr = fb - f0 < WAVES ? (int)(fb - f0) : WAVES;  // frames of this round
      const bool final_round = f0 + WAVES >= fb;
      const long long P0 = f0 * hop;
      const int span = N + (nfr - 1) * hop;
      float* __restrict__ obase = orow + P0;
      // the round's samples this run owns, and those that are complete after it, as offsets from P0
      const int lo_own = own_lo > P0 ? (int)(own_lo - P0 < span ? own_lo - P0 : span) : 0;
      const int hi_own = own_hi > P0 ? (int)(own_hi - P0 < span ? own_hi - P0 : span) : 0;
      const int emit_hi = final_round ? span : WAVES * hop;
      // every sample of the round has all its frames (none clipped at the row's first or last frame)
      const bool interior = P0 >= N - 1 && (P0 + span - 1) / hop <= p.F - 1;
      auto envelope = [&](int j) {  // of a sample some of whose frames do not exist (the row's ends)
        const long long s = P0 + j;
        long long first = s - N + 1;
        first = first <= 0 ? 0 : (first + hop - 1) / hop;
        long long last = s / hop;
        if (last > p.F - 1) last = p.F - 1;
        float wsum = 0.0f;
        for (long long ff = first; ff <= last; ++ff) {
          const float wv = win[(int)(s - ff * hop)];
          wsum = fmaf(wv, wv, wsum);
        }
        return wsum;
      };
      if ((hop & 1) == 0) {
        // an even hop: a thread takes two neighbouring samples, which are the two halves of one
        // complex point in every frame that covers them (ring, window and envelope read in pairs)
        const float2* __restrict__ win2 = reinterpret_cast<const float2*>(win);
        const float2* __restrict__ env2 = reinterpret_cast<const float2*>(env);
        float2* __restrict__ acc2 = reinterpret_cast<float2*>(acc);
        const int rstep = (2 * WG) % hop;
        int r = (2 * tid) % hop;  // (P0 + j) mod hop: P0 is a multiple of hop
        for (int j = 2 * tid; j < span; j += 2 * WG) {
          int slot = origin + j;
          if (slot >= RING) slot -= RING;
          float2 sum = acc2[slot >> 1];
#pragma unroll
          for (int w = 0; w < WAVES; ++w) {
            const int n = j - w * hop;
            if (w < nfr && n >= 0 && n < N) {
              const float2 y = bufs[w * LB + pos(n >> 1)];
              const float2 wv = win2[n >> 1];
              sum.x = fmaf(wv.x, y.x * inv_n, sum.x);
              sum.y = fmaf(wv.y, -y.y * inv_n, sum.y);
            }
          }
          if (j < emit_hi) {  // no later frame of the run reaches these samples
            float2 ws = {0.0f, 0.0f};
            if (interior) ws = env2[r >> 1];
            if (j >= lo_own && j < hi_own) {
              obase[j] = sum.x / fmaxf(interior ? ws.x : envelope(j), 1e-8f) * p.scale;
            }
            if (j + 1 >= lo_own && j + 1 < hi_own) {
              obase[j + 1] = sum.y / fmaxf(interior ? ws.y : envelope(j + 1), 1e-8f) * p.scale;
            }
            acc2[slot >> 1] = {0.0f, 0.0f};
          } else {
            acc2[slot >> 1] = sum;
          }
          r += rstep;
          if (r >= hop) r -= hop;
        }
      } else {
        const int rstep = WG % hop;
        int r = tid % hop;
        for (int j = tid; j < span; j += WG) {
          int slot = origin + j;
          if (slot >= RING) slot -= RING;
          float sum = acc[slot];
#pragma unroll
          for (int w = 0; w < WAVES; ++w) {
            const int n = j - w * hop;
            if (w < nfr && n >= 0 && n < N) {
              const float2 y = bufs[w * LB + pos(n >> 1)];
              const float x = (n & 1) ? -y.y * inv_n : y.x * inv_n;
              sum = fmaf(win[n], x, sum);
            }
          }
          if (j < emit_hi) {  // no later frame of the run reaches this sample
            if (j >= lo_own && j < hi_own) {
              obase[j] = sum / fmaxf(interior ? env[r] : envelope(j), 1e-8f) * p.scale;
            }
            acc[slot] = 0.0f;
          } else {
            acc[slot] = sum;
          }
          r += rstep;
          if (r >= hop) r -= hop;
        }
      }
      origin += WAVES * hop;
      while (origin >= RING) origin -= RING;
      __syncthreads();  // the next round overwrites the frames
    }
    // samples behind the last frame (an out_len beyond the frames' reach): no frame, zero
    if (fb == p.F) {
      const long long covered = (p.F - 1) * hop + N;
      for (long long s = covered + tid; s < p.out_len; s += WG) orow[s] = 0.0f;
    }
  }
}

static size_t lds_bytes(int M, int bins, int L, int mode, int n_mels, int nnz, bool with_tables, int fpw,
                        int nbuf) {
  const int gf = 64 / L;
  size_t b = sizeof(float2) * ((size_t)WAVES * gf * nbuf * buf_len(M) + M);
  if (mode != 0) {
    b += sizeof(float) * (size_t)(WAVES * fpw * gf) * (bins | 1);
    if (with_tables) b += sizeof(float) * (size_t)nnz + 3 * sizeof(int) * (size_t)n_mels;
  }
  return b;
}

}  // namespace wavefft

// The butterflies of the plan for n_fft and the lanes a frame takes, or false: a (half) length
// above 1024, a prime factor above 7 or more than six passes.  Among the ordered factorisations
// into radices 2 .. 16 the cheapest per frame by a count of wave instructions: a pass of radix R
// costs its rounds (butterflies / lanes per frame, rounded up) times the instructions of one
// round (loads, twiddle products and stores per point plus the butterfly itself), and a frame
// that needs only L of the 64 lanes shares the wave with 64 / L - 1 others.  The first pass has
// no twiddles, so the largest radix goes first.
bool forward_wave_plan(int n_fft, int* n_pass, int radix[6], int* lanes_per_frame) {
  using namespace wavefft;
  if (n_fft < 3 || (n_fft % 2 == 0 ? n_fft / 2 : n_fft) > MAXM) return false;
  static const int allowed[] = {16, 15, 12, 10, 8, 7, 6, 5, 4, 3, 2};
  static const int bfly[] = {170, 220, 110, 100, 60, 90, 45, 40, 16, 16, 4};  // instructions of Dft<R>
  constexpr int NA = (int)(sizeof(allowed) / sizeof(allowed[0]));
  const int M = n_fft % 2 == 0 ? n_fft / 2 : n_fft;  // an odd n_fft is transformed unpacked
  int best[MAXP], best_n = 0, best_l = 64, cur[MAXP];
  double best_cost = -1.0;
  // frames side by side in a wave each need their two buffers: at most 48 KB of them per workgroup,
  // so that the power tile and a second workgroup still fit the CU
  int lmin = 64;
  while (lmin > 1 && (size_t)(64 / (lmin / 2)) * buf_len(M) <= 768) lmin /= 2;
  auto lanes_of = [&](int n) {
    int widest = 1;
    for (int i = 0; i < n; ++i) widest = M / cur[i] > widest ? M / cur[i] : widest;
    int l = lmin;
    while (l < widest && l < 64) l *= 2;
    return l;
  };
  auto cost_of = [&](int n, int l) {
    double c = 0;
    for (int i = 0; i < n; ++i) {
      int a = 0;
      while (allowed[a] != cur[i]) ++a;
      const int rounds = (M / cur[i] + l - 1) / l;
      c += (double)rounds * ((i == 0 ? 6 : 10) * cur[i] + bfly[a]) + 12;  // + per-pass set-up and sync
    }
    return c * l / 64.0;  // 64 / l frames share the wave's instructions
  };
  // depth-first over non-increasing radix sequences (the order of the passes does not change the cost)
  int idx[MAXP + 1] = {0};
  int rem[MAXP + 1];
  rem[0] = M;
  int d = 0;
  while (d >= 0) {
    if (rem[d] == 1) {
      if (d >= 1) {
        const int l = lanes_of(d);
        const double c = cost_of(d, l);
        if (best_cost < 0 || c < best_cost) {
          best_cost = c;
          best_n = d;
          best_l = l;
          for (int i = 0; i < d; ++i) best[i] = cur[i];
        }
      }
      --d;
      continue;
    }
    if (d == MAXP || idx[d] >= NA) {
      --d;
      continue;
    }
    const int a = idx[d]++;
    if (rem[d] % allowed[a] != 0) continue;
    cur[d] = allowed[a];
    rem[d + 1] = rem[d] / allowed[a];
    idx[d + 1] = a;  // non-increasing
    ++d;
  }
  if (best_cost < 0) return false;
  for (int i = 0; i < best_n; ++i) radix[i] = best[i];
  *n_pass = best_n;
  if (lanes_per_frame) *lanes_per_frame = best_l;
  return true;
}

// Buffers per frame: one for a compiled-in plan (its passes run in place), two otherwise.
static int wave_buffers(const wavefft::Passes& ps, int M, int L, int packed) {
  using namespace wavefft;
  int nbuf = 2;
#define MIA_HAS(PLAN) \
  if (PLAN::matches(ps, M, L, packed)) nbuf = 1;
  MIA_WAVE_FIXED_PLANS(MIA_HAS)
#undef MIA_HAS
  return nbuf;
}

// Does a compiled-in plan serve this n_fft (1), or the run-time form (0)?  -1: not a per-wave size.
int forward_wave_fixed_plan(int n_fft) {
  using namespace wavefft;
  Passes ps{};
  int L = 64;
  if (!forward_wave_plan(n_fft, &ps.n, ps.radix, &L)) return -1;
  const int packed = n_fft % 2 == 0 ? 1 : 0;
  return wave_buffers(ps, packed ? n_fft / 2 : n_fft, L, packed) == 1 ? 1 : 0;
}

// Frame-major rows leave through the FFT buffers: a tile's n_mels x frames values must fit them
// (true for every long transform; short ones, which put many frames into a tile, only with few bands).
bool forward_wave_frame_major_ok(int n_fft, int n_mels) {
  using namespace wavefft;
  int np = 0, radix[MAXP], l = 64;
  if (!forward_wave_plan(n_fft, &np, radix, &l)) return false;
  const int M = n_fft % 2 == 0 ? n_fft / 2 : n_fft;
  const size_t gf = 64 / l;
  return n_mels % 4 == 0 && sizeof(float) * (size_t)(WAVES * MAXFPW) * gf * n_mels <=
                                sizeof(float2) * (size_t)WAVES * gf * buf_len(M);  // (one buffer per frame: what every plan has)
}

int launch_forward_wave(const mia_plan* plan, const FwdParams& p0, int mode, hipStream_t s) {
  using namespace wavefft;
  Passes ps{};
  int L = 64;
  if (!forward_wave_plan(plan->n_fft, &ps.n, ps.radix, &L)) {
    set_error("n_fft=%d is not served by the per-wave transform", plan->n_fft);
    return MIA_ERR_UNSUPPORTED;
  }
  const int packed = plan->n_fft % 2 == 0 ? 1 : 0;
  const int M = packed ? plan->n_fft / 2 : plan->n_fft;
  const int bins = plan->n_fft / 2 + 1;
  FwdParams p = p0;
  const int kmode = mode == 0 ? 0 : (p.frame_major ? 3 : 1);
  if (kmode == 3 && !forward_wave_frame_major_ok(plan->n_fft, p.n_mels)) {
    set_error("frame-major mel rows do not fit this plan's FFT buffers");
    return MIA_ERR_UNSUPPORTED;
  }
  // Frames per wave and tile: two (a tile of eight frames: 32-byte runs of a mel row), or one
  // where that lets a second workgroup onto the CU (long transforms with a power tile).
  const int nbuf = wave_buffers(ps, M, L, packed);
  int fpw = MAXFPW;
  if (kmode != 0 && lds_bytes(M, bins, L, kmode, p.n_mels, p.nnz, false, 2, nbuf) > 80 * 1024 &&
      lds_bytes(M, bins, L, kmode, p.n_mels, p.nnz, false, 1, nbuf) <= 80 * 1024) {
    fpw = 1;
  }
  const int FT = WAVES * fpw * (64 / L);
  p.tiles = (int)((p.F + FT - 1) / FT);
  const long long total = p.B * p.tiles;
  if (total <= 0) return MIA_OK;
  if (total >= 0xffffffffLL) {  // the kernel divides tile indices in 32 bits
    set_error("%lld tiles do not fit one launch (split the batch)", total);
    return MIA_ERR_UNSUPPORTED;
  }
  size_t lds = lds_bytes(M, bins, L, kmode, p.n_mels, p.nnz, false, fpw, nbuf);
  p.lds_tables = 0;
  if (kmode != 0 && lds_bytes(M, bins, L, kmode, p.n_mels, p.nnz, true, fpw, nbuf) <= 80 * 1024) {
    lds = lds_bytes(M, bins, L, kmode, p.n_mels, p.nnz, true, fpw, nbuf);  // band tables ride along while two workgroups fit a CU
    p.lds_tables = 1;
  }
  if (lds > 160 * 1024) {
    set_error("n_fft=%d: %zu bytes of LDS", plan->n_fft, lds);
    return MIA_ERR_UNSUPPORTED;
  }
  auto pick = [&](auto plan_tag) {
    using PL = decltype(plan_tag);
    return kmode == 0 ? fwd_wave_kernel<0, PL> : (kmode == 1 ? fwd_wave_kernel<1, PL> : fwd_wave_kernel<3, PL>);
  };
  auto kern = pick(RuntimePlan{});
#define MIA_PICK(PLAN) \
  if (PLAN::matches(ps, M, L, packed)) kern = pick(PLAN{});
  MIA_WAVE_FIXED_PLANS(MIA_PICK)
#undef MIA_PICK
  if (lds > 64 * 1024) {
    MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  int per_cu = 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), WG, lds) !=
          hipSuccess || per_cu < 1) {
    per_cu = 1;
  }
  long long grid = (long long)(plan->cus > 0 ? plan->cus : 256) * per_cu;
  if (grid > total) grid = total;
  const long long per_wg = (total + grid - 1) / grid;
  if (per_wg > 0x7fffffffLL) {
    set_error("batch too large for one launch: %lld tiles (split the batch)", total);
    return MIA_ERR_UNSUPPORTED;
  }
  p.tiles_per_wg = (int)per_wg;
  grid = (total + per_wg - 1) / per_wg;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WG), lds, s, p, ps, M, fpw, packed, L);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int launch_inverse_wave(const mia_plan* plan, const InvParams& p, hipStream_t s) {
  using namespace wavefft;
  Passes ps{};
  int L = 64;
  if (!forward_wave_plan(plan->n_fft, &ps.n, ps.radix, &L)) {
    set_error("n_fft=%d is not served by the per-wave transform", plan->n_fft);
    return MIA_ERR_UNSUPPORTED;
  }
  const int packed = plan->n_fft % 2 == 0 ? 1 : 0;
  const int M = packed ? plan->n_fft / 2 : plan->n_fft;
  const int per_round = WAVES * (64 / L);  // frames a workgroup takes at a time
  const long long total = p.B * p.F;
  if (total <= 0) return MIA_OK;
  const size_t lds = sizeof(float2) * ((size_t)per_round * wave_buffers(ps, M, L, packed) * buf_len(M) + M);
  auto kern = inv_wave_kernel<RuntimePlan>;
#define MIA_PICK(PLAN) \
  if (PLAN::matches(ps, M, L, packed)) kern = inv_wave_kernel<PLAN>;
  MIA_WAVE_FIXED_PLANS(MIA_PICK)
#undef MIA_PICK
  if (lds > 64 * 1024) {
    MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  int per_cu = 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), WG, lds) !=
          hipSuccess || per_cu < 1) {
    per_cu = 1;
  }
  long long grid = (long long)(plan->cus > 0 ? plan->cus : 256) * per_cu;
  const long long groups = (total + per_round - 1) / per_round;
  if (grid > groups) grid = groups;
  const long long per_wg = ((groups + grid - 1) / grid) * per_round;
  grid = (total + per_wg - 1) / per_wg;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WG), lds, s, p, ps, M, packed, L, per_wg);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

// The framings istft_wave_fused_kernel serves: a compiled-in packed plan with one frame per wave,
// at most eight frames over a sample, everything within the LDS of a CU.
template <class PL>
static bool wave_fused_fits(int n_fft, int hop) {
  using namespace wavefft;
  if constexpr (!(PL::PACKED && PL::L == 64)) {
    return false;
  } else {
    return n_fft == 2 * PL::M && hop >= 1 && hop <= n_fft && (n_fft + hop - 1) / hop <= 8;
  }
}
static size_t wave_fused_lds(int M, int hop) {
  using namespace wavefft;
  const int N = 2 * M;
  return sizeof(float2) * ((size_t)WAVES * buf_len(M) + M + M / 2 + 1) +
         sizeof(float) * ((size_t)N + N + (size_t)(WAVES - 1) * hop + hop);
}

bool istft_wave_fused_supported(int n_fft, int hop) {
  using namespace wavefft;
  if (n_fft % 2 != 0) return false;
  bool ok = false;
#define MIA_FITS(PLAN) \
  if (wave_fused_fits<PLAN>(n_fft, hop)) ok = true;
  MIA_WAVE_INVERSE_ONLY_PLANS(MIA_FITS)
#undef MIA_FITS
  Passes ps{};
  int L = 64;
  if (!ok && forward_wave_plan(n_fft, &ps.n, ps.radix, &L)) {
#define MIA_FITS(PLAN) \
  if (PLAN::matches(ps, n_fft / 2, L, 1) && wave_fused_fits<PLAN>(n_fft, hop)) ok = true;
    MIA_WAVE_FIXED_PLANS(MIA_FITS)
#undef MIA_FITS
  }
  return ok && wave_fused_lds(n_fft / 2, hop) <= 160 * 1024;
}

template <class PL>
static int launch_wave_fused(const mia_plan* plan, const InvParams& p, hipStream_t s) {
  using namespace wavefft;
  if constexpr (!(PL::PACKED && PL::L == 64)) {
    set_error("internal: plan has no fused inverse");
    return MIA_ERR_UNSUPPORTED;
  } else {
    const size_t lds = wave_fused_lds(PL::M, p.hop);
    auto kern = istft_wave_fused_kernel<PL>;
    if (lds > 64 * 1024) {
      MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    int per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), WG, lds) !=
            hipSuccess || per_cu < 1) {
      per_cu = 1;
    }
    // runs of 64 frames (the warm-up frames in front of a run are transformed twice: <= 5 %),
    // shorter when that is what it takes to give every workgroup of the device a run
    const long long want = (long long)(plan->cus > 0 ? plan->cus : 256) * per_cu;
    long long fpr = 64;
    while (fpr > 16 && p.B * ((p.F + fpr - 1) / fpr) < want) fpr /= 2;
    const long long runs_per_row = (p.F + fpr - 1) / fpr;
    const long long total = p.B * runs_per_row;
    if (total <= 0) return MIA_OK;
    long long grid = want < total ? want : total;
    const long long per_wg = (total + grid - 1) / grid;
    grid = (total + per_wg - 1) / per_wg;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WG), lds, s, p, (int)fpr, runs_per_row, total, per_wg);
    MIA_CHECK_HIP(hipGetLastError());
    return MIA_OK;
  }
}

int launch_istft_wave_fused(const mia_plan* plan, const InvParams& p, hipStream_t s) {
  using namespace wavefft;
#define MIA_GO(PLAN) \
  if (wave_fused_fits<PLAN>(plan->n_fft, p.hop)) return launch_wave_fused<PLAN>(plan, p, s);
  MIA_WAVE_INVERSE_ONLY_PLANS(MIA_GO)
#undef MIA_GO
  Passes ps{};
  int L = 64;
  if (!forward_wave_plan(plan->n_fft, &ps.n, ps.radix, &L)) {
    set_error("n_fft=%d is not served by the per-wave transform", plan->n_fft);
    return MIA_ERR_UNSUPPORTED;
  }
#define MIA_GO(PLAN) \
  if (PLAN::matches(ps, plan->n_fft / 2, L, 1) && wave_fused_fits<PLAN>(plan->n_fft, p.hop)) \
    return launch_wave_fused<PLAN>(plan, p, s);
  MIA_WAVE_FIXED_PLANS(MIA_GO)
#undef MIA_GO
  set_error("n_fft=%d hop=%d has no fused per-wave inverse", plan->n_fft, p.hop);
  return MIA_ERR_UNSUPPORTED;
}

}  // namespace mia

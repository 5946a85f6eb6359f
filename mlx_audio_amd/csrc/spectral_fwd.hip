// Forward path: fused pad + frame + window + real FFT, with either a complex
// spectrum epilogue (stft) or a power -> mel band projection -> log epilogue
// (melspectrogram / Whisper log-mel / dB).
//
// Work decomposition.  A "group" of T threads transforms one frame; a
// workgroup holds G groups and walks ITERS frames per group, so one workgroup
// produces FT = G*ITERS consecutive frames of one signal.  T = M / (2*R_last)
// (M = n_fft/2) so that in the last Stockham pass every thread owns the two
// butterflies j and M/R_last - j, whose outputs are exactly the bin pairs
// (k, M-k) the real-FFT split step needs: the split runs in registers.
//
// In mel mode the power spectrum of the tile goes to an LDS tile P[FT][bins];
// after a barrier the band projection runs with consecutive lanes on
// consecutive frames, so the (n_mels, frames) output is written coalesced.
#include "fft_engine.h"
#include "mia_internal.h"

namespace mia {

// ---------------------------------------------------------------------------
// Configuration of one n_fft: group size and radices.
// ---------------------------------------------------------------------------
template <int N_, int T_, int R0_, int R1_, int R2_, int PADQ_, int R3_ = 1, int OCC_ = 2>
struct FftCfg {
  static constexpr int OCC = OCC_;                      // waves per SIMD the register budget is for
  static constexpr int N = N_, M = N_ / 2, T = T_;
  static constexpr int R0 = R0_, R1 = R1_, R2 = R2_, R3 = R3_;
  static constexpr int NPASS = 2 + (R2_ > 1) + (R3_ > 1);
  static constexpr int RL = (NPASS == 4) ? R3_ : ((NPASS == 3) ? R2_ : R1_);   // last radix
  static constexpr int NSL = M / RL;                    // stride of last pass
  static constexpr int PADQ = PADQ_;                    // one pad slot per PADQ; 0: XOR swizzle instead
  static constexpr bool XORSW = PADQ_ == 0;
  static constexpr int LBUF = XORSW ? M : M + M / (XORSW ? 1 : PADQ_) + 2;
  static constexpr bool MULTIWAVE = (T_ > 64);
  static constexpr int GPW = MULTIWAVE ? 0 : 64 / T_;   // groups per wave
  static_assert(R0_ * R1_ * R2_ * R3_ == M, "radices must multiply to n_fft/2");
  static_assert(R3_ == 1 || R2_ > 1, "a fourth pass needs a third");
  static_assert(2 * T_ * RL == M, "group size must be M/(2*R_last)");
  static_assert((M / R0_) % T_ == 0, "pass 0 must split evenly");
  static_assert(NPASS == 2 || (M / R1_) % T_ == 0, "pass 1 must split evenly");
  static_assert(NPASS < 4 || (M / R2_) % T_ == 0, "pass 2 must split evenly");
  static_assert(NSL % 2 == 0, "last stride must be even");

  template <int WG>
  static constexpr int groups() {
    return MULTIWAVE ? WG / T_ : (WG / 64) * GPW;
  }
  static __device__ __forceinline__ void map(int tid, int& g, int& t, bool& active) {
    if constexpr (MULTIWAVE) {
      g = tid / T_;
      t = tid % T_;
      active = true;
    } else {
      const int lane = tid & 63, wave = tid >> 6;
      g = wave * GPW + lane / T_;
      t = lane % T_;
      active = lane < GPW * T_;
    }
  }
  // LDS slot of logical element i.  Padding (one slot per PADQ elements) breaks the
  // power-of-two strides of the Stockham writes but leaves the contiguous reads of a 16- or
  // 32-lane group one slot too long for the banks (2-way on every such read).  The XOR form
  // (bits 3..6 of the index folded into bits 0..3) is a permutation inside every 128 elements:
  // contiguous runs stay conflict-free, the strided writes spread over all banks (a bank
  // simulation of the four passes gives 152 LDS cycles per frame and wave against 224 with
  // padding, 144 being conflict-free), and the buffer needs no extra slots.
  static __device__ __forceinline__ int pos(int i) {
    if constexpr (XORSW) {
      return i ^ ((i >> 3) & 15);
    } else {
      return i + i / (XORSW ? 1 : PADQ);
    }
  }
  template <int S>
  static __device__ __forceinline__ int pos_strided(int j, int r) {
    if constexpr (XORSW) {
      if constexpr (S % 128 == 0) {
        return pos(j) + r * S;  // a multiple of 128 leaves bits 0..6 alone
      } else {
        return pos(j + r * S);
      }
    } else if constexpr (S % (XORSW ? 1 : PADQ) == 0) {
      return pos(j) + r * (S + S / (XORSW ? 1 : PADQ));
    } else {
      return pos(j + r * S);
    }
  }
};

// One bin of the spectrum to memory.  Plans whose waves store whole 512-byte runs (a frame over
// one or more whole waves) mark the rows nontemporal: they are not read again by the launch, and
// the samples that neighbouring frames share stay in the L2 (HTDemucs round trip 11.34 -> 11.12 ms,
// 2048-point stft -5 %); the small plans' short runs lose by it (800 points +58 %).
typedef float fwd_f2 __attribute__((ext_vector_type(2)));
template <bool STREAM>
__device__ __forceinline__ void store_bin(float2* dst, float2 v) {
  if constexpr (STREAM) {
    __builtin_nontemporal_store(fwd_f2{v.x, v.y}, reinterpret_cast<fwd_f2*>(dst));
  } else {
    *dst = v;
  }
}

template <class C>
__device__ __forceinline__ void group_sync() {
  if constexpr (C::MULTIWAVE) {
    __syncthreads();
  } else {
    // The group lives inside one wave: program order plus in-order LDS
    // service make the exchange visible; the fence stops the compiler from
    // moving LDS accesses across the exchange point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

// A middle Stockham pass of radix R after passes whose radices multiply to NS: thread t takes
// butterflies t + bb * T; twiddles of its butterflies in twm (entry 0 of a row unused).
template <class C, int R, int NS, int NB>
__device__ __forceinline__ void middle_pass(float2* __restrict__ buf, int t, const float2 (&twm)[NB][R]) {
  constexpr int S = C::M / R;
  float2 v[NB][R];
#pragma unroll
  for (int bb = 0; bb < NB; ++bb) {
    const int j = t + bb * C::T;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float2 x = buf[C::template pos_strided<S>(j, r)];
      if (r > 0) x = cmul(x, twm[bb][r]);
      v[bb][r] = x;
    }
  }
  group_sync<C>();
#pragma unroll
  for (int bb = 0; bb < NB; ++bb) Dft<R>::run(v[bb]);
#pragma unroll
  for (int bb = 0; bb < NB; ++bb) {
    const int j = t + bb * C::T;
    const int base = (j / NS) * NS * R + (j % NS);
#pragma unroll
    for (int r = 0; r < R; ++r) buf[C::template pos_strided<NS>(base, r)] = v[bb][r];
  }
  group_sync<C>();
}
// The twiddles middle_pass needs: exp(-2 pi i (j mod NS) r / (NS R)) from the n_fft-entry table.
template <class C, int R, int NS, int NB>
__device__ __forceinline__ void middle_twiddles(const float2* __restrict__ tw, int tt, float2 (&twm)[NB][R]) {
  constexpr int TWS = 2 * (C::M / (NS * R));  // step in the n_fft-entry table
#pragma unroll
  for (int bb = 0; bb < NB; ++bb) {
    const int jm = (tt + bb * C::T) % NS;
#pragma unroll
    for (int r = 1; r < R; ++r) twm[bb][r] = tw[jm * r * TWS];
  }
#pragma unroll
  for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
    for (int r = 1; r < R; ++r) asm volatile("" : "+v"(twm[bb][r].x), "+v"(twm[bb][r].y));
  }
}

// Which registers pair up in the split step for thread 0 of a group (it owns
// the two self-paired butterflies 0 and NSL/2).  Slot s yields bins k and M-k.
struct T0Pair {
  bool u_b, v_b;   // operand lives in the B (second) butterfly
  int ui, vi;      // register index
  int k_half;      // k in units of NSL/2
};
template <int R>
constexpr T0Pair t0_pair(int s) {
  constexpr int NA = (R % 2 == 0) ? R / 2 : (R + 1) / 2;  // A-side slots
  if (s < NA) {
    return {false, false, s, (R - s) % R, 2 * s};
  }
  const int i = s - NA;
  return {true, true, i, R - 1 - i, 1 + 2 * i};
}

template <class C, int MODE, int WG, int ITERS>
__global__ __launch_bounds__(WG, C::OCC) void fwd_kernel(FwdParams p) {
  constexpr int M = C::M, T = C::T, N = C::N;
  constexpr int G = C::template groups<WG>();
  constexpr int FT = G * ITERS;
  constexpr int PS = (M + 1) | 1;  // odd row stride of the power tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2* bufs = reinterpret_cast<float2*>(smem_raw);
  float* P = reinterpret_cast<float*>(bufs + G * C::LBUF);

  const int tid = threadIdx.x;
  int g, t;
  bool active;
  C::map(tid, g, t, active);
  // a group of whole waves: its index is the same in every lane, which the compiler cannot see
  // (frame numbers, sample offsets and their range checks then run on the scalar unit)
  if constexpr (C::T >= 64) g = __builtin_amdgcn_readfirstlane(g);
  float2* buf = bufs + (active ? g : 0) * C::LBUF;
  const float2* __restrict__ tw = p.tw;
  const float2* __restrict__ win2 = reinterpret_cast<const float2*>(p.win);
  const bool t0 = (t == 0);

  // Persistent workgroup: a contiguous run of tiles.  Everything a thread needs from
  // the window and twiddle tables depends on its position t in the group only, so it
  // is fetched once, here, and stays in registers for the whole run (a wave-wide table
  // load costs the texture addresser as much as a 16-byte-per-lane data load, and the
  // passes below would issue one per element).
  constexpr int R0 = C::R0, S0 = M / R0, NB0 = S0 / T;
  constexpr int R1 = C::R1, S1 = M / R1, NB1 = (C::NPASS >= 3) ? S1 / T : 1;
  constexpr int R2 = C::R2, S2 = M / R2, NB2 = (C::NPASS == 4) ? S2 / T : 1;
  constexpr int RL = C::RL, NSL = C::NSL;
  float2 wn[NB0][R0];        // window taps of the thread's sample pairs
  float2 tw1[NB1][R1];       // middle-pass twiddles (3- and 4-pass plans)
  float2 tw2[NB2][R2];       // second middle pass (4-pass plans)
  float2 twa[RL], twb[RL];   // last-pass twiddles of butterflies t and NSL - t
  float2 tws[RL];            // split-step twiddles
  {
    const int tt = active ? t : 0;
#pragma unroll
    for (int bb = 0; bb < NB0; ++bb) {
#pragma unroll
      for (int r = 0; r < R0; ++r) wn[bb][r] = win2[tt + bb * T + r * S0];
    }
    if constexpr (C::NPASS == 3) {
      constexpr int NS = C::R0, TWS = 2 * (M / (NS * R1));  // step in the n_fft-entry table
#pragma unroll
      for (int bb = 0; bb < NB1; ++bb) {
        const int jm = (tt + bb * T) % NS;
#pragma unroll
        for (int r = 1; r < R1; ++r) tw1[bb][r] = tw[jm * r * TWS];
      }
    }
    if constexpr (C::NPASS == 4) {
      middle_twiddles<C, R1, C::R0, NB1>(tw, tt, tw1);
      middle_twiddles<C, R2, C::R0 * C::R1, NB2>(tw, tt, tw2);
    }
    const int jA = tt, jB = tt == 0 ? NSL / 2 : NSL - tt;
#pragma unroll
    for (int r = 1; r < RL; ++r) {
      twa[r] = tw[2 * jA * r];
      twb[r] = tw[2 * jB * r];
    }
    static_for<0, RL>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      constexpr T0Pair ps = t0_pair<RL>(r);
      tws[r] = tw[tt == 0 ? ps.k_half * (NSL / 2) : tt + NSL * r];
    });
    // opaque from here on: never re-fetched inside the loop
#pragma unroll
    for (int bb = 0; bb < NB0; ++bb) {
#pragma unroll
      for (int r = 0; r < R0; ++r) asm volatile("" : "+v"(wn[bb][r].x), "+v"(wn[bb][r].y));
    }
    if constexpr (C::NPASS == 3) {
#pragma unroll
      for (int bb = 0; bb < NB1; ++bb) {
#pragma unroll
        for (int r = 1; r < R1; ++r) asm volatile("" : "+v"(tw1[bb][r].x), "+v"(tw1[bb][r].y));
      }
    }
#pragma unroll
    for (int r = 1; r < RL; ++r) {
      asm volatile("" : "+v"(twa[r].x), "+v"(twa[r].y), "+v"(twb[r].x), "+v"(twb[r].y));
    }
#pragma unroll
    for (int r = 0; r < RL; ++r) asm volatile("" : "+v"(tws[r].x), "+v"(tws[r].y));
  }

  // Band tables of the mel stage: in LDS behind the power tile when the launcher found
  // room (one copy per persistent workgroup), else read from global memory.
  float* const lds_wts = P + FT * PS;
  // MODE 3 is MODE 1 (mel) with the tile written as (B, F, n_mels) rows: its own instantiation,
  // so that the choice costs the mel kernels no registers.
  constexpr bool MEL = MODE == 1 || MODE == 3;
  int* const lds_rows = reinterpret_cast<int*>(lds_wts + (MEL ? p.nnz : 0));
  if constexpr (MEL) {
    if (p.lds_tables) {
      for (int i = tid; i < p.nnz; i += WG) lds_wts[i] = p.wts[i];
      for (int i = tid; i < p.n_mels; i += WG) {
        lds_rows[i] = p.row_lo[i];
        lds_rows[p.n_mels + i] = p.row_len[i];
        lds_rows[2 * p.n_mels + i] = p.row_off[i];
      }
      __syncthreads();
    }
  }

  // Statistics mode: the bin frequencies sit in LDS behind the magnitude tile (every frame reads
  // all of them twice; from global memory that was a vector load per bin and lane).
  float* const lds_freq = P + FT * PS;
  if constexpr (MODE == 2) {
    for (int i = tid; i <= M; i += WG) lds_freq[i] = p.freq[i];
    __syncthreads();
  }

  const long long total = p.B * p.tiles;
  const long long first = (long long)blockIdx.x * p.tiles_per_wg;
  const long long last = first + p.tiles_per_wg < total ? first + p.tiles_per_wg : total;

  // Raw sample pairs of the frame this group transforms next, fetched one frame ahead
  // (behind the last pass's LDS reads of the current frame) when that frame lies wholly
  // inside its signal and is 8-byte aligned; other frames are loaded when they are used.
  float2 xin[NB0][R0];
  bool pre_ok = false;
#pragma unroll
  for (int bb = 0; bb < NB0; ++bb) {
#pragma unroll
    for (int r = 0; r < R0; ++r) xin[bb][r] = {0.0f, 0.0f};
  }
  auto prefetch = [&](long long idx_n, int it_n) {
    const long long bn = (long long)((unsigned)idx_n / (unsigned)p.tiles);  // the launcher keeps the tile count below 2^32
    const long long fn = (idx_n - bn * p.tiles) * FT + it_n * G + g;
    const float* __restrict__ ybn = p.y + bn * p.T;
    const long long s0 = fn * p.hop - p.pad;
    pre_ok = idx_n < last && fn < p.F && s0 >= 0 && s0 + N <= p.T &&
             ((reinterpret_cast<uintptr_t>(ybn + s0) & 7) == 0);
    if (pre_ok) {
      const float2* __restrict__ src = reinterpret_cast<const float2*>(ybn + s0) + t;
#pragma unroll
      for (int bb = 0; bb < NB0; ++bb) {
#pragma unroll
        for (int r = 0; r < R0; ++r) xin[bb][r] = src[bb * T + r * S0];
      }
    }
  };

  __shared__ float en_part[MODE == 7 ? FT * (T >= 64 ? T / 64 : 1) : 1];  // MODE 7: per frame and wave
  for (long long idx = first; idx < last; ++idx) {
  double mom_s = 0.0, mom_q = 0.0;  // MODE 5: this thread's share of the tile's moments
  const long long b = (long long)((unsigned)idx / (unsigned)p.tiles);
  const int tile = (int)(idx - b * p.tiles);
  const long long f0 = (long long)tile * FT;
  const float* __restrict__ yb = p.y + b * p.T;

  if (active) {
    for (int it = 0; it < ITERS; ++it) {
      const int fl = it * G + g;
      const long long f = f0 + fl;
      const bool valid = f < p.F;
      if (!C::MULTIWAVE && !valid) continue;
      const long long s0 = f * p.hop - p.pad;  // first sample of the frame
      const bool interior = valid && s0 >= 0 && s0 + N <= p.T;
      const bool aligned8 = ((reinterpret_cast<uintptr_t>(yb + s0) & 7) == 0);

      // ---- pass 0: windowed samples -> radix R0 -> LDS ----------------------
      {
        constexpr int R = R0, S = S0, NB = NB0;
        float2 v[NB][R];
        if constexpr (T >= 64) {
          // a group is one or more whole waves: one (wave-uniform) decision per frame on
          // where the samples come from.  Measured: 2048-point plan 5.3 -> 3.5 ms on the
          // MFCC workload, while the 1024-point plan (two groups per wave) prefers the
          // per-pair form below (1.83 vs 2.03 ms).
          if (!pre_ok) {
            if (interior && aligned8) {
              const float2* __restrict__ src = reinterpret_cast<const float2*>(yb + s0) + t;
#pragma unroll
              for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
                for (int r = 0; r < R; ++r) xin[bb][r] = src[bb * T + r * S];
              }
            } else {
#pragma unroll
              for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                  const int n = t + bb * T + r * S;
                  float2 x = {0.0f, 0.0f};
                  if (interior) {
                    x.x = yb[s0 + 2 * n];
                    x.y = yb[s0 + 2 * n + 1];
                  } else if (valid) {
                    x.x = fetch_sample(yb, p.T, s0 + 2 * n, p.pad_mode);
                    x.y = fetch_sample(yb, p.T, s0 + 2 * n + 1, p.pad_mode);
                  }
                  xin[bb][r] = x;
                }
              }
            }
          }
#pragma unroll
          for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
              v[bb][r] = {xin[bb][r].x * wn[bb][r].x, xin[bb][r].y * wn[bb][r].y};
            }
          }
        } else if (__all(pre_ok)) {
          // the usual case, decided once for the wave's groups together
#pragma unroll
          for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
              v[bb][r] = {xin[bb][r].x * wn[bb][r].x, xin[bb][r].y * wn[bb][r].y};
            }
          }
        } else {
#pragma unroll
          for (int bb = 0; bb < NB; ++bb) {
            const int j = t + bb * T;
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const int n = j + r * S;
              float2 x;
              if (pre_ok) {
                x = xin[bb][r];
              } else if (interior) {
                if (aligned8) {
                  x = *reinterpret_cast<const float2*>(yb + s0 + 2 * n);
                } else {
                  x.x = yb[s0 + 2 * n];
                  x.y = yb[s0 + 2 * n + 1];
                }
              } else if (valid) {
                x.x = fetch_sample(yb, p.T, s0 + 2 * n, p.pad_mode);
                x.y = fetch_sample(yb, p.T, s0 + 2 * n + 1, p.pad_mode);
              } else {
                x = {0.0f, 0.0f};
              }
              v[bb][r] = {x.x * wn[bb][r].x, x.y * wn[bb][r].y};
            }
          }
        }
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) Dft<R>::run(v[bb]);
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
          const int j = t + bb * T;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int slot = (C::PADQ == R) ? j * (R + 1) + r : C::pos(j * R + r);
            buf[slot] = v[bb][r];
          }
        }
      }
      group_sync<C>();

      // ---- middle passes (3- and 4-pass plans) -------------------------------
      if constexpr (C::NPASS == 3) {
        constexpr int R = R1, S = S1, NB = NB1, NS = C::R0;
        float2 v[NB][R];
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
          const int j = t + bb * T;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            float2 x = buf[C::template pos_strided<S>(j, r)];
            if (r > 0) x = cmul(x, tw1[bb][r]);
            v[bb][r] = x;
          }
        }
        group_sync<C>();
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) Dft<R>::run(v[bb]);
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
          const int j = t + bb * T;
          const int base = (j / NS) * NS * R + (j % NS);
#pragma unroll
          for (int r = 0; r < R; ++r) buf[C::template pos_strided<NS>(base, r)] = v[bb][r];
        }
        group_sync<C>();
      }
      if constexpr (C::NPASS == 4) {
        middle_pass<C, R1, C::R0, NB1>(buf, t, tw1);
        middle_pass<C, R2, C::R0 * C::R1, NB2>(buf, t, tw2);
      }

      // ---- last pass: butterflies t and NSL - t ------------------------------
      float2 A[RL], Bv[RL];
      {
        const int jA = t;
        const int jB = t0 ? NSL / 2 : NSL - t;
#pragma unroll
        for (int r = 0; r < RL; ++r) {
          float2 xa = buf[C::template pos_strided<NSL>(jA, r)];
          float2 xb = buf[C::template pos_strided<NSL>(jB, r)];
          if (r > 0) {
            xa = cmul(xa, twa[r]);
            xb = cmul(xb, twb[r]);
          }
          A[r] = xa;
          Bv[r] = xb;
        }
        group_sync<C>();  // next iteration overwrites buf
        if (it + 1 < ITERS) {
          prefetch(idx, it + 1);
        } else {
          prefetch(idx + 1, 0);
        }
        Dft<RL>::run(A);
        Dft<RL>::run(Bv);
      }

      // ---- split step in registers: bins k and M-k ---------------------------
      // (frames past the end exist only in plans whose groups span waves; they took part
      // in the passes for the barriers' sake and produce nothing)
      if (!valid) continue;
      float2* __restrict__ crow = (MODE == 0 || MODE == 5 || MODE == 7) ? p.cout + (b * p.F + f) * (long long)(M + 1) : nullptr;
      float en = 0.0f;  // MODE 7: this thread's share of the frame's sum of |X|^2
      const bool mom_frame = MODE == 5 && f >= p.mom_f_lo && f < p.mom_f_hi;
      float* __restrict__ prow = P + fl * PS;
      const float half_scale = 0.5f * p.scale;
      auto emit = [&](int k, float2 X) {
        if constexpr (MODE == 0) {
#ifdef MIA_DEBUG_HOOKS  // ablations (garbage results): bit 4 no stores, bit 32 every row to the first frame's row (L2-resident)
          if (p.dbg & 4) {
            asm volatile("" ::"v"(X.x), "v"(X.y));
            return;
          }
          if (p.dbg & 64) {  // rows at a 128-byte-aligned pitch (2048 bins, the last one dropped)
            if (k < M) p.cout[(b * p.F + f) * (long long)M + k] = {X.x * half_scale, X.y * half_scale};
            return;
          }
          if (p.dbg & 32) {
            p.cout[(blockIdx.x & 1023) * (long long)(M + 1) + k] = {X.x * half_scale, X.y * half_scale};
            return;
          }
#endif
          store_bin<(C::N >= 2048)>(crow + k, {X.x * half_scale, X.y * half_scale});
        } else if constexpr (MODE == 7) {
          const float2 z = {X.x * half_scale, X.y * half_scale};
          store_bin<(C::N >= 2048)>(crow + k, z);
          en += __builtin_fmaf(z.x, z.x, z.y * z.y);
        } else if constexpr (MODE == 5) {
          const float2 z = {X.x * half_scale, X.y * half_scale};
          store_bin<(C::N >= 2048)>(crow + k, z);
          if (mom_frame && k < p.mom_bins) {
            mom_s += (double)z.x + (double)z.y;
            mom_q += (double)z.x * (double)z.x + (double)z.y * (double)z.y;
          }
        } else {
          prow[k] = 0.25f * (X.x * X.x + X.y * X.y);  // |X|^2; other exponents: the pass below
        }
      };
      static_for<0, RL>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr T0Pair ps = t0_pair<RL>(r);
        const float2 U0 = ps.u_b ? Bv[ps.ui] : A[ps.ui];
        const float2 V0 = ps.v_b ? Bv[ps.vi] : A[ps.vi];
        const float2 U = t0 ? U0 : A[r];
        const float2 V = t0 ? V0 : Bv[RL - 1 - r];
        const int k = t0 ? ps.k_half * (NSL / 2) : t + NSL * r;
        const float2 w = tws[r];
        // E = U + conj(V);  O = -i (U - conj(V));  X[k] = (E + w O)/2,
        // X[M-k] = conj(E - w O)/2.
        const float2 E = {U.x + V.x, U.y - V.y};
        const float2 O = {U.y + V.y, V.x - U.x};
        const float2 WO = cmul(O, w);
        emit(k, {E.x + WO.x, E.y + WO.y});
        emit(M - k, {E.x - WO.x, WO.y - E.y});
      });
      if (t0) {  // the self-paired bin M/2: X = conj(Z[M/2])
        const float2 Z = (RL % 2 == 0) ? A[RL / 2] : Bv[(RL - 1) / 2];
        emit(M / 2, {2.0f * Z.x, -2.0f * Z.y});
      }
      if constexpr (MODE == 7) {
        // the frame's energy: a fixed shuffle tree over each wave, the waves' sums added in order
        // once the tile is done (a frame is one or more whole waves in the plans that take this mode)
        static_assert(MODE != 7 || T % 64 == 0, "a frame must be whole waves");
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) en += __shfl_xor(en, off);
        if ((tid & 63) == 0) en_part[fl * (T / 64) + (t >> 6)] = en;
      }
    }
  }
  if constexpr (MODE == 7) {
    __syncthreads();
    if (tid < FT && f0 + tid < p.F) {
      float sum = en_part[tid * (T / 64)];
#pragma unroll
      for (int w = 1; w < T / 64; ++w) sum += en_part[tid * (T / 64) + w];
      p.energy[b * p.F + f0 + tid] = sum / (float)(M + 1);
    }
    __syncthreads();  // the next tile reuses the slots
  }

  if constexpr (MODE == 5) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mom_s += __shfl_xor(mom_s, off);
      mom_q += __shfl_xor(mom_q, off);
    }
    __shared__ double mom_red[2][WG / 64];
    if ((tid & 63) == 0) {
      mom_red[0][tid >> 6] = mom_s;
      mom_red[1][tid >> 6] = mom_q;
    }
    __syncthreads();
    if (tid == 0) {
      double s = mom_red[0][0], q = mom_red[1][0];
#pragma unroll
      for (int w = 1; w < WG / 64; ++w) {
        s += mom_red[0][w];
        q += mom_red[1][w];
      }
      double* __restrict__ dst = p.mom_sums + 2 * (b / p.mom_group);
      atomicAdd(dst, s);
      atomicAdd(dst + 1, q);
    }
    __syncthreads();  // the next tile reuses the slots
  }
  if constexpr (MODE != 0 && MODE != 7) {
    // Magnitudes or |X|^p instead of powers: one pass over the tile, decided once per tile.  In
    // the split step above the choice was two scalar branches per bin pair, and seventeen
    // inlined copies of powf stood between the hot path's instructions.
    if (p.power_mode != 0) {
      __syncthreads();
      for (int i = tid; i < FT * PS; i += WG) {
        const int fl = i / PS, k = i - fl * PS;
        if (k <= M && f0 + fl < p.F) {
          const float mag = sqrtf(P[i]);
          P[i] = p.power_mode == 1 ? mag : powf(mag, p.power);
        }
      }
    }
  }
  if constexpr (MODE == 2) {
    __syncthreads();
    const long long frem2 = p.F - f0;
    const int fcount = frem2 < FT ? (int)frem2 : FT;
    {
      // Spectral statistics of the tile's frames straight from the magnitude rows in LDS:
      // a wave per frame, every lane a contiguous run of bins (odd run lengths and the odd row
      // pitch keep the reads conflict-free), wave reductions for the sums and a wave scan for
      // the roll-off point.  out: (B, 4, F) = centroid, bandwidth, roll-off, flatness.
      constexpr int BINS = M + 1;
      constexpr int RUN = ((BINS + 63) / 64) | 1;
      const int lane = tid & 63;
      const int k0 = lane * RUN < BINS ? lane * RUN : BINS;
      const int k1 = k0 + RUN < BINS ? k0 + RUN : BINS;
      for (int fl = tid >> 6; fl < fcount; fl += WG / 64) {
        const float* __restrict__ pr = P + fl * PS;
        float ts = 0.0f, ws = 0.0f, ls = 0.0f, as = 0.0f;
#pragma unroll 6
        for (int k = k0; k < k1; ++k) {
          const float v = pr[k];
          ts += v;
          ws += lds_freq[k] * v;
          float u = p.flat_power == 2.0f ? v * v : (p.flat_power == 1.0f ? v : powf(v, p.flat_power));
          u = u < p.flat_amin ? p.flat_amin : u;
          ls += logf(u);
          as += u;
        }
        float incl = ts;  // inclusive scan of the lanes' sums
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const float t = __shfl_up(incl, off);
          if (lane >= off) incl += t;
        }
        const float excl = incl - ts;
        const float total = __shfl(incl, 63);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          ws += __shfl_xor(ws, off);
          ls += __shfl_xor(ls, off);
          as += __shfl_xor(as, off);
        }
        const float centroid = ws / (total + 1e-8f);
        float bw = 0.0f;
#pragma unroll 6
        for (int k = k0; k < k1; ++k) {
          const float d = fabsf(lds_freq[k] - centroid);
          const float dp = p.stat_p == 2.0f ? d * d : (p.stat_p == 1.0f ? d : powf(d, p.stat_p));
          bw += pr[k] * dp;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bw += __shfl_xor(bw, off);
        if (p.stat_norm) bw = bw / (total + 1e-8f);
        bw = p.stat_p == 2.0f ? sqrtf(bw) : (p.stat_p == 1.0f ? bw : powf(bw, 1.0f / p.stat_p));
        // roll-off: the first lane whose running sum reaches the threshold finds the bin
        const float thr = p.stat_roll * total;
        const unsigned long long reached = __ballot(incl >= thr && k1 > k0);
        int bin = BINS - 1;
        if (reached != 0ULL) {
          const int first = __ffsll((long long)reached) - 1;
          int mine = k1 - 1;
          if (lane == first) {
            float cs = excl;
            for (int k = k0; k < k1; ++k) {
              cs += pr[k];
              if (cs >= thr) {
                mine = k;
                break;
              }
            }
          }
          bin = __shfl(mine, first);
        }
        if (lane == 0) {
          float* __restrict__ o = p.mout + (b * 4) * p.F + f0 + fl;
          o[0] = centroid;
          o[p.F] = bw;
          o[2 * p.F] = lds_freq[bin];
          o[3 * p.F] = expf(ls / (float)BINS) / (as / (float)BINS + 1e-8f);
        }
      }
    }
    __syncthreads();  // the next tile overwrites the magnitude rows
  }
  if constexpr (MEL) {
    __syncthreads();
    const long long frem = p.F - f0;
    const int fcount = frem < FT ? (int)frem : FT;
    int kmax = kKeyMinusInf;  // integer keys: a NaN value wins the maximum, as in the reference
    float vmin = INFINITY;
    // one instantiation per table location, so that each knows its address space
    // frame_major: the tile's values go to (B, F, n_mels) rows through the (now idle) FFT
    // buffers, so that the tile leaves as one contiguous run of whole lines
    float* const stage = reinterpret_cast<float*>(bufs);
    auto project = [&](const int* row_lo, const int* row_len, const int* row_off, const float* wts) {
      for (int idx2 = tid; idx2 < p.n_mels * FT; idx2 += WG) {
        const int m = idx2 / FT;
        const int fl = idx2 - m * FT;
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
      float* __restrict__ dst = p.mout + (b * p.F + f0) * p.n_mels;  // 16-byte aligned: n_mels % 4 == 0
      const int n4 = fcount * p.n_mels / 4;
      for (int i = tid; i < n4; i += WG) {
        reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(stage)[i];
      }
    }
    if (p.epi != MIA_EPI_NONE && p.clip_max_key != nullptr) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        kmax = max(kmax, __shfl_xor(kmax, off));
        vmin = fminf(vmin, __shfl_xor(vmin, off));
      }
      __shared__ int red_max[WG / 64];
      __shared__ float red_min[WG / 64];
      if ((tid & 63) == 0) {
        red_max[tid >> 6] = kmax;
        red_min[tid >> 6] = vmin;
      }
      __syncthreads();
      if (tid == 0) {
#pragma unroll
        for (int w = 1; w < WG / 64; ++w) {
          kmax = max(kmax, red_max[w]);
          vmin = fminf(vmin, red_min[w]);
        }
        atomicMax(p.clip_max_key + b, kmax);
        atomicMin(p.clip_min_key + b, float_to_key(vmin));
      }
    }
    __syncthreads();  // the next tile overwrites the power tile and the reduction slots
  }
  }  // tiles of this workgroup
}

// ---------------------------------------------------------------------------
// Size table and dispatch
// ---------------------------------------------------------------------------
//                    N     T   R0  R1  R2  PADQ
#ifndef MIA_CFG256
#define MIA_CFG256 FftCfg<256, 8, 16, 8, 1, 16>
#endif
#ifndef MIA_IT256
#define MIA_IT256 2
#endif
using Cfg256 = MIA_CFG256;
// 400 and 800 points (n_fft / 2 = 2^3 5^2 and 2^4 5^2 complex points): twenty threads per frame
// and ten points per thread and pass.  The two-pass plans these sizes had before (radix 20 x 10
// and 20 x 20 over ten threads) kept 20 / 40 points and as many window taps, prefetched samples
// and twiddles per thread: 256 registers plus 140 - 1 860 bytes of scratch per lane, and the
// 800-point plan ran ten times below the power-of-two sizes' rate per n log n (256 x 30 s at
// hop n_fft / 4, mel / stft: 5.53 / 4.59 ms -> 1.16 / 0.88 ms; 400 points 1.37 / 0.99 -> 1.00 /
// 0.85 ms; tools/ab_sizes.sh).  800 points run under the register budget of one wave per SIMD
// (the overflow goes to accumulation registers, not to memory).
#ifndef MIA_CFG400
#define MIA_CFG400 FftCfg<400, 20, 10, 2, 2, 10, 5>
#endif
#ifndef MIA_CFG800
#define MIA_CFG800 FftCfg<800, 20, 10, 4, 10, 10, 1, 1>
#endif
#ifndef MIA_WG400
#define MIA_WG400 256
#endif
#ifndef MIA_IT400
#define MIA_IT400 4
#endif
#ifndef MIA_WG800
#define MIA_WG800 128
#endif
#ifndef MIA_IT800
#define MIA_IT800 4
#endif
using Cfg400 = MIA_CFG400;
using Cfg512 = FftCfg<512, 32, 8, 8, 4, 8>;
using Cfg800 = MIA_CFG800;
// 1024, 2048 and 4096 points: eight points per thread and pass (four passes of radix 8 / 4, a
// frame is one / two / four whole waves): half the per-thread state of a three-pass plan.
// What that state buys is occupancy: with 256-thread workgroups, two frames per group and tile,
// and the register budget of THREE waves per SIMD (168 registers, no spills) instead of two,
// CLAP went 1.54 -> 1.20 ms and MFCC 12.7 -> 11.7 ms on one box (tools/ab_w.sh); four waves per
// SIMD spill and lose (2.15 / 22.5 ms), as do the one-wave three-pass plan for 2048 points under
// the same budget (39 ms) and 128-thread workgroups (2.0 ms).  The smaller tile (8 / 4 frames)
// costs nothing where the output leaves frame-major (MFCC); CLAP's rows leave in 32-byte pieces.
// The knobs stay macros for such A/B builds (tools/build_variant.sh).
#ifndef MIA_WG1024
#define MIA_WG1024 256
#endif
#ifndef MIA_IT1024
#define MIA_IT1024 2
#endif
#ifndef MIA_OCC1024
#define MIA_OCC1024 3
#endif
#ifndef MIA_WG2048
#define MIA_WG2048 256
#endif
#ifndef MIA_IT2048
#define MIA_IT2048 2
#endif
#ifndef MIA_OCC2048
#define MIA_OCC2048 3
#endif
#ifndef MIA_OCC4096
#define MIA_OCC4096 2
#endif
// LDS layout of the FFT buffers (A/B on one box, tools/ab_w.sh): for 1024 points the compiler
// pairs the LDS reads (ds_read2_b64: 16-lane groups over 32 banks), which a pad slot per 16
// elements keeps conflict-free where one per 8 makes every contiguous read 2-way (CLAP 1.215 ->
// 1.115 ms); 2048 and 4096 points read with single ds_read_b64 and prefer one per 8 (MFCC 11.75
// against 12.0 ms).  The XOR swizzle (PADQ 0) removes nearly all conflicts in the bank
// simulation but its address arithmetic costs more than they did (1.43 / 14.4 ms).
#ifndef MIA_PADQ1024
#define MIA_PADQ1024 16
#endif
#ifndef MIA_PADQ
#define MIA_PADQ 8
#endif
using Cfg1024 = FftCfg<1024, 64, 8, 4, 4, MIA_PADQ1024, 4, MIA_OCC1024>;
#ifdef MIA_FWD2048_ONE_WAVE
using Cfg2048 = FftCfg<2048, 64, 16, 8, 8, 16, 1, MIA_OCC2048>;
#else
using Cfg2048 = FftCfg<2048, 128, 8, 8, 4, MIA_PADQ, 4, MIA_OCC2048>;
#endif
using Cfg4096 = FftCfg<4096, 256, 8, 8, 8, MIA_PADQ, 4, MIA_OCC4096>;
#define MIA_FWD_BIG(X) \
  X(1024, Cfg1024, MIA_WG1024, MIA_IT1024) \
  X(2048, Cfg2048, MIA_WG2048, MIA_IT2048) \
  X(4096, Cfg4096, 256, 4)

template <class C, int MODE, int WG, int ITERS>
static int launch_cfg(const FwdParams& p0, hipStream_t s) {
  constexpr int G = C::template groups<WG>();
  constexpr int FT = G * ITERS;
  constexpr int PS = (C::M + 1) | 1;
  FwdParams p = p0;
  p.tiles = (int)((p.F + FT - 1) / FT);
  constexpr bool TILE = MODE != 0 && MODE != 5 && MODE != 7;  // modes that keep a power / magnitude tile
  size_t lds = sizeof(float2) * G * C::LBUF + (TILE ? sizeof(float) * FT * PS : 0);
  if (MODE == 2) lds += sizeof(float) * (C::M + 1);  // the bin frequencies
  if (MODE == 1 || MODE == 3) {
    // band tables ride along in LDS when two workgroups still fit a CU
    const size_t tables = sizeof(float) * (size_t)p.nnz + 3 * sizeof(int) * (size_t)p.n_mels;
    if (lds + tables <= 80 * 1024) {
      lds += tables;
      p.lds_tables = 1;
    }
  }
  static_assert(sizeof(float2) * G * C::LBUF + (TILE ? sizeof(float) * FT * PS : 0) <=
                    160 * 1024,
                "LDS budget");
  const long long total = p.B * p.tiles;
  if (total <= 0) return MIA_OK;
  if (total >= 0xffffffffLL) {  // the kernel divides tile indices in 32 bits
    set_error("%lld tiles do not fit one launch (split the batch)", total);
    return MIA_ERR_UNSUPPORTED;
  }
  if (MODE == 3 &&
      (p.n_mels % 4 != 0 || sizeof(float) * (size_t)FT * p.n_mels > sizeof(float2) * G * C::LBUF)) {
    set_error("frame-major mel rows do not fit this plan's FFT buffers");
    return MIA_ERR_UNSUPPORTED;
  }
  auto kern = fwd_kernel<C, MODE, WG, ITERS>;
  if (lds > 64 * 1024) {
    MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  // Persistent grid: as many workgroups as the device holds at once, each walking a
  // contiguous run of tiles.
  static int resident[16] = {0};  // workgroups per device for this instantiation
  int dev = 0;
  MIA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) dev = 0;
  if (resident[dev] == 0) {
    int cus = 256, per_cu = 1;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), WG,
                                                     lds) != hipSuccess || per_cu < 1) {
      per_cu = 1;
    }
    resident[dev] = cus * per_cu;
  }
  long long grid = resident[dev];
  if (grid > total) grid = total;
  const long long per_wg = (total + grid - 1) / grid;
  if (per_wg > 0x7fffffffLL) {
    set_error("batch too large for one launch: %lld tiles (split the batch)", total);
    return MIA_ERR_UNSUPPORTED;
  }
  p.tiles_per_wg = (int)per_wg;
  grid = (total + per_wg - 1) / per_wg;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WG), lds, s, p);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

// Per size: workgroup threads and frames per group (ITERS), chosen so that the
// group buffers plus the power tile stay under 80 KiB (two workgroups per CU).
#define MIA_FWD_SIZES(X) \
  X(256, Cfg256, 256, MIA_IT256)  \
  X(400, Cfg400, MIA_WG400, MIA_IT400)  \
  X(512, Cfg512, 256, 4)  \
  X(800, Cfg800, MIA_WG800, MIA_IT800)  \
  MIA_FWD_BIG(X)

template <int MODE>
static int dispatch_mode(int n_fft, const FwdParams& p, hipStream_t s) {
  switch (n_fft) {
#define X(NFFT, CFG, WG, ITERS) \
  case NFFT: return launch_cfg<CFG, MODE, WG, ITERS>(p, s);
    MIA_FWD_SIZES(X)
#undef X
    default:
      set_error("n_fft=%d has no compiled FFT plan (supported: 256, 400, 512, 800, 1024, "
                "2048, 4096)", n_fft);
      return MIA_ERR_UNSUPPORTED;
  }
}

bool fft_size_supported(int n_fft) {
  switch (n_fft) {
#define X(NFFT, CFG, WG, ITERS) case NFFT:
    MIA_FWD_SIZES(X)
#undef X
    return true;
    default: return false;
  }
}

// Whether launch_forward(mode 1) can write (B, F, n_mels) rows for this plan.
bool forward_frame_major_ok(int n_fft, int n_mels) {
  if (n_mels % 4 != 0) return false;
  switch (n_fft) {
#define X(NFFT, CFG, WG, ITERS)                                                              \
  case NFFT:                                                                                 \
    return sizeof(float) * (size_t)(CFG::groups<WG>() * ITERS) * n_mels <=                   \
           sizeof(float2) * (size_t)CFG::groups<WG>() * CFG::LBUF;
    MIA_FWD_SIZES(X)
#undef X
    default: return false;
  }
}

int forward_tile_frames(int n_fft, int /*mode*/) {
  switch (n_fft) {
#define X(NFFT, CFG, WG, ITERS) \
  case NFFT: return CFG::groups<WG>() * ITERS;
    MIA_FWD_SIZES(X)
#undef X
    default: return 0;
  }
}

int launch_forward(const mia_plan* plan, const FwdParams& p, int mode, hipStream_t s) {
  // frame-major mel rows at n_fft 1024 by the wave-private kernel (spectral_lane.hip): an A/B
  // option - on the CLAP workload it equals the four-pass kernel's time, see the profiles
  if (mode == 1 && p.frame_major && plan->opt_lane &&
      forward_lane_ok(plan->n_fft, plan->hop, p.n_mels, p.nnz)) {
    return launch_forward_lane(p, plan->cus, s);
  }
  if (mode == 0) return dispatch_mode<0>(plan->n_fft, p, s);
  if (mode == 5) return dispatch_mode<5>(plan->n_fft, p, s);
  if (mode == 7) {  // spectrum + frame energies: the plans whose frames are whole waves
    switch (plan->n_fft) {
#define X(NFFT, CFG, WG, ITERS) \
  case NFFT: return launch_cfg<CFG, 7, WG, ITERS>(p, s);
      MIA_FWD_BIG(X)
#undef X
      default:
        set_error("n_fft=%d has no spectrum + energy kernel (1024, 2048, 4096 have)", plan->n_fft);
        return MIA_ERR_UNSUPPORTED;
    }
  }
  if (mode == 1) return p.frame_major ? dispatch_mode<3>(plan->n_fft, p, s) : dispatch_mode<1>(plan->n_fft, p, s);
  return dispatch_mode<2>(plan->n_fft, p, s);
}

}  // namespace mia

// Inverse path: complex-to-real FFT of every STFT frame (the irfft step of
// istft), writing time-domain frames (B, F, n_fft).
//
// Mirror image of the forward kernel: the split step comes first.  In the
// first Stockham pass every thread owns the two butterflies j and S0 - j
// (S0 = M / R0), whose inputs are exactly the bin pairs (k, M-k), so
// Z[k] = E + iO is formed in registers from X[k] and X[M-k].  The inverse
// transform is computed as swap(FFT(swap(Z))) so the forward butterflies and
// the forward twiddle table are reused unchanged.
#include "fft_engine.h"
#include "mia_internal.h"

namespace mia {

template <int N_, int T_, int R0_, int R1_, int R2_, int PADQ_>
struct InvCfg {
  static constexpr int N = N_, M = N_ / 2, T = T_;
  static constexpr int R0 = R0_, R1 = R1_, R2 = R2_;
  static constexpr int NPASS = (R2_ > 1) ? 3 : 2;
  static constexpr int RL = (NPASS == 3) ? R2_ : R1_;
  static constexpr int S0 = M / R0_;
  static constexpr int PADQ = PADQ_;
  static constexpr int LBUF = M + M / PADQ_ + 2;
  static constexpr bool MULTIWAVE = (T_ > 64);
  static constexpr int GPW = MULTIWAVE ? 0 : 64 / T_;
  static_assert(R0_ * R1_ * R2_ == M, "radices must multiply to n_fft/2");
  static_assert(2 * T_ == S0, "group size must be M/(2*R0)");
  static_assert(S0 % 2 == 0, "first stride must be even");
  static_assert((M / RL) % T_ == 0, "last pass must split evenly");
  static_assert(NPASS == 2 || (M / R1_) % T_ == 0, "pass 1 must split evenly");

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
  static __device__ __forceinline__ int pos(int i) { return i + i / PADQ; }
  template <int S>
  static __device__ __forceinline__ int pos_strided(int j, int r) {
    if constexpr (S % PADQ == 0) {
      return pos(j) + r * (S + S / PADQ);
    } else {
      return pos(j + r * S);
    }
  }
};

template <class C>
__device__ __forceinline__ void inv_group_sync() {
  if constexpr (C::MULTIWAVE) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

// Thread 0 of a group owns the self-paired butterflies 0 and S0/2.  For load
// slot s (bins k and M-k) this says which registers receive Z[k] and Z[M-k].
struct T0Slot {
  bool b_side;   // registers of the B (second) butterfly
  int zk, zmk;   // register receiving Z[k] / Z[M-k]
  int k_half;    // k in units of S0/2
};
template <int R>
constexpr T0Slot t0_slot(int s) {
  constexpr int NA = (R % 2 == 0) ? R / 2 : (R + 1) / 2;
  if (s < NA) return {false, s, (R - s) % R, 2 * s};
  const int i = s - NA;
  return {true, i, R - 1 - i, 1 + 2 * i};
}

template <class C, int WG, int ITERS>
__global__ __launch_bounds__(WG) void inv_kernel(InvParams p) {
  constexpr int M = C::M, T = C::T, N = C::N;
  constexpr int G = C::template groups<WG>();
  constexpr int FT = G * ITERS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2* bufs = reinterpret_cast<float2*>(smem_raw);

  const int tid = threadIdx.x;
  int g, t;
  bool active;
  C::map(tid, g, t, active);
  float2* buf = bufs + (active ? g : 0) * C::LBUF;
  const long long b = blockIdx.x / p.tiles;
  const int tile = blockIdx.x % p.tiles;
  const long long f0 = (long long)tile * FT;
  const float2* __restrict__ tw = p.tw;
  const bool t0 = (t == 0);
  const float inv_m = 1.0f / float(M);

  if (!active) return;
  for (int it = 0; it < ITERS; ++it) {
    const long long f = f0 + it * G + g;
    const bool valid = f < p.F;
    if (!C::MULTIWAVE && !valid) continue;
    const float2* __restrict__ xs = p.spec + b * p.sb + (valid ? f : 0) * p.sf;

    // ---- pass 0: bins -> Z (split step) -> radix R0 -> LDS -------------------
    {
      constexpr int R = C::R0, S0 = C::S0;
      float2 zk[R], zmk[R];
#pragma unroll
      for (int s = 0; s < R; ++s) {
        // generic thread: k = t + S0*s; thread 0: table below
        const T0Slot ts = t0_slot<R>(s);
        const int k = t0 ? ts.k_half * (S0 / 2) : t + S0 * s;
        float2 xk = {0.0f, 0.0f}, xm = {0.0f, 0.0f};
        if (valid) {
          xk = xs[(long long)k * p.sk];
          xm = xs[(long long)(M - k) * p.sk];
        }
        if (k == 0) {  // the real transform ignores Im X[0] and Im X[M]
          xk.y = 0.0f;
          xm.y = 0.0f;
        }
        const float2 w = tw[k];
        // E = (X[k] + conj X[M-k])/2, O = conj(w) (X[k] - conj X[M-k])/2,
        // Z[k] = E + iO, Z[M-k] = conj(E) + i conj(O).  The 1/2 is folded
        // into the final scale.
        const float2 E = {xk.x + xm.x, xk.y - xm.y};
        const float2 D = {xk.x - xm.x, xk.y + xm.y};
        const float2 O = cmul_conj(D, w);
        zk[s] = {E.x - O.y, E.y + O.x};
        zmk[s] = {E.x + O.y, O.x - E.y};
      }
      float2 A[R], Bv[R];
      // generic threads: A[s] = Z[t + S0 s], B[R-1-s] = Z[M - k_s]
#pragma unroll
      for (int s = 0; s < R; ++s) {
        A[s] = zk[s];
        Bv[R - 1 - s] = zmk[s];
      }
      if (t0) {
        float2 A0[R], B0[R];
        static_for<0, R>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          constexpr T0Slot ts = t0_slot<R>(s);
          if constexpr (ts.b_side) {
            B0[ts.zk] = zk[s];
            B0[ts.zmk] = zmk[s];
          } else {
            A0[ts.zmk] = zmk[s];
            A0[ts.zk] = zk[s];
          }
        });
        // the self-paired bin M/2: Z = conj(X[M/2]) (times 2 for the folded 1/2)
        float2 xh = {0.0f, 0.0f};
        if (valid) xh = xs[(long long)(M / 2) * p.sk];
        const float2 zh = {2.0f * xh.x, -2.0f * xh.y};
        if constexpr (R % 2 == 0) {
          A0[R / 2] = zh;
        } else {
          B0[(R - 1) / 2] = zh;
        }
#pragma unroll
        for (int s = 0; s < R; ++s) {
          A[s] = A0[s];
          Bv[s] = B0[s];
        }
      }
      // inverse = swap(FFT(swap(.)))
#pragma unroll
      for (int s = 0; s < R; ++s) {
        A[s] = {A[s].y, A[s].x};
        Bv[s] = {Bv[s].y, Bv[s].x};
      }
      Dft<R>::run(A);
      Dft<R>::run(Bv);
      const int jA = t;
      const int jB = t0 ? S0 / 2 : S0 - t;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int sa = (C::PADQ == R) ? jA * (R + 1) + r : C::pos(jA * R + r);
        const int sb = (C::PADQ == R) ? jB * (R + 1) + r : C::pos(jB * R + r);
        buf[sa] = A[r];
        buf[sb] = Bv[r];
      }
    }
    inv_group_sync<C>();

    // ---- middle pass -----------------------------------------------------------
    if constexpr (C::NPASS == 3) {
      constexpr int R = C::R1, S = M / R, NB = S / T, NS = C::R0;
      constexpr int TWS = 2 * (M / (NS * R));
      float2 v[NB][R];
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) {
        const int j = t + bb * T;
        const int jm = j % NS;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float2 x = buf[C::template pos_strided<S>(j, r)];
          if (r > 0) x = cmul(x, tw[jm * r * TWS]);
          v[bb][r] = x;
        }
      }
      inv_group_sync<C>();
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) Dft<R>::run(v[bb]);
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) {
        const int j = t + bb * T;
        const int base = (j / NS) * NS * R + (j % NS);
#pragma unroll
        for (int r = 0; r < R; ++r) buf[C::template pos_strided<NS>(base, r)] = v[bb][r];
      }
      inv_group_sync<C>();
    }

    // ---- last pass: outputs are time samples ------------------------------------
    {
      constexpr int R = C::RL, NS = M / R, NB = NS / T;
      float2 v[NB][R];
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) {
        const int j = t + bb * T;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float2 x = buf[C::template pos_strided<NS>(j, r)];
          if (r > 0) x = cmul(x, tw[2 * j * r]);
          v[bb][r] = x;
        }
      }
      inv_group_sync<C>();  // next iteration overwrites buf
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) Dft<R>::run(v[bb]);
      if (valid) {
        float2* __restrict__ orow =
            reinterpret_cast<float2*>(p.frames + (b * p.F + f) * (long long)N);
        const float sc = 0.5f * inv_m;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
          const int j = t + bb * T;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            // swap back: x[2n] = Re z[n] = v.y, x[2n+1] = Im z[n] = v.x
            orow[j + r * NS] = {v[bb][r].y * sc, v[bb][r].x * sc};
          }
        }
      }
    }
  }
}

//                    N     T   R0  R1  R2  PADQ
using Inv256 = InvCfg<256, 8, 8, 16, 1, 8>;
using Inv400 = InvCfg<400, 10, 10, 20, 1, 10>;
using Inv512 = InvCfg<512, 32, 4, 8, 8, 4>;
using Inv800 = InvCfg<800, 10, 20, 20, 1, 20>;
using Inv1024 = InvCfg<1024, 32, 8, 8, 8, 8>;
using Inv2048 = InvCfg<2048, 64, 8, 8, 16, 8>;
using Inv4096 = InvCfg<4096, 128, 8, 16, 16, 8>;

template <class C, int WG, int ITERS>
static int launch_inv(const InvParams& p0, hipStream_t s) {
  constexpr int G = C::template groups<WG>();
  constexpr int FT = G * ITERS;
  InvParams p = p0;
  p.tiles = (int)((p.F + FT - 1) / FT);
  const size_t lds = sizeof(float2) * G * C::LBUF;
  static_assert(sizeof(float2) * G * C::LBUF <= 160 * 1024, "LDS budget");
  const long long blocks = p.B * p.tiles;
  if (blocks <= 0) return MIA_OK;
  if (blocks > 0x7fffffffLL) {
    set_error("grid too large: %lld workgroups", blocks);
    return MIA_ERR_INVALID;
  }
  auto kern = inv_kernel<C, WG, ITERS>;
  if (lds > 64 * 1024) {
    MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WG), lds, s, p);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

int launch_inverse(const mia_plan* plan, const InvParams& p, hipStream_t s) {
  switch (plan->n_fft) {
    case 256: return launch_inv<Inv256, 256, 4>(p, s);
    case 400: return launch_inv<Inv400, 256, 4>(p, s);
    case 512: return launch_inv<Inv512, 256, 4>(p, s);
    case 800: return launch_inv<Inv800, 128, 4>(p, s);
    case 1024: return launch_inv<Inv1024, 256, 4>(p, s);
    case 2048: return launch_inv<Inv2048, 256, 4>(p, s);
    case 4096: return launch_inv<Inv4096, 256, 4>(p, s);
    default:
      set_error("n_fft=%d has no compiled FFT plan", plan->n_fft);
      return MIA_ERR_UNSUPPORTED;
  }
}

}  // namespace mia

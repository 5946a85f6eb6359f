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

template <int N_, int T_, int R0_, int R1_, int R2_, int PADQ_, int R3_ = 1>
struct InvCfg {
  static constexpr int N = N_, M = N_ / 2, T = T_;
  static constexpr int R0 = R0_, R1 = R1_, R2 = R2_, R3 = R3_;
  static constexpr int NPASS = 2 + (R2_ > 1) + (R3_ > 1);
  static constexpr int RL = (NPASS == 4) ? R3_ : ((NPASS == 3) ? R2_ : R1_);
  static constexpr int S0 = M / R0_;
  static constexpr int PADQ = PADQ_;
  static constexpr int LBUF = M + M / PADQ_ + 2;
  static constexpr bool MULTIWAVE = (T_ > 64);
  static constexpr int GPW = MULTIWAVE ? 0 : 64 / T_;
  static_assert(R0_ * R1_ * R2_ * R3_ == M, "radices must multiply to n_fft/2");
  static_assert(R3_ == 1 || R2_ > 1, "a fourth pass needs a third");
  static_assert(NPASS < 4 || (M / R2_) % T_ == 0, "pass 2 must split evenly");
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

// Complex products with the rounding spelled out (one product rounded, the other fused), so
// that every kernel built on inverse_frame gives the same bits whether its twiddle operand
// sits in a register or comes from the table - left to the compiler, the choice of which
// product to fuse follows the operand's origin.
__device__ __forceinline__ float2 cmul_x(float2 a, float2 w) {
  return {__builtin_fmaf(a.x, w.x, -(a.y * w.y)), __builtin_fmaf(a.x, w.y, a.y * w.x)};
}
__device__ __forceinline__ float2 cmul_conj_x(float2 a, float2 w) {  // a * conj(w)
  return {__builtin_fmaf(a.x, w.x, a.y * w.y), __builtin_fmaf(a.y, w.x, -(a.x * w.y))};
}

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

// Inverse real FFT of one frame by the group of T threads that owns `buf`.  `xs` points
// at the frame's bin 0 (stride p.sk between bins); `sink(n, v)` receives the time samples
// (x[2n], x[2n+1]) = v, already scaled by 1/N.  With `to_buf` the samples are instead left
// in buf[n] (plain layout) once every thread of the group has read its last-pass inputs.
// The bins a thread needs in pass 0: X[k_s], X[M - k_s] for its R0 slots, plus X[M/2].
template <class C>
struct FrameBins {
  float2 xk[C::R0], xm[C::R0], xh;
  bool ready = false;  // holds the bins of the frame to transform next
};
template <class C>
__device__ __forceinline__ void fetch_bins(FrameBins<C>& fb, const InvParams& p,
                                           const float2* __restrict__ xs, bool valid, int t) {
  constexpr int M = C::M, R = C::R0, S0 = C::S0;
  const bool t0 = (t == 0);
#pragma unroll
  for (int s = 0; s < R; ++s) {
    // generic thread: k = t + S0*s; thread 0: its own table
    const T0Slot ts = t0_slot<R>(s);
    const int k = t0 ? ts.k_half * (S0 / 2) : t + S0 * s;
    fb.xk[s] = {0.0f, 0.0f};
    fb.xm[s] = {0.0f, 0.0f};
    if (valid) {
      fb.xk[s] = xs[(long long)k * p.sk];
      fb.xm[s] = xs[(long long)(M - k) * p.sk];
    }
  }
  fb.xh = {0.0f, 0.0f};
  if (valid && t0) fb.xh = xs[(long long)(M / 2) * p.sk];
}

// Every twiddle factor a thread needs for a frame, for kernels that keep them in registers
// across their run of frames (the table loads otherwise sit in front of every pass).
template <class C>
struct FrameTwiddles {
  static constexpr int NB1 = C::NPASS >= 3 ? (C::M / C::R1) / C::T : 1;
  static constexpr int NB2 = C::NPASS == 4 ? (C::M / C::R2) / C::T : 1;
  static constexpr int NBL = (C::M / C::RL) / C::T;
  float2 p0[C::R0], m1[NB1][C::R1], m2[NB2][C::R2], pl[NBL][C::RL];
  __device__ __forceinline__ void load(const float2* __restrict__ tw, int t) {
    constexpr int M = C::M, T = C::T, S0 = C::S0;
    static_for<0, C::R0>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr T0Slot ts = t0_slot<C::R0>(s);
      p0[s] = tw[t == 0 ? ts.k_half * (S0 / 2) : t + S0 * s];
    });
    if constexpr (C::NPASS >= 3) {
      constexpr int NS = C::R0, TWS = 2 * (M / (NS * C::R1));
#pragma unroll
      for (int bb = 0; bb < NB1; ++bb) {
#pragma unroll
        for (int r = 1; r < C::R1; ++r) m1[bb][r] = tw[((t + bb * T) % NS) * r * TWS];
      }
    }
    if constexpr (C::NPASS == 4) {
      constexpr int NS = C::R0 * C::R1, TWS = 2 * (M / (NS * C::R2));
#pragma unroll
      for (int bb = 0; bb < NB2; ++bb) {
#pragma unroll
        for (int r = 1; r < C::R2; ++r) m2[bb][r] = tw[((t + bb * T) % NS) * r * TWS];
      }
    }
#pragma unroll
    for (int bb = 0; bb < NBL; ++bb) {
#pragma unroll
      for (int r = 1; r < C::RL; ++r) pl[bb][r] = tw[2 * (t + bb * T) * r];
    }
    // opaque from here on: never re-fetched inside the loop
#pragma unroll
    for (int s = 0; s < C::R0; ++s) asm volatile("" : "+v"(p0[s].x), "+v"(p0[s].y));
    if constexpr (C::NPASS >= 3) {
#pragma unroll
      for (int bb = 0; bb < NB1; ++bb) {
#pragma unroll
        for (int r = 1; r < C::R1; ++r) asm volatile("" : "+v"(m1[bb][r].x), "+v"(m1[bb][r].y));
      }
    }
    if constexpr (C::NPASS == 4) {
#pragma unroll
      for (int bb = 0; bb < NB2; ++bb) {
#pragma unroll
        for (int r = 1; r < C::R2; ++r) asm volatile("" : "+v"(m2[bb][r].x), "+v"(m2[bb][r].y));
      }
    }
#pragma unroll
    for (int bb = 0; bb < NBL; ++bb) {
#pragma unroll
      for (int r = 1; r < C::RL; ++r) asm volatile("" : "+v"(pl[bb][r].x), "+v"(pl[bb][r].y));
    }
  }
};

// `fb` holds this frame's bins on entry if fb.ready (else they are fetched here), and on
// exit the bins of the frame at `xs_next` (fetched behind the last pass's LDS reads, so
// that their latency hides behind the last butterflies and the stores).
// `rt`: the thread's twiddles in registers, or nullptr (then they come from the table `tw`).
// `gain(k)`: a real factor for bin k, applied as the bins enter the split step (NoGain: none).
struct NoGain {};
template <class C, bool TO_BUF, bool RT, class Sink, class Gain = NoGain>
__device__ __forceinline__ void inverse_frame(const InvParams& p, const float2* __restrict__ tw,
                                              float2* __restrict__ buf,
                                              const float2* __restrict__ xs, bool valid, int t,
                                              FrameBins<C>& fb, const float2* __restrict__ xs_next,
                                              bool valid_next, Sink sink,
                                              const FrameTwiddles<C>* rt = nullptr,
                                              Gain gain = Gain{}) {
  constexpr int M = C::M, T = C::T;
  constexpr bool GATED = !std::is_same<Gain, NoGain>::value;
  const bool t0 = (t == 0);
  const float inv_m = 1.0f / float(M);
  if (!fb.ready) fetch_bins<C>(fb, p, xs, valid, t);
  {
    // ---- pass 0: bins -> Z (split step) -> radix R0 -> LDS -------------------
  {
    constexpr int R = C::R0, S0 = C::S0;
    float2 zk[R], zmk[R];
#pragma unroll
    for (int s = 0; s < R; ++s) {
      // generic thread: k = t + S0*s; thread 0: table below
      const T0Slot ts = t0_slot<R>(s);
      const int k = t0 ? ts.k_half * (S0 / 2) : t + S0 * s;
      float2 xk = fb.xk[s], xm = fb.xm[s];
      if constexpr (GATED) {
        const float gk = gain(k), gm = gain(M - k);
        xk = {xk.x * gk, xk.y * gk};
        xm = {xm.x * gm, xm.y * gm};
      }
      if (k == 0) {  // the real transform ignores Im X[0] and Im X[M]
        xk.y = 0.0f;
        xm.y = 0.0f;
      }
      float2 w;
      if constexpr (RT) w = rt->p0[s]; else w = tw[k];
      // E = (X[k] + conj X[M-k])/2, O = conj(w) (X[k] - conj X[M-k])/2,
      // Z[k] = E + iO, Z[M-k] = conj(E) + i conj(O).  The 1/2 is folded
      // into the final scale.
      const float2 E = {xk.x + xm.x, xk.y - xm.y};
      const float2 D = {xk.x - xm.x, xk.y + xm.y};
      const float2 O = cmul_conj_x(D, w);
      zk[s] = {E.x - O.y, E.y + O.x};
      zmk[s] = {E.x + O.y, O.x - E.y};
    }
    // The bins are consumed: the next frame's are fetched here, so that their way from memory
    // is covered by the remaining passes (behind the last pass there is too little left).
    float2 xh_cur = fb.xh;
    if constexpr (GATED) {
      if (t0) {
        const float gh = gain(M / 2);
        xh_cur = {xh_cur.x * gh, xh_cur.y * gh};
      }
    }
    fb.ready = xs_next != nullptr;
    if (fb.ready) fetch_bins<C>(fb, p, xs_next, valid_next, t);
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
      const float2 xh = xh_cur;
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

  // ---- middle passes (one in 3-pass plans, two in 4-pass plans) -------------------
  auto middle = [&](auto Rc, auto NSc, auto Wc) {
    constexpr int R = decltype(Rc)::value, NS = decltype(NSc)::value, WHICH = decltype(Wc)::value;
    constexpr int S = M / R, NB = S / T;
    constexpr int TWS = 2 * (M / (NS * R));
    float2 v[NB][R];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
      const int j = t + bb * T;
      const int jm = j % NS;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float2 x = buf[C::template pos_strided<S>(j, r)];
        if (r > 0) {
          if constexpr (RT) {
            if constexpr (WHICH == 1) x = cmul_x(x, rt->m1[bb][r]); else x = cmul_x(x, rt->m2[bb][r]);
          } else {
            x = cmul_x(x, tw[jm * r * TWS]);
          }
        }
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
  };
  if constexpr (C::NPASS >= 3) {
    middle(std::integral_constant<int, C::R1>{}, std::integral_constant<int, C::R0>{},
           std::integral_constant<int, 1>{});
  }
  if constexpr (C::NPASS == 4) {
    middle(std::integral_constant<int, C::R2>{}, std::integral_constant<int, C::R0 * C::R1>{},
           std::integral_constant<int, 2>{});
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
        if (r > 0) {
          if constexpr (RT) x = cmul_x(x, rt->pl[bb][r]); else x = cmul_x(x, tw[2 * j * r]);
        }
        v[bb][r] = x;
      }
    }
    inv_group_sync<C>();  // buf is free again
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) Dft<R>::run(v[bb]);
    const float sc = 0.5f * inv_m;
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
      const int j = t + bb * T;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        // swap back: x[2n] = Re z[n] = v.y, x[2n+1] = Im z[n] = v.x
        const float2 o = {v[bb][r].y * sc, v[bb][r].x * sc};
        if constexpr (TO_BUF) {
          buf[j + r * NS] = o;
        } else {
          sink(j + r * NS, o, bb + NB * r);  // third argument: (n - t) / T, a constant after unrolling
        }
      }
    }
  }
  }
}

template <class C, int WG, int ITERS>
__global__ __launch_bounds__(WG) void inv_kernel(InvParams p) {
  constexpr int N = C::N;
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

  if (!active) return;
  FrameBins<C> fb;
  constexpr bool RT = C::NPASS == 4;  // the lean plans keep their twiddles in registers
  FrameTwiddles<C> rt;
  if constexpr (RT) rt.load(tw, t);
  for (int it = 0; it < ITERS; ++it) {
    const long long f = f0 + it * G + g;
    const bool valid = f < p.F;
    if (!C::MULTIWAVE && !valid) break;  // later frames of this group do not exist either
    const float2* __restrict__ xs = p.spec + b * p.sb + (valid ? f : 0) * p.sf;
    float2* __restrict__ orow =
        reinterpret_cast<float2*>(p.frames + (b * p.F + (valid ? f : 0)) * (long long)N);
    // (no prefetch of the next frame's bins here: it pushes the 4096-point plan past
    // 256 VGPRs, measured 2.43 against 2.24 ms on the round-trip workload)
    fb.ready = false;
    inverse_frame<C, false, RT>(
        p, tw, buf, xs, valid, t, fb, nullptr, false,
        [&](int n, float2 v, int) {
          if (valid) orow[n] = v;
        },
        &rt);
  }
}

// Fused istft for hop = n_fft / 4: inverse FFT + windowed overlap-add + normalisation by the
// window envelope, no frame workspace.  A persistent workgroup (= one frame's T threads) walks
// a contiguous run of FT-frame tiles; within a row it carries the n_fft - hop samples that
// later frames still add to in an LDS ring, and when it starts in the middle of a row it first
// re-transforms (without storing) the tile whose frames reach into its run.
// In the last pass thread t receives the sample pairs n = t + T c, c = 0 .. M / T - 1, of EVERY
// frame, and hop is a multiple of 2 T, so in the overlap ring those are always the same slots
// (q = 2 t, 2 t + 1 mod 2 T): a thread owns its slots.  It therefore adds its outputs to the
// ring straight from registers - no frame buffer round trip and no workgroup barrier around the
// accumulation - normalises and stores them itself (8-byte stores, 512 B per wave), and the
// next frame's bins are fetched as soon as pass 0 has consumed this frame's.  Arithmetic and
// its order equal ola_kernel's (fma(w, x, sum) over ascending frames, fma(w, w, wsum),
// sum / max(wsum, 1e-8) * scale): the results are the same bits.
template <class C, int FT>
__global__ __launch_bounds__(C::T) void istft_own_kernel(InvParams p) {
  constexpr int N = C::N, M = N / 2, HOP = N / 4, WG = C::T;
  constexpr int NC = M / WG;          // sample pairs per thread and frame
  constexpr int SPAN = FT * HOP;      // samples a tile completes
  constexpr int RING = SPAN + N;      // accumulator length
  constexpr int HC = HOP / (2 * WG);  // pairs per thread and hop
  static_assert(HOP % (2 * WG) == 0 && HC >= 1, "hop must be a multiple of twice the workgroup size");
  static_assert(FT >= 3, "one warm-up tile must cover the three frames that reach into a run");
  static_assert((M / C::RL) % WG == 0 && (WG % 64) == 0, "a frame is the whole workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2* buf = reinterpret_cast<float2*>(smem_raw);
  float2* acc = buf + C::LBUF;  // RING / 2 sample pairs

  const int t = threadIdx.x;
  const float2* __restrict__ tw = p.tw;
  const float2* __restrict__ win2 = reinterpret_cast<const float2*>(p.win);

  const long long total = p.B * p.tiles;
  const long long first = (long long)blockIdx.x * p.tiles_per_wg;
  const long long last = first + p.tiles_per_wg < total ? first + p.tiles_per_wg : total;
  if (first >= last) return;

  float2 wreg[NC];  // window taps of the thread's pairs
#pragma unroll
  for (int c = 0; c < NC; ++c) wreg[c] = win2[t + c * WG];
  // envelope of a sample whose four frames all exist: sum of w^2 over positions r + c hop,
  // c = 3..0 (ascending frames = descending position in the frame)
  float2 wsq[HC];
#pragma unroll
  for (int m = 0; m < HC; ++m) {
    float2 ws = {0.0f, 0.0f};
#pragma unroll
    for (int c = 3; c >= 0; --c) {
      const float2 w = wreg[m + c * HC];
      ws = {fmaf(w.x, w.x, ws.x), fmaf(w.y, w.y, ws.y)};
    }
    wsq[m] = ws;
  }

  long long idx = (first % p.tiles) > 0 ? first - 1 : first;  // one warm-up tile inside a row
  for (int q = t; q < RING / 2; q += WG) acc[q] = {0.0f, 0.0f};
  int origin = 0;  // ring slot (in pairs) of the tile's first sample
  auto slot = [&](int q) {
    const int s2 = origin + q;
    return s2 >= RING / 2 ? s2 - RING / 2 : s2;
  };
  __syncthreads();

  FrameTwiddles<C> rt;
  rt.load(tw, t);
  FrameBins<C> fb;
  fb.ready = false;
  for (; idx < last; ++idx) {
    const long long b = idx / p.tiles;
    const int tile = (int)(idx - b * p.tiles);
    const long long f0 = (long long)tile * FT;
    const int nf = p.F - f0 < FT ? (int)(p.F - f0) : FT;  // frames of this tile that exist
    const float2* __restrict__ xrow = p.spec + b * p.sb;
#pragma unroll 1
    for (int it = 0; it < FT; ++it) {
      const bool valid = it < nf;
      const float2* __restrict__ xs = xrow + (valid ? f0 + it : 0) * p.sf;
      // the frame after this one (same tile or the next tile of the run), for the prefetch
      const float2* xs_next = nullptr;
      bool valid_next = false;
      if (it + 1 < FT) {
        valid_next = it + 1 < nf;
        xs_next = xrow + (valid_next ? f0 + it + 1 : 0) * p.sf;
      } else if (idx + 1 < last) {
        const long long bn = (idx + 1) / p.tiles;
        const long long fn = ((idx + 1) - bn * p.tiles) * FT;
        valid_next = true;  // the first frame of a tile always exists
        xs_next = p.spec + bn * p.sb + fn * p.sf;
      }
      const int qf = it * (HOP / 2) + t;  // pair index of the thread's first pair of this frame
      inverse_frame<C, false, true>(
          p, tw, buf, xs, valid, t, fb, xs_next, valid_next,
          [&](int, float2 v, int c) {
            if (valid) {
              const int a = slot(qf + c * WG);
              const float2 sum = acc[a];
              acc[a] = {fmaf(wreg[c].x, v.x, sum.x), fmaf(wreg[c].y, v.y, sum.y)};
            }
          },
          &rt);
    }
    // ---- the first SPAN samples are complete (all of them at the end of a row) ------------
    const long long base = f0 * HOP;  // output position of the tile's first sample
    const bool row_end = tile == p.tiles - 1;
    const long long left = p.out_len - base;  // samples of the row from `base` on
    float* __restrict__ orow = p.out + b * p.out_len + base;
    const bool aligned8 = (reinterpret_cast<uintptr_t>(orow) & 7) == 0;
    if (!row_end && f0 >= 3 && nf == FT && left >= SPAN && aligned8) {
      // every sample of the span has its four frames: the thread's own pairs, envelope from registers
#pragma unroll
      for (int k = 0; k < SPAN / (2 * WG); ++k) {
        const int a = slot(t + k * WG);
        const float2 sum = acc[a];
        acc[a] = {0.0f, 0.0f};
        if (idx >= first) {
          const float2 ws = wsq[k % HC];
          reinterpret_cast<float2*>(orow)[t + k * WG] = {sum.x / fmaxf(ws.x, 1e-8f) * p.scale,
                                                         sum.y / fmaxf(ws.y, 1e-8f) * p.scale};
        }
      }
    } else {
      // row ends and the first tile of a row: any thread takes any sample
      __syncthreads();
      float* __restrict__ accf = reinterpret_cast<float*>(acc);
      const int avail = row_end ? RING : SPAN;
      const int count = left < avail ? (left < 0 ? 0 : (int)left) : avail;
      for (int q = t; q < avail; q += WG) {
        int a = 2 * origin + q;
        a = a >= RING ? a - RING : a;
        const float sum = accf[a];
        accf[a] = 0.0f;
        if (q < count && idx >= first) {
          // frames f0 + fa .. f0 + fb contain this sample (relative to the tile, may be < 0)
          int fa = q - N + 1;
          fa = fa <= 0 ? -((-fa) / HOP) : (fa + HOP - 1) / HOP;
          if (f0 + fa < 0) fa = (int)-f0;
          int fbx = q / HOP;
          if (f0 + fbx > p.F - 1) fbx = (int)(p.F - 1 - f0);
          float wsum = 0.0f;
          for (int f = fa; f <= fbx; ++f) {
            const float w = p.win[q - f * HOP];
            wsum = fmaf(w, w, wsum);
          }
          orow[q] = sum / fmaxf(wsum, 1e-8f) * p.scale;
        }
      }
      if (row_end && idx >= first) {  // anything past the accumulator is beyond the last frame: zeros
        for (long long i = base + RING + t; i < p.out_len; i += WG) p.out[b * p.out_len + i] = 0.0f;
      }
      __syncthreads();
    }
    if (row_end) {
      origin = 0;
      fb.ready = fb.ready && idx + 1 < last;  // (the prefetched frame is the next row's first: still right)
    } else {
      origin = slot(SPAN / 2);
    }
  }
}

// istft_own_kernel for the spectral gate: the spectrum it reads is the un-gated STFT, (B, F,
// bins) with contiguous rows of bins, and every bin is scaled on its way into the split step by
// the gate's mask - soft mask from |z| and a per-bin threshold, averaged over s1 and then s2
// bins and then over st frames, then the keep step - so neither the mask nor the gated spectrum
// ever exists in memory.  The workgroup keeps the st rows of twice-averaged mask values that the
// frame it is about to transform needs in an LDS ring (row of frame f in slot f mod st) and makes
// the next row just before it is needed: mask values of one frame into the (idle) transform
// buffer, windows summed from there.  Indices beyond the ends repeat the end value in both
// directions, every mean is its window's double-precision sum times 1 / size rounded to float32,
// in the order the separate kernels of the gate take them.  A run that starts inside a row makes
// the rows its warm-up frames need first.  Element i of a row sits at i + i / 8, so that threads
// on runs of eight bins and threads on consecutive bins both spread over the banks.
constexpr int kGateMaxS1 = 16, kGateMaxS2 = 4, kGateMaxSt = 8;
constexpr int kGateNO = 9 + kGateMaxS2 - 1;        // first averages a thread makes at most
constexpr int kGateNW = kGateNO + kGateMaxS1 - 1;  // mask values it reads for them at most
__host__ __device__ constexpr int gate_pos(int i) { return i + (i >> 3); }
// (the ring's rows are plain: a padded row would cost the 2048-point kernel a workgroup per CU)
template <int M>
constexpr int gate_row_stride() { return M + 4; }

// |z| of a spectrum value (no scaling against overflow: an STFT of float32 audio is far from it)
__device__ __forceinline__ float gate_abs(float2 z) {
#ifdef MIA_GATE_HYPOT
  return hypotf(z.x, z.y);
#else
  return sqrtf(__builtin_fmaf(z.x, z.x, z.y * z.y));
#endif
}

// m1[i] = mean of w[i .. i + S1 - 1], i < kGateNO: the window's double-precision sum times 1 / S1,
// as box_filter_kernel takes it.  The sum slides (one value in, one out): sums of up to sixteen
// float32 mask values are exact in double precision unless one of them is below 2^-25 of the
// largest, so this is the ascending sum to the last bit wherever that bit means anything.
template <int S1>
__device__ __forceinline__ void gate_window_means(const float* __restrict__ src, int first,
                                                  float (&m1)[kGateNO], double inv) {
  float w[kGateNO + S1 - 1];  // element i of the row sits at i + i / 8
#pragma unroll
  for (int x = 0; x < kGateNO + S1 - 1; ++x) w[x] = src[gate_pos(first + x)];
  if constexpr (S1 == 1) {
#pragma unroll
    for (int i = 0; i < kGateNO; ++i) m1[i] = w[i];
  } else {
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < S1; ++j) sum += (double)w[j];
    m1[0] = window_mean(sum, inv);
#pragma unroll
    for (int i = 1; i < kGateNO; ++i) {
      sum += (double)w[i + S1 - 1];
      sum -= (double)w[i - 1];
      m1[i] = window_mean(sum, inv);
    }
  }
}

template <class C, int FT>
__global__ __launch_bounds__(C::T, 2) void istft_gate_kernel(InvParams p, GateParams g) {
  constexpr int N = C::N, M = N / 2, HOP = N / 4, WG = C::T;
  constexpr int NC = M / WG;
  constexpr int SPAN = FT * HOP;
  constexpr int RING = SPAN + N;
  constexpr int HC = HOP / (2 * WG);
  constexpr int RS = gate_row_stride<M>();
  static_assert(HOP % (2 * WG) == 0 && HC >= 1, "hop must be a multiple of twice the workgroup size");
  static_assert(FT >= 3, "one warm-up tile must cover the three frames that reach into a run");
  static_assert((M / C::RL) % WG == 0 && (WG % 64) == 0, "a frame is the whole workgroup");
  static_assert(M == 8 * WG, "a thread makes the mask values of eight bins");
  static_assert(gate_pos(M + kGateNW + 2) + 1 <= 2 * C::LBUF, "a frame's mask values must fit the transform buffer");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2* buf = reinterpret_cast<float2*>(smem_raw);
  float2* acc = buf + C::LBUF;  // RING / 2 sample pairs
  float* ring = reinterpret_cast<float*>(acc) + RING;  // g.st rows of RS mask values
  float* tmp0 = reinterpret_cast<float*>(buf);

  const int t = threadIdx.x;
  const float2* __restrict__ tw = p.tw;
  const float2* __restrict__ win2 = reinterpret_cast<const float2*>(p.win);

  const long long total = p.B * p.tiles;
  const long long first = (long long)blockIdx.x * p.tiles_per_wg;
  const long long last = first + p.tiles_per_wg < total ? first + p.tiles_per_wg : total;
  if (first >= last) return;

  float2 wreg[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) wreg[c] = win2[t + c * WG];
  float2 wsq[HC];
#pragma unroll
  for (int m = 0; m < HC; ++m) {
    float2 ws = {0.0f, 0.0f};
#pragma unroll
    for (int c = 3; c >= 0; --c) {
      const float2 w = wreg[m + c * HC];
      ws = {fmaf(w.x, w.x, ws.x), fmaf(w.y, w.y, ws.y)};
    }
    wsq[m] = ws;
  }

  long long idx = (first % p.tiles) > 0 ? first - 1 : first;
  for (int q = t; q < RING / 2; q += WG) acc[q] = {0.0f, 0.0f};
  int origin = 0;
  auto slot = [&](int q) {
    const int s2 = origin + q;
    return s2 >= RING / 2 ? s2 - RING / 2 : s2;
  };
  __syncthreads();

  // ---- the mask rows ---------------------------------------------------------------------
  const int s1 = g.s1, s2w = g.s2, st = g.st;
  const int h1 = s1 / 2, h2 = s2w / 2;
  const int back = st / 2, ahead = (st - 1) / 2;  // frames of the time window before / after its own
  const int nF = (int)p.F;
  const int last_slot = (nF - 1) % st;
  const double inv1 = window_inv(s1), inv2 = window_inv(s2w), invt = window_inv(st);
  // A row's spectrum values are fetched a row ahead (they arrive behind a transform): mz holds
  // bins t + i WG of frame pf_f of row pf_b, mzM bin M (last thread).
  float2 mz[8], mzM = {0.0f, 0.0f};
  long long pf_b = -1, th_b = -1;
  int pf_f = -1;
  float thv[8], thM = 0.0f;
  auto fetch_row = [&](long long b, int fr) {
    const float2* __restrict__ zr = p.spec + b * p.sb + (long long)fr * p.sf;
#pragma unroll
    for (int i = 0; i < 8; ++i) mz[i] = zr[t + i * WG];
    if (t == WG - 1) mzM = zr[M];
    pf_b = b;
    pf_f = fr;
  };
  // tmp0[gate_pos(kGateMaxS2 / 2 + h1 + k)] = mask value of bin k, the end values repeated on
  // both sides; the window of the first average at bin e starts at kGateMaxS2 / 2 + e.
  constexpr int SH = kGateMaxS2 / 2;
  auto mask_row = [&](long long b, int fr) {
    if (pf_b != b || pf_f != fr) fetch_row(b, fr);
    if (th_b != b) {  // the row's thresholds stay in registers for all of its frames
      const float* __restrict__ th = g.thr + b * g.thr_stride;
#pragma unroll
      for (int i = 0; i < 8; ++i) thv[i] = th[t + i * WG];
      thM = th[M];
      th_b = b;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = t + i * WG;
      const float v = gate_mask_value(gate_abs(mz[i]), thv[i], g.gain);
      tmp0[gate_pos(SH + h1 + k)] = v;
      if (k == 0) {
        for (int j = 0; j < SH + h1; ++j) tmp0[gate_pos(j)] = v;
      }
    }
    if (t == WG - 1) {
      // (the end value goes on past the window's reach to the last element any thread reads: the
      // sliding sums must never take in what an earlier transform left here)
      const float v = gate_mask_value(gate_abs(mzM), thM, g.gain);
      for (int j = 0; j <= s1 + 2; ++j) tmp0[gate_pos(SH + h1 + M + j)] = v;
    }
    if (fr + 1 < nF) fetch_row(b, fr + 1);
    __syncthreads();
    // thread t: bins 8 t .. 8 t + 7 (the last thread also bin M).  Its first averages are those
    // at bins e0 + i, e0 = 8 t - s2 / 2; beyond the ends they repeat the end's.
    const int e0 = 8 * t - h2;
    float m1[kGateNO];
    switch (s1) {
#define MIA_GATE_S1(S) case S: gate_window_means<S>(tmp0, SH + e0, m1, inv1); break;
      MIA_GATE_S1(1) MIA_GATE_S1(2) MIA_GATE_S1(3) MIA_GATE_S1(4) MIA_GATE_S1(5) MIA_GATE_S1(6)
      MIA_GATE_S1(7) MIA_GATE_S1(8) MIA_GATE_S1(9) MIA_GATE_S1(10) MIA_GATE_S1(11) MIA_GATE_S1(12)
      MIA_GATE_S1(13) MIA_GATE_S1(14) MIA_GATE_S1(15)
#undef MIA_GATE_S1
      default: gate_window_means<16>(tmp0, SH + e0, m1, inv1); break;
    }
    if (t == 0) {  // bins below 0: the average at bin 0, which is m1[h2]
      const float at0 = h2 == 2 ? m1[2] : (h2 == 1 ? m1[1] : m1[0]);
      if (h2 >= 1) m1[0] = at0;
      if (h2 >= 2) m1[1] = at0;
    }
    if (t == WG - 1) {  // bins above M: the average at bin M, which is m1[8 + h2]
      const float atM = h2 == 2 ? m1[10] : (h2 == 1 ? m1[9] : m1[8]);
#pragma unroll
      for (int i = 9; i < kGateNO; ++i) {
        if (i > 8 + h2) m1[i] = atM;
      }
    }
    float* __restrict__ row = ring + (fr % st) * RS;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
      double sb = 0.0;
#pragma unroll
      for (int q = 0; q < kGateMaxS2; ++q) {
        if (q < s2w) sb += (double)m1[o + q];
      }
      const float m2 = s2w == 1 ? (float)sb : window_mean(sb, inv2);
      if (o < 8 || t == WG - 1) row[8 * t + o] = m2;
    }
    __syncthreads();
  };
  long long mb = -1;  // row whose mask rows the ring holds
  int mhi = -1;       // ... the last frame made

  FrameTwiddles<C> rt;
  rt.load(tw, t);
  FrameBins<C> fb;
  fb.ready = false;
  for (; idx < last; ++idx) {
    const long long b = idx / p.tiles;
    const int tile = (int)(idx - b * p.tiles);
    const long long f0 = (long long)tile * FT;
    const int nf = p.F - f0 < FT ? (int)(p.F - f0) : FT;
    const float2* __restrict__ xrow = p.spec + b * p.sb;
#pragma unroll 1
    for (int it = 0; it < FT; ++it) {
      const bool valid = it < nf;
      const int f = (int)f0 + it;
      if (valid) {
        if (b != mb) {
          mb = b;
          mhi = (f - back > 0 ? f - back : 0) - 1;
        }
        const int need = f + ahead < nF - 1 ? f + ahead : nF - 1;
        while (mhi < need) mask_row(b, ++mhi);
      }
      int off[kGateMaxSt];  // ring rows of the frames f - back .. f + ahead, ends repeated
      {
        int sl = (f - back + st) % st;  // (back < st)
#pragma unroll
        for (int q = 0; q < kGateMaxSt; ++q) {
          const int fq = f - back + q;
          off[q] = (fq < 0 ? 0 : (fq > nF - 1 ? last_slot : sl)) * RS;
          sl = sl + 1 == st ? 0 : sl + 1;
        }
      }
      auto gain_of = [&](int k) {
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < kGateMaxSt; ++q) {
          if (q < st) sum += (double)ring[off[q] + k];
        }
        float m = st == 1 ? (float)sum : window_mean(sum, invt);
        if (g.keep) m = 1.0f - g.prop * (1.0f - m);
        return m;
      };
      const float2* __restrict__ xs = xrow + (valid ? f : 0) * p.sf;
      const float2* xs_next = nullptr;
      bool valid_next = false;
      if (it + 1 < FT) {
        valid_next = it + 1 < nf;
        xs_next = xrow + (valid_next ? f + 1 : 0) * p.sf;
      } else if (idx + 1 < last) {
        const long long bn = (idx + 1) / p.tiles;
        const long long fn = ((idx + 1) - bn * p.tiles) * FT;
        valid_next = true;
        xs_next = p.spec + bn * p.sb + fn * p.sf;
      }
      const int qf = it * (HOP / 2) + t;
      inverse_frame<C, false, true>(
          p, tw, buf, xs, valid, t, fb, xs_next, valid_next,
          [&](int, float2 v, int c) {
            if (valid) {
              const int a = slot(qf + c * WG);
              const float2 sum = acc[a];
              acc[a] = {fmaf(wreg[c].x, v.x, sum.x), fmaf(wreg[c].y, v.y, sum.y)};
            }
          },
          &rt, gain_of);
    }
    const long long base = f0 * HOP;
    const bool row_end = tile == p.tiles - 1;
    const long long left = p.out_len - base;
    float* __restrict__ orow = p.out + b * p.out_len + base;
    const bool aligned8 = (reinterpret_cast<uintptr_t>(orow) & 7) == 0;
    if (!row_end && f0 >= 3 && nf == FT && left >= SPAN && aligned8) {
#pragma unroll
      for (int k = 0; k < SPAN / (2 * WG); ++k) {
        const int a = slot(t + k * WG);
        const float2 sum = acc[a];
        acc[a] = {0.0f, 0.0f};
        if (idx >= first) {
          const float2 ws = wsq[k % HC];
          reinterpret_cast<float2*>(orow)[t + k * WG] = {sum.x / fmaxf(ws.x, 1e-8f) * p.scale,
                                                         sum.y / fmaxf(ws.y, 1e-8f) * p.scale};
        }
      }
    } else {
      __syncthreads();
      float* __restrict__ accf = reinterpret_cast<float*>(acc);
      const int avail = row_end ? RING : SPAN;
      const int count = left < avail ? (left < 0 ? 0 : (int)left) : avail;
      for (int q = t; q < avail; q += WG) {
        int a = 2 * origin + q;
        a = a >= RING ? a - RING : a;
        const float sum = accf[a];
        accf[a] = 0.0f;
        if (q < count && idx >= first) {
          int fa = q - N + 1;
          fa = fa <= 0 ? -((-fa) / HOP) : (fa + HOP - 1) / HOP;
          if (f0 + fa < 0) fa = (int)-f0;
          int fbx = q / HOP;
          if (f0 + fbx > p.F - 1) fbx = (int)(p.F - 1 - f0);
          float wsum = 0.0f;
          for (int f = fa; f <= fbx; ++f) {
            const float w = p.win[q - f * HOP];
            wsum = fmaf(w, w, wsum);
          }
          orow[q] = sum / fmaxf(wsum, 1e-8f) * p.scale;
        }
      }
      if (row_end && idx >= first) {
        for (long long i = base + RING + t; i < p.out_len; i += WG) p.out[b * p.out_len + i] = 0.0f;
      }
      __syncthreads();
    }
    if (row_end) {
      origin = 0;
      fb.ready = fb.ready && idx + 1 < last;
    } else {
      origin = slot(SPAN / 2);
    }
  }
}

//                    N     T   R0  R1  R2  PADQ
using Inv256 = InvCfg<256, 8, 8, 16, 1, 8>;
// 400 and 800 points: twenty threads per frame and ten / twenty points per thread in four passes
// (the two-pass plans over ten threads needed 186 / 256 registers; inverse STFT of 256 x 30 s at
// hop n_fft / 4: 1.72 -> 1.55 ms and 1.90 -> 1.58 ms, most of which is the frame workspace's
// round trip through the overlap-add kernel; tools/ab_inv_sizes.sh).
#ifndef MIA_INV400
#define MIA_INV400 InvCfg<400, 20, 5, 2, 2, 5, 10>
#endif
#ifndef MIA_INV800
#define MIA_INV800 InvCfg<800, 20, 10, 2, 2, 10, 10>
#endif
using Inv400 = MIA_INV400;
using Inv512 = InvCfg<512, 32, 4, 8, 8, 4>;
using Inv800 = MIA_INV800;
// 1024 / 2048 / 4096 points: eight points per thread and pass (four passes of radix 4 / 8), a
// frame is one, two or four whole waves = the whole workgroup of the fused kernel
// (istft_own_kernel), twiddles resident in registers.  Measured on the HTDemucs round trip
// (128 rows x 180 s) against the three-pass plan <4096, 128, 8, 16, 16> with table twiddles and
// two frames per workgroup: inverse 9.9 -> 5.3 ms.  The table loads in front of
// every pass were the difference: the same four-pass plan with table twiddles took 11.2 ms.
// LDS layout (tools/ab_inv.sh, fused inverse of 256 x 30 s): one pad slot per 16 elements instead
// of per 8 takes the 1024-point plan from 0.90 to 0.63 ms (as in the forward plan of that size,
// the compiler pairs its LDS accesses, and pairs 8 elements apart collide) and the 2048-point
// plan from 0.89 to 0.87; at 4096 points the two are equal (5.38 ms for 128 rows x 180 s), one
// per 32 or per 4 loses everywhere.
using Inv1024 = InvCfg<1024, 64, 4, 8, 4, 16, 4>;
using Inv2048 = InvCfg<2048, 128, 4, 8, 8, 16, 4>;
using Inv4096 = InvCfg<4096, 256, 4, 8, 8, 8, 8>;

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

template <class C, int FT>
static int launch_fused_own(const InvParams& p0, hipStream_t s) {
  constexpr int WG = C::T;
  InvParams p = p0;
  p.tiles = (int)((p.F + FT - 1) / FT);
  const long long total = p.B * p.tiles;
  if (total <= 0) return MIA_OK;
  constexpr size_t lds = sizeof(float2) * C::LBUF + sizeof(float) * ((size_t)FT * (C::N / 4) + C::N);
  auto kern = istft_own_kernel<C, FT>;
  if (lds > 64 * 1024) {
    MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  int dev = 0, cus = 256, per_cu = 1;
  MIA_CHECK_HIP(hipGetDevice(&dev));
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), WG,
                                                   lds) != hipSuccess || per_cu < 1) {
    per_cu = 1;
  }
  long long grid = (long long)cus * per_cu;
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

#ifndef MIA_GATE_FT
#define MIA_GATE_FT 3  // frames per tile: with 3, four workgroups of the 2048-point kernel fit a CU's LDS
#endif
template <class C, int FT>
static int launch_gated_own(const InvParams& p0, const GateParams& g, hipStream_t s) {
  constexpr int WG = C::T;
  InvParams p = p0;
  p.tiles = (int)((p.F + FT - 1) / FT);
  const long long total = p.B * p.tiles;
  if (total <= 0) return MIA_OK;
  const size_t lds = sizeof(float2) * C::LBUF + sizeof(float) * ((size_t)FT * (C::N / 4) + C::N) +
                     sizeof(float) * (size_t)g.st * gate_row_stride<C::N / 2>();
  auto kern = istft_gate_kernel<C, FT>;
  if (lds > 64 * 1024) {
    MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  int dev = 0, cus = 256, per_cu = 1;
  MIA_CHECK_HIP(hipGetDevice(&dev));
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), WG,
                                                   lds) != hipSuccess || per_cu < 1) {
    per_cu = 1;
  }
  long long grid = (long long)cus * per_cu;
  if (grid > total) grid = total;
  const long long per_wg = (total + grid - 1) / grid;
  if (per_wg > 0x7fffffffLL) {
    set_error("batch too large for one launch: %lld tiles (split the batch)", total);
    return MIA_ERR_UNSUPPORTED;
  }
  p.tiles_per_wg = (int)per_wg;
  grid = (total + per_wg - 1) / per_wg;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WG), lds, s, p, g);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

// The gate inside the inverse: 1024, 2048 and 4096 points at hop n_fft / 4, windows the rows in LDS hold.
bool istft_gated_supported(int n_fft, int hop, int s1, int s2, int st) {
  return (n_fft == 1024 || n_fft == 2048 || n_fft == 4096) && hop * 4 == n_fft && s1 >= 1 && s1 <= kGateMaxS1 &&
         s2 >= 1 && s2 <= kGateMaxS2 && st >= 1 && st <= kGateMaxSt;
}

int launch_istft_gated(const mia_plan* plan, const InvParams& p, const GateParams& g, hipStream_t s) {
  if (!istft_gated_supported(plan->n_fft, p.hop, g.s1, g.s2, g.st)) {
    set_error("no gated inverse for n_fft=%d hop=%d windows %d / %d / %d", plan->n_fft, p.hop, g.s1,
              g.s2, g.st);
    return MIA_ERR_UNSUPPORTED;
  }
  switch (plan->n_fft) {
    case 1024: return launch_gated_own<Inv1024, MIA_GATE_FT>(p, g, s);
    case 2048: return launch_gated_own<Inv2048, MIA_GATE_FT>(p, g, s);
    default: return launch_gated_own<Inv4096, MIA_GATE_FT>(p, g, s);
  }
}

// The framings a fused kernel covers: hop = n_fft / 4 with hop a multiple of 256 on the compiled
// plans; the compiled-in plans of the per-wave transform at any hop from n_fft / 8 up.
static bool istft_own_supported(int n_fft, int hop) {
  return (n_fft == 1024 || n_fft == 2048 || n_fft == 4096) && hop * 4 == n_fft;
}
bool istft_fused_supported(int n_fft, int hop) {
  return istft_own_supported(n_fft, hop) || (!fft_size_supported(n_fft) && istft_wave_fused_supported(n_fft, hop));
}

int launch_istft_fused(const mia_plan* plan, const InvParams& p, hipStream_t s) {
  switch (plan->n_fft) {
    case 1024: return launch_fused_own<Inv1024, 4>(p, s);
    case 2048: return launch_fused_own<Inv2048, 4>(p, s);
    case 4096: return launch_fused_own<Inv4096, 4>(p, s);
    default:
      if (istft_wave_fused_supported(plan->n_fft, p.hop)) return launch_istft_wave_fused(plan, p, s);
      set_error("n_fft=%d has no fused inverse", plan->n_fft);
      return MIA_ERR_UNSUPPORTED;
  }
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

// Register-resident small DFTs and compile-time twiddles for the LDS-staged
// Stockham real FFT used by every spectral kernel in this library (gfx950).
//
// A length-N real transform runs as a length-M = N/2 complex transform of
// z[n] = x[2n] + i x[2n+1] followed by a split step.  The complex transform is
// 2 or 3 Stockham passes; each thread keeps whole radix-R butterflies
// (R in {8, 10, 16, 20, ...}) in registers and exchanges data through LDS once
// between passes.  Everything inside a butterfly is unrolled at compile time,
// so all inner twiddles are literal constants.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

namespace mia {

// ----------------------------------------------------------------------------
// compile-time loop
// ----------------------------------------------------------------------------
template <int I, int N, class F>
__device__ __host__ __forceinline__ constexpr void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(std::forward<F>(f));
  }
}

// ----------------------------------------------------------------------------
// constexpr sin/cos of 2*pi*p/q with exact octant reduction on the integers
// ----------------------------------------------------------------------------
constexpr double kPi = 3.14159265358979323846264338327950288;

constexpr double cx_sin_small(double x) {  // |x| <= pi/4
  double x2 = x * x, term = x, sum = x;
  for (int i = 1; i < 14; ++i) {
    term *= -x2 / double((2 * i) * (2 * i + 1));
    sum += term;
  }
  return sum;
}
constexpr double cx_cos_small(double x) {  // |x| <= pi/4
  double x2 = x * x, term = 1.0, sum = 1.0;
  for (int i = 1; i < 14; ++i) {
    term *= -x2 / double((2 * i - 1) * (2 * i));
    sum += term;
  }
  return sum;
}
struct cx_sc {
  double s, c;
};
constexpr cx_sc cx_sincos2pi(long long p, long long q) {
  p %= q;
  if (p < 0) p += q;
  long long o = (8 * p) / q;        // octant 0..7
  long long rho = 8 * p - o * q;    // 0 <= rho < q
  double phi = (kPi / 4) * double(rho) / double(q);
  double psi = (kPi / 4) * double(q - rho) / double(q);
  double sp = cx_sin_small(phi), cp = cx_cos_small(phi);
  double ss = cx_sin_small(psi), cs = cx_cos_small(psi);
  if (rho == 0) { sp = 0.0; cp = 1.0; }
  switch (o) {
    case 0: return {sp, cp};
    case 1: return {cs, ss};
    case 2: return {cp, -sp};
    case 3: return {ss, -cs};
    case 4: return {-sp, -cp};
    case 5: return {-cs, -ss};
    case 6: return {-cp, sp};
    default: return {-ss, cs};
  }
}

// ----------------------------------------------------------------------------
// complex helpers (float2 = re, im)
// ----------------------------------------------------------------------------
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
  return {a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x};
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 w) {  // a * conj(w)
  return {a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y};
}
__device__ __forceinline__ float2 cconj(float2 a) { return {a.x, -a.y}; }

// a * exp(-2*pi*i*K/N), K and N compile-time.
template <int K, int N>
__device__ __forceinline__ float2 mul_w(float2 a) {
  constexpr int k = ((K % N) + N) % N;
  if constexpr (k == 0) {
    return a;
  } else if constexpr (4 * k == N) {
    return {a.y, -a.x};
  } else if constexpr (2 * k == N) {
    return {-a.x, -a.y};
  } else if constexpr (4 * k == 3 * N) {
    return {-a.y, a.x};
  } else if constexpr (8 * k == N) {
    constexpr float h = 0.70710678118654752440f;
    return {(a.x + a.y) * h, (a.y - a.x) * h};
  } else if constexpr (8 * k == 3 * N) {
    constexpr float h = 0.70710678118654752440f;
    return {(a.y - a.x) * h, -(a.x + a.y) * h};
  } else if constexpr (8 * k == 5 * N) {
    constexpr float h = 0.70710678118654752440f;
    return {-(a.x + a.y) * h, (a.x - a.y) * h};
  } else if constexpr (8 * k == 7 * N) {
    constexpr float h = 0.70710678118654752440f;
    return {(a.x - a.y) * h, (a.x + a.y) * h};
  } else {
    constexpr cx_sc sc = cx_sincos2pi(k, N);
    constexpr float c = float(sc.c), s = float(sc.s);
    return {a.x * c + a.y * s, a.y * c - a.x * s};
  }
}

// ----------------------------------------------------------------------------
// Forward DFTs of length R on a register array (natural order in and out).
// ----------------------------------------------------------------------------
template <int R>
struct Dft;

template <>
struct Dft<1> {
  static __device__ __forceinline__ void run(float2 (&)[1]) {}
};

template <>
struct Dft<2> {
  static __device__ __forceinline__ void run(float2 (&v)[2]) {
    float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  }
};

template <>
struct Dft<3> {
  static __device__ __forceinline__ void run(float2 (&v)[3]) {
    constexpr float c = -0.5f;
    constexpr float s = 0.86602540378443864676f;
    float2 t = cadd(v[1], v[2]);
    float2 d = csub(v[1], v[2]);
    float2 m = {v[0].x + c * t.x, v[0].y + c * t.y};
    v[0] = cadd(v[0], t);
    v[1] = {m.x + s * d.y, m.y - s * d.x};
    v[2] = {m.x - s * d.y, m.y + s * d.x};
  }
};

template <>
struct Dft<4> {
  static __device__ __forceinline__ void run(float2 (&v)[4]) {
    float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
    float2 c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
    v[0] = cadd(a, c);
    v[2] = csub(a, c);
    v[1] = {b.x + d.y, b.y - d.x};  // b - i d
    v[3] = {b.x - d.y, b.y + d.x};  // b + i d
  }
};

template <>
struct Dft<5> {
  static __device__ __forceinline__ void run(float2 (&v)[5]) {
    constexpr float c1 = 0.30901699437494742410f;   // cos(2pi/5)
    constexpr float c2 = -0.80901699437494742410f;  // cos(4pi/5)
    constexpr float s1 = 0.95105651629515357212f;   // sin(2pi/5)
    constexpr float s2 = 0.58778525229247312917f;   // sin(4pi/5)
    float2 t1 = cadd(v[1], v[4]), t3 = csub(v[1], v[4]);
    float2 t2 = cadd(v[2], v[3]), t4 = csub(v[2], v[3]);
    float2 a1 = {v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y};
    float2 a2 = {v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y};
    float2 b1 = {s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y};
    float2 b2 = {s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y};
    v[0] = {v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y};
    v[1] = {a1.x + b1.y, a1.y - b1.x};  // a1 - i b1
    v[4] = {a1.x - b1.y, a1.y + b1.x};  // a1 + i b1
    v[2] = {a2.x + b2.y, a2.y - b2.x};
    v[3] = {a2.x - b2.y, a2.y + b2.x};
  }
};

// Cooley-Tukey composite R = A*B, all indices compile-time:
//   n = A*n2 + n1, k = B*k1 + k2
//   X[B k1 + k2] = sum_n1 W_A^{n1 k1} ( W_R^{n1 k2} sum_n2 x[A n2 + n1] W_B^{n2 k2} )
template <int A, int B>
struct DftCT {
  static __device__ __forceinline__ void run(float2 (&v)[A * B]) {
    float2 y[A][B];
    static_for<0, A>([&](auto n1c) {
      constexpr int n1 = decltype(n1c)::value;
      float2 t[B];
      static_for<0, B>([&](auto n2c) {
        constexpr int n2 = decltype(n2c)::value;
        t[n2] = v[A * n2 + n1];
      });
      Dft<B>::run(t);
      static_for<0, B>([&](auto k2c) {
        constexpr int k2 = decltype(k2c)::value;
        y[n1][k2] = mul_w<n1 * k2, A * B>(t[k2]);
      });
    });
    static_for<0, B>([&](auto k2c) {
      constexpr int k2 = decltype(k2c)::value;
      float2 t[A];
      static_for<0, A>([&](auto n1c) {
        constexpr int n1 = decltype(n1c)::value;
        t[n1] = y[n1][k2];
      });
      Dft<A>::run(t);
      static_for<0, A>([&](auto k1c) {
        constexpr int k1 = decltype(k1c)::value;
        v[B * k1 + k2] = t[k1];
      });
    });
  }
};

// Prime length by the definition (all twiddles compile-time constants); used for 7.
template <int P>
struct DftDirect {
  static __device__ __forceinline__ void run(float2 (&v)[P]) {
    float2 o[P];
    static_for<0, P>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      float2 acc = v[0];
      static_for<1, P>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        acc = cadd(acc, mul_w<n * k, P>(v[n]));
      });
      o[k] = acc;
    });
    static_for<0, P>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      v[k] = o[k];
    });
  }
};
template <> struct Dft<7> : DftDirect<7> {};

template <> struct Dft<6> : DftCT<2, 3> {};
template <> struct Dft<8> : DftCT<2, 4> {};
template <> struct Dft<10> : DftCT<2, 5> {};
template <> struct Dft<12> : DftCT<4, 3> {};
template <> struct Dft<15> : DftCT<3, 5> {};
template <> struct Dft<16> : DftCT<4, 4> {};
template <> struct Dft<20> : DftCT<4, 5> {};
template <> struct Dft<25> : DftCT<5, 5> {};
template <> struct Dft<32> : DftCT<4, 8> {};
template <> struct Dft<9> : DftCT<3, 3> {};
template <> struct Dft<14> : DftCT<2, 7> {};
template <> struct Dft<18> : DftCT<2, 9> {};
template <> struct Dft<21> : DftCT<3, 7> {};
template <> struct Dft<24> : DftCT<4, 6> {};
template <> struct Dft<27> : DftCT<3, 9> {};
template <> struct Dft<28> : DftCT<4, 7> {};
template <> struct Dft<30> : DftCT<5, 6> {};
template <> struct Dft<35> : DftCT<5, 7> {};
template <> struct Dft<36> : DftCT<4, 9> {};

// ----------------------------------------------------------------------------
// Real-input DFTs.  rdft5: 5 real samples -> X[0] (real), X[1], X[2]
// (X[3] = conj X[2], X[4] = conj X[1]).
// ----------------------------------------------------------------------------
__device__ __forceinline__ void rdft5(float a0, float a1, float a2, float a3, float a4, float& X0,
                                      float2& X1, float2& X2) {
  constexpr float c1 = 0.30901699437494742410f;   // cos(2pi/5)
  constexpr float c2 = -0.80901699437494742410f;  // cos(4pi/5)
  constexpr float s1 = 0.95105651629515357212f;   // sin(2pi/5)
  constexpr float s2 = 0.58778525229247312917f;   // sin(4pi/5)
  const float p = a1 + a4, q = a2 + a3, u = a1 - a4, v = a2 - a3;
  X0 = a0 + p + q;
  X1 = {a0 + c1 * p + c2 * q, -(s1 * u + s2 * v)};
  X2 = {a0 + c2 * p + c1 * q, s1 * v - s2 * u};
}

// rdft25: 25 real samples -> bins 0..12 (bin 0 has zero imaginary part).
// n = 5 n1 + n2, k = k1 + 5 k2; only k1 = 0, 1, 2 are computed, the columns
// k1 = 3, 4 being their conjugates: X[9] = conj X[16], X[4] = conj X[21],
// X[8] = conj X[17], X[3] = conj X[22].
__device__ __forceinline__ void rdft25(const float (&x)[25], float2 (&X)[13]) {
  float A0[5];
  float2 B1[5], B2[5];
  static_for<0, 5>([&](auto n2c) {
    constexpr int n2 = decltype(n2c)::value;
    float2 a1, a2;
    rdft5(x[n2], x[5 + n2], x[10 + n2], x[15 + n2], x[20 + n2], A0[n2], a1, a2);
    B1[n2] = mul_w<n2, 25>(a1);
    B2[n2] = mul_w<2 * n2, 25>(a2);
  });
  float r0;
  rdft5(A0[0], A0[1], A0[2], A0[3], A0[4], r0, X[5], X[10]);
  X[0] = {r0, 0.0f};
  Dft<5>::run(B1);
  X[1] = B1[0];
  X[6] = B1[1];
  X[11] = B1[2];
  X[9] = cconj(B1[3]);
  X[4] = cconj(B1[4]);
  Dft<5>::run(B2);
  X[2] = B2[0];
  X[7] = B2[1];
  X[12] = B2[2];
  X[8] = cconj(B2[3]);
  X[3] = cconj(B2[4]);
}

// The same for windowed input: bins 0..12 of the DFT of x[n] * w[n], with the window
// multiply folded into the first butterflies (x1 w1 +/- x4 w4 as one product and two fused
// multiply-adds): 10 instructions fewer than 25 products in front of rdft25.
__device__ __forceinline__ void rdft5w(float x0, float w0, float x1, float w1, float x2, float w2,
                                       float x3, float w3, float x4, float w4, float& X0, float2& X1,
                                       float2& X2) {
  constexpr float c1 = 0.30901699437494742410f;   // cos(2pi/5)
  constexpr float c2 = -0.80901699437494742410f;  // cos(4pi/5)
  constexpr float s1 = 0.95105651629515357212f;   // sin(2pi/5)
  constexpr float s2 = 0.58778525229247312917f;   // sin(4pi/5)
  const float a0 = x0 * w0, t4 = x4 * w4, t3 = x3 * w3;
  const float p = fmaf(x1, w1, t4), u = fmaf(x1, w1, -t4);
  const float q = fmaf(x2, w2, t3), v = fmaf(x2, w2, -t3);
  X0 = a0 + p + q;
  X1 = {a0 + c1 * p + c2 * q, -(s1 * u + s2 * v)};
  X2 = {a0 + c2 * p + c1 * q, s1 * v - s2 * u};
}
__device__ __forceinline__ void rdft25w(const float (&x)[25], const float (&w)[25], float2 (&X)[13]) {
  float A0[5];
  float2 B1[5], B2[5];
  static_for<0, 5>([&](auto n2c) {
    constexpr int n2 = decltype(n2c)::value;
    float2 a1, a2;
    rdft5w(x[n2], w[n2], x[5 + n2], w[5 + n2], x[10 + n2], w[10 + n2], x[15 + n2], w[15 + n2],
           x[20 + n2], w[20 + n2], A0[n2], a1, a2);
    B1[n2] = mul_w<n2, 25>(a1);
    B2[n2] = mul_w<2 * n2, 25>(a2);
  });
  float r0;
  rdft5(A0[0], A0[1], A0[2], A0[3], A0[4], r0, X[5], X[10]);
  X[0] = {r0, 0.0f};
  Dft<5>::run(B1);
  X[1] = B1[0];
  X[6] = B1[1];
  X[11] = B1[2];
  X[9] = cconj(B1[3]);
  X[4] = cconj(B1[4]);
  Dft<5>::run(B2);
  X[2] = B2[0];
  X[7] = B2[1];
  X[12] = B2[2];
  X[8] = cconj(B2[3]);
  X[3] = cconj(B2[4]);
}

}  // namespace mia

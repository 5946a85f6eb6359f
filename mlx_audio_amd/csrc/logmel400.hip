// Fused mel front-end for n_fft = 400, hop = 160 (Whisper's framing):
// pad + frame + window + 400-point real FFT + power + mel band projection +
// log epilogue in one kernel, 64 frames of one clip per 512-thread workgroup.
//
// Mapping.  LANE = frame, WAVE = role inside the FFT.  The 200-point complex
// transform of z[n] = x[2n] + i x[2n+1] is split 8 x 25:
//   pass 1: wave t (0..7) does, for its lane's frame, the radix-25 DFT of
//           z[t + 8r], r = 0..24;
//   pass 2: the 25 radix-8 butterflies over t are dealt to the waves in pairs
//           (k1, 25-k1), whose outputs are the bin pairs (k, 200-k) of the
//           real-FFT split step, so split and |X|^2 run in registers.
// Because the role is wave-uniform every twiddle, window tap and LDS offset
// is a scalar or an instruction immediate, all 64 lanes work, and with odd /
// 2-mod-4 row strides every LDS access is bank-conflict free.
//
// LDS (one 53.8 KB buffer, reused three times per tile, 2 workgroups per CU):
//   span : the tile's 10480 samples, rows of 160 samples at stride 162 dwords
//   ex   : pass-1 -> pass-2 exchange, per frame 8 roles x 13 complex, stride 210
//   P    : power spectrum, bin-major [201][64]
#include "fft_engine.h"
#include "mia_internal.h"

namespace mia {

namespace w400 {
constexpr int N = 400, M = 200, HOP = 160, FT = 64, WG = 512;
constexpr int SPAN = (FT - 1) * HOP + N;  // 10480 samples per tile
constexpr int RS = 162;                   // span row stride (dwords)
constexpr int ES = 210;                   // exchange stride per frame (dwords)
constexpr int LDS_DWORDS = FT * ES;       // 13440 >= 66*162 and >= 201*64
static_assert(LDS_DWORDS >= ((SPAN + HOP - 1) / HOP) * RS, "span fits");
static_assert(LDS_DWORDS >= (M + 1) * FT, "power tile fits");
constexpr int MAXLEN = 16;                // padded filter row length
}  // namespace w400

// Butterflies of pass 2 owned by each wave: a0 (+ partner 25-a0), optional a1
// (+ partner 25-a1, or no partner when a1 == 0).
__constant__ int kW400A0[8] = {1, 3, 5, 7, 9, 10, 11, 12};
__constant__ int kW400A1[8] = {2, 4, 6, 8, 0, -1, -1, -1};

// One split + power step for the bin pair (k, 200-k): U = Z[k], V = Z[200-k].
__device__ __forceinline__ void split_power(float2 U, float2 V, float2 w, float& pk,
                                            float& pmk) {
  const float2 E = {U.x + V.x, U.y - V.y};
  const float2 O = {U.y + V.y, V.x - U.x};
  const float2 WO = cmul(O, w);
  const float ax = E.x + WO.x, ay = E.y + WO.y;
  const float bx = E.x - WO.x, by = E.y - WO.y;
  pk = 0.25f * (ax * ax + ay * ay);
  pmk = 0.25f * (bx * bx + by * by);
}

__device__ __forceinline__ float apply_power(float pw, int mode, float power) {
  if (mode == 1) return sqrtf(pw);
  if (mode == 2) return powf(sqrtf(pw), power);
  return pw;
}

__global__ __launch_bounds__(w400::WG, 4) void logmel400_kernel(FwdParams p, W400Tables tb) {
  using namespace w400;
  __shared__ __attribute__((aligned(16))) float lds[LDS_DWORDS];
  __shared__ float red[2][8];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int t = __builtin_amdgcn_readfirstlane(tid >> 6);  // role
  const long long b = blockIdx.x / p.tiles;
  const int tile = blockIdx.x % p.tiles;
  const long long f0 = (long long)tile * FT;
  const float* __restrict__ yb = p.y + b * p.T;
  const long long s_begin = f0 * HOP - p.pad;

  // ---- stage the tile's samples: global -> LDS, rows of 160 at stride 162 ----
  {
    const bool interior = s_begin >= 0 && s_begin + SPAN <= p.T &&
                          ((reinterpret_cast<uintptr_t>(yb + s_begin) & 15) == 0);
    if (interior) {
      const float4* __restrict__ src = reinterpret_cast<const float4*>(yb + s_begin);
#pragma unroll
      for (int i = 0; i < (SPAN / 4 + WG - 1) / WG; ++i) {
        const int q = tid + i * WG;
        if (q < SPAN / 4) {
          const float4 v = src[q];
          float2* dst = reinterpret_cast<float2*>(lds + RS * (q / 40) + 4 * (q % 40));
          dst[0] = {v.x, v.y};
          dst[1] = {v.z, v.w};
        }
      }
    } else {
      for (int s = tid; s < SPAN; s += WG) {
        lds[RS * (s / HOP) + (s % HOP)] = fetch_sample_any(yb, p.T, s_begin + s, p.pad_mode);
      }
    }
  }
  __syncthreads();

  // ---- pass 1: radix-25 over z[t + 8r] of this lane's frame --------------------
  float2 y[25];
  {
    const float2* __restrict__ sp = reinterpret_cast<const float2*>(lds + RS * lane + 2 * t);
    const float2* __restrict__ win2 = reinterpret_cast<const float2*>(p.win) + t;
#pragma unroll
    for (int r = 0; r < 25; ++r) {
      const float2 x = sp[(16 * r + 2 * (r / 10)) / 2];
      const float2 w = win2[8 * r];
      y[r] = {x.x * w.x, x.y * w.y};
    }
    Dft<25>::run(y);
  }
  __syncthreads();  // all span reads done; the buffer becomes the exchange area

  float2* __restrict__ ex = reinterpret_cast<float2*>(lds);
  const int exw = (ES * lane + 26 * t) / 2;  // this thread's 13 exchange slots
  const int exr = (ES * lane) / 2;           // pass-2 reads: + 13*t' + c
  const int a0k = kW400A0[t];
  const int a1k = kW400A1[t];

  // phase A: outputs k1 = 0..12
#pragma unroll
  for (int c = 0; c < 13; ++c) ex[exw + c] = y[c];
  __syncthreads();
  float2 a0[8], a1[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) a0[q] = ex[exr + 13 * q + a0k];
  if (a1k >= 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) a1[q] = ex[exr + 13 * q + a1k];
  }
  __syncthreads();
  // phase B: outputs k1 = 13..24 at slot k1 - 13
#pragma unroll
  for (int c = 0; c < 12; ++c) ex[exw + c] = y[13 + c];
  __syncthreads();
  float2 b0[8], b1[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) b0[q] = ex[exr + 13 * q + (12 - a0k)];
  if (a1k > 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) b1[q] = ex[exr + 13 * q + (12 - a1k)];
  }
  __syncthreads();  // exchange reads done; the buffer becomes the power tile

  // ---- pass 2 + split + power, written bin-major: P[k][frame] ---------------------
  float* __restrict__ P = lds;
  auto twiddle8 = [&](float2 (&v)[8], int k1) {
    const float2* __restrict__ w = tb.tw2 + 8 * k1;
#pragma unroll
    for (int q = 1; q < 8; ++q) v[q] = cmul(v[q], w[q]);
    Dft<8>::run(v);
  };
  auto do_pair = [&](float2 (&a)[8], float2 (&bb)[8], int k1) {
    twiddle8(a, k1);
    twiddle8(bb, 25 - k1);
    const float2* __restrict__ ws = tb.tws + 8 * k1;
    float* __restrict__ pa = P + FT * k1 + lane;         // bins k1 + 25*k2
    float* __restrict__ pb = P + FT * (25 - k1) + lane;  // bins (25-k1) + 25*k2
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      float pk, pmk;
      split_power(a[k2], bb[7 - k2], ws[k2], pk, pmk);
      pa[FT * 25 * k2] = apply_power(pk, p.power_mode, p.power);
      pb[FT * 25 * (7 - k2)] = apply_power(pmk, p.power_mode, p.power);
    }
  };
  do_pair(a0, b0, a0k);
  if (a1k > 0) {
    do_pair(a1, b1, a1k);
  } else if (a1k == 0) {
    // butterfly 0: bins 25*k2, paired inside the butterfly (k2 <-> 8-k2)
    Dft<8>::run(a1);
    const float2* __restrict__ ws = tb.tws;
    float* __restrict__ pz = P + lane;
    float pk, pmk;
    split_power(a1[0], a1[0], ws[0], pk, pmk);
    pz[0] = apply_power(pk, p.power_mode, p.power);
    pz[FT * 200] = apply_power(pmk, p.power_mode, p.power);
#pragma unroll
    for (int k2 = 1; k2 < 4; ++k2) {
      split_power(a1[k2], a1[8 - k2], ws[k2], pk, pmk);
      pz[FT * 25 * k2] = apply_power(pk, p.power_mode, p.power);
      pz[FT * 25 * (8 - k2)] = apply_power(pmk, p.power_mode, p.power);
    }
    pz[FT * 100] = apply_power(a1[4].x * a1[4].x + a1[4].y * a1[4].y, p.power_mode, p.power);
  }
  __syncthreads();

  // ---- mel band projection + log epilogue: wave t owns mels [m0, m1) ---------------
  const long long f = f0 + lane;
  const bool valid = f < p.F;
  const int m0 = tb.mpart[t], m1 = tb.mpart[t + 1];
  float vmax = -INFINITY, vmin = INFINITY;
  float* __restrict__ orow = p.mout + (b * p.n_mels) * p.F + f;
  for (int m = m0; m < m1; ++m) {
    const int lo = p.row_lo[m], len = p.row_len[m];
    const float* __restrict__ w = tb.wpad + MAXLEN * m;
    const float* __restrict__ pr = P + FT * lo + lane;
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXLEN; ++j) {
      if (j < len) acc = fmaf(w[j], pr[FT * j], acc);
    }
    float outv = acc;
    if (p.epi == MIA_EPI_WHISPER) {
      const float yv = acc <= 1e-10f ? -10.0f : __log10f(acc);
      vmax = fmaxf(vmax, yv);
      outv = (yv + 4.0f) * 0.25f;
      vmin = fminf(vmin, outv);
    } else if (p.epi == MIA_EPI_DB) {
      outv = acc <= p.amin ? p.db_floor : p.coef * __log10f(acc * p.inv_ref);
      vmax = fmaxf(vmax, outv);
      vmin = fminf(vmin, outv);
    }
    if (valid) orow[(long long)m * p.F] = outv;
  }
  if (p.epi != MIA_EPI_NONE && p.clip_max_key != nullptr) {
    if (!valid) {
      vmax = -INFINITY;
      vmin = INFINITY;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      vmax = fmaxf(vmax, __shfl_xor(vmax, off));
      vmin = fminf(vmin, __shfl_xor(vmin, off));
    }
    if (lane == 0) {
      red[0][t] = vmax;
      red[1][t] = vmin;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
      for (int w = 1; w < 8; ++w) {
        vmax = fmaxf(vmax, red[0][w]);
        vmin = fminf(vmin, red[1][w]);
      }
      atomicMax(p.clip_max_key + b, float_to_key(vmax));
      atomicMin(p.clip_min_key + b, float_to_key(vmin));
    }
  }
}

int launch_logmel400(const FwdParams& p0, const W400Tables& tb, hipStream_t s) {
  FwdParams p = p0;
  p.tiles = (int)((p.F + w400::FT - 1) / w400::FT);
  const long long blocks = p.B * p.tiles;
  if (blocks <= 0) return MIA_OK;
  if (blocks > 0x7fffffffLL) {
    set_error("grid too large: %lld workgroups", blocks);
    return MIA_ERR_INVALID;
  }
  hipLaunchKernelGGL(logmel400_kernel, dim3((unsigned)blocks), dim3(w400::WG), 0, s, p, tb);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // namespace mia

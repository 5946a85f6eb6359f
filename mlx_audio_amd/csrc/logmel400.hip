// Fused mel front-end for n_fft = 400, hop = 160 (Whisper's framing):
// pad + frame + window + 400-point real FFT + |X|^2 + mel band projection +
// log epilogue in one persistent kernel; a 512-thread workgroup walks a run of
// 64-frame tiles.
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
// Mel stage.  Each bin k of a triangular filterbank feeds at most two
// filters: the rising edge of filter s(k) and the falling edge of filter
// s(k)-1.  Every wave owns a run of filters and sweeps the bins of those
// filters once: two FMAs per bin, one emit (log, store, running max/min) per
// filter.  The per-bin weights live in two VGPRs (lane i = i-th bin of the
// wave) and are broadcast with v_readlane; the control (when a filter is
// complete) is a 2-bit-per-bin scalar mask.
//
// LDS (one 53.8 KB buffer, reused three times per tile, 2 workgroups per CU):
//   span : the tile's 10480 samples, rows of 160 samples at stride 162 dwords
//   ex   : pass-1 -> pass-2 exchange, per frame 8 roles x 13 complex, stride 210
//   P    : power spectrum, bin-major [201][64]
// The next tile's samples are prefetched into registers while the current
// tile computes.
#include "fft_engine.h"
#include "mia_internal.h"

namespace mia {

namespace w400 {
constexpr int N = 400, M = 200, HOP = 160, FT = 64, WG = 512;
constexpr int SPAN = (FT - 1) * HOP + N;  // 10480 samples per tile
constexpr int SPAN4 = SPAN / 4;           // 2620 float4
constexpr int NPRE = (SPAN4 + WG - 1) / WG;  // 6 float4 per thread
constexpr int RS = 162;                   // span row stride (dwords)
constexpr int ES = 210;                   // exchange stride per frame (dwords)
constexpr int LDS_DWORDS = FT * ES;       // 13440 >= 66*162 and >= 201*64 (+8 rows slack)
constexpr int MEL_MAXC = 13;              // mel sweep: chunks of 8 bins per wave, at most
constexpr int MEL_REC = 32;               // dwords per chunk record: wA[8], wB[8], ctrl, pad
static_assert(LDS_DWORDS >= ((SPAN + HOP - 1) / HOP) * RS, "span fits");
static_assert(LDS_DWORDS >= (M + 1 + 8) * FT, "power tile (+ read slack) fits");
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

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__device__ unsigned long long* g_dbgbuf = nullptr;
#ifdef MIA_W400_STAMPS
#define STAMP(i)                                                                          \
  if ((p.dbg & 8) && g_dbgbuf && blockIdx.x == 77 && (threadIdx.x & 63) == 0 && it < 4)    \
    g_dbgbuf[(it * 8 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_readcyclecounter();
#else
#define STAMP(i)
#endif

template <int EPI>
__global__ __launch_bounds__(w400::WG, 4) void logmel400_kernel(
    FwdParams p, const float* __restrict__ g_y, const float* __restrict__ g_win,
    const float2* __restrict__ g_tw2, const float2* __restrict__ g_tws,
    const int* __restrict__ g_meta, const float* __restrict__ g_sw, float* __restrict__ g_out,
    int* __restrict__ g_maxkey, int* __restrict__ g_minkey) {
  // Every table and the input come in as __restrict__ kernel arguments so that
  // their wave-uniform loads are scalar (s_load) loads; through a struct member
  // the compiler must assume they alias the output and falls back to vector loads.
  using namespace w400;
#ifdef MIA_W400_LDS_PAD
  __shared__ __attribute__((aligned(16))) float lds[LDS_DWORDS + MIA_W400_LDS_PAD];
#else
  __shared__ __attribute__((aligned(16))) float lds[LDS_DWORDS];
#endif

  const int tid0 = threadIdx.x;
  const int lane0 = tid0 & 63;
  const int t = __builtin_amdgcn_readfirstlane(tid0 >> 6);  // role
  const int total = (int)(p.B * p.tiles);  // launcher guarantees < 2^31
  const int first = (int)blockIdx.x * p.tiles_per_wg;
  const int last = min(first + p.tiles_per_wg, total);
  if (first >= last) return;

  // ---- per-wave constants -------------------------------------------------------
  const int a0k = kW400A0[t];
  const int a1k = kW400A1[t];
  const int* __restrict__ meta = g_meta + 16 * t;
  const int klo = meta[0], nchunks = meta[1], m0 = meta[2], npre = meta[3];
  // this wave's sweep program: one 32-dword record per 8-bin chunk
  const float4* __restrict__ prog =
      reinterpret_cast<const float4*>(g_sw) + t * (MEL_MAXC * MEL_REC / 4);

  // ---- span prefetch: the tile's samples -> registers -------------------------------
  int tid = tid0, lane = lane0;
  int cur_b = first / p.tiles;             // clip and tile-in-clip of the current tile,
  int cur_tile = first - cur_b * p.tiles;  // advanced incrementally

  int it = 0;
  for (int idx = first; idx < last; ++idx, ++it) {
    const long long b = cur_b;
    const int tile = cur_tile;
    if (++cur_tile == p.tiles) {
      cur_tile = 0;
      ++cur_b;
    }
    const long long f0 = (long long)tile * FT;
    // Re-derive the lane-dependent addresses every tile instead of keeping a
    // few dozen loop-invariant VGPRs alive across the whole loop body.
    tid = tid0;
    asm volatile("" : "+v"(tid)::"memory");  // also keeps this tile's loads below this point
    lane = tid & 63;
    STAMP(0)

    // ---- stage the tile's samples: global -> LDS, rows of 160 at stride 162 ---------
    {
      const float* __restrict__ yb = g_y + b * p.T;
      const long long s_begin = f0 * HOP - p.pad;
      // tiles wholly inside the clip (no padding, 16-byte aligned) take the
      // vector path; the edge tiles of a clip a plain per-sample loop
      const bool interior = s_begin >= 0 && s_begin + SPAN <= p.T &&
                            ((reinterpret_cast<uintptr_t>(yb + s_begin) & 15) == 0);
      if (p.dbg & 2) {
        for (int s = tid; s < LDS_DWORDS; s += WG) lds[s] = 0.001f * (s & 63);
      } else if (interior) {
        const float4* __restrict__ src = reinterpret_cast<const float4*>(yb + s_begin);
        float4 pre[NPRE];
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
          const int q = tid + i * WG;
          pre[i] = src[q < SPAN4 ? q : SPAN4 - 1];
        }
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
          const int q = tid + i * WG;
          if (q < SPAN4) {
            float2* dst = reinterpret_cast<float2*>(lds + RS * (q / 40) + 4 * (q % 40));
            dst[0] = {pre[i].x, pre[i].y};
            dst[1] = {pre[i].z, pre[i].w};
          }
        }
      } else {
#pragma unroll 1
        for (int s = tid; s < SPAN; s += WG) {
          lds[RS * (s / HOP) + (s % HOP)] = fetch_sample(yb, p.T, s_begin + s, p.pad_mode);
        }
      }
    }
    STAMP(1)
    __syncthreads();
    STAMP(2)

    // ---- pass 1: radix-25 over z[t + 8r] of this lane's frame --------------------------
    float2 y[25];
    {
      const float2* __restrict__ sp = reinterpret_cast<const float2*>(lds + RS * lane + 2 * t);
      const float2* __restrict__ win2 = reinterpret_cast<const float2*>(g_win) + t;
#pragma unroll
      for (int r = 0; r < 25; ++r) {
        const float2 x = sp[(16 * r + 2 * (r / 10)) / 2];
        const float2 w = win2[8 * r];
        y[r] = {x.x * w.x, x.y * w.y};
      }
      Dft<25>::run(y);
    }
    STAMP(3)
    __syncthreads();  // all span reads done; the buffer becomes the exchange area
    STAMP(4)

    float2* __restrict__ ex = reinterpret_cast<float2*>(lds);
    const int exw = (ES * lane + 26 * t) / 2;  // this thread's 13 exchange slots
    const int exr = (ES * lane) / 2;           // pass-2 reads: + 13*t' + c

    // phase A: outputs k1 = 0..12
#pragma unroll
    for (int c = 0; c < 13; ++c) ex[exw + c] = y[c];
    STAMP(5)
    __syncthreads();
    STAMP(6)
    float2 a0[8], a1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a0[q] = ex[exr + 13 * q + a0k];
    if (a1k >= 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) a1[q] = ex[exr + 13 * q + a1k];
    }
    STAMP(7)
    __syncthreads();
    STAMP(8)
    // phase B: outputs k1 = 13..24 at slot k1 - 13
#pragma unroll
    for (int c = 0; c < 12; ++c) ex[exw + c] = y[13 + c];
    STAMP(9)
    __syncthreads();
    STAMP(10)
    // ---- pass 2 + split + power ------------------------------------------------------
    // The partner butterflies arrive with phase B: finish one pair at a time so
    // that only one pair's inputs are live, keep the 16 + 16 power values in
    // registers, and write them once every wave has read its exchange slots.
    auto twiddle8 = [&](float2 (&v)[8], int k1) {
      const float2* __restrict__ w = g_tw2 + 8 * k1;
#pragma unroll
      for (int q = 1; q < 8; ++q) v[q] = cmul(v[q], w[q]);
      Dft<8>::run(v);
    };
    auto do_pair = [&](float2 (&a)[8], int k1, float (&pk)[8], float (&pm)[8]) {
      float2 bb[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) bb[q] = ex[exr + 13 * q + (12 - k1)];
      twiddle8(a, k1);
      twiddle8(bb, 25 - k1);
      const float2* __restrict__ ws = g_tws + 8 * k1;
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) split_power(a[k2], bb[7 - k2], ws[k2], pk[k2], pm[7 - k2]);
    };
    float pk0[8], pm0[8], pk1[8], pm1[8];
    do_pair(a0, a0k, pk0, pm0);
    if (a1k > 0) {
      do_pair(a1, a1k, pk1, pm1);
    } else if (a1k == 0) {
      // butterfly 0: bins 25*k2, paired inside the butterfly (k2 <-> 8-k2);
      // pk1[k2] = bin 25*k2 (k2 = 0..4), pm1[k2] = bin 200 - 25*k2 (k2 = 0..3)
      Dft<8>::run(a1);
      const float2* __restrict__ ws = g_tws;
      split_power(a1[0], a1[0], ws[0], pk1[0], pm1[0]);
#pragma unroll
      for (int k2 = 1; k2 < 4; ++k2) split_power(a1[k2], a1[8 - k2], ws[k2], pk1[k2], pm1[k2]);
      pk1[4] = a1[4].x * a1[4].x + a1[4].y * a1[4].y;
    }
    STAMP(11)
    __syncthreads();  // exchange reads done; the buffer becomes the power tile
    STAMP(12)

    float* __restrict__ P = lds;  // bin-major: P[k][frame]
    P[FT * (M + 1) + tid] = 0.0f;  // rows 201..208: read (with zero weights) by a padded last chunk
    {
      float* __restrict__ pa = P + FT * a0k + lane;         // bins a0k + 25*k2
      float* __restrict__ pb = P + FT * (25 - a0k) + lane;  // bins (25-a0k) + 25*k2
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        pa[FT * 25 * k2] = pk0[k2];
        pb[FT * 25 * k2] = pm0[k2];
      }
    }
    if (a1k > 0) {
      float* __restrict__ pa = P + FT * a1k + lane;
      float* __restrict__ pb = P + FT * (25 - a1k) + lane;
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        pa[FT * 25 * k2] = pk1[k2];
        pb[FT * 25 * k2] = pm1[k2];
      }
    } else if (a1k == 0) {
      float* __restrict__ pz = P + lane;
#pragma unroll
      for (int k2 = 0; k2 < 5; ++k2) pz[FT * 25 * k2] = pk1[k2];
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) pz[FT * (200 - 25 * k2)] = pm1[k2];
    }
    STAMP(13)
    __syncthreads();
    STAMP(14)

    // ---- mel sweep + log epilogue ---------------------------------------------------------
    if (!(p.dbg & 1)) {
      const bool valid = f0 + lane < p.F;
      // wave-uniform row pointer + lane offset: the row step is scalar arithmetic
      float* __restrict__ rowp = g_out + ((b * p.n_mels + m0) * p.F + f0);
      const float* __restrict__ pr = P + FT * klo + lane;
      float acc_cur = 0.0f, acc_next = 0.0f;
      float vmax = -INFINITY, vmin = INFINITY;
      auto emit = [&]() {  // the filter in progress is complete
        float outv = acc_cur;
        if (EPI == MIA_EPI_WHISPER) {
          // (log10(max(x, 1e-10)) + 4) / 4 as one FMA on log2
          outv = fmaf(__builtin_amdgcn_logf(fmaxf(acc_cur, 1e-10f)), 0.25f * 0.30102999566f, 1.0f);
        } else if (EPI == MIA_EPI_DB) {
          // coef * log10(max(x, amin) / ref)
          outv = fmaf(__builtin_amdgcn_logf(fmaxf(acc_cur, p.amin)), p.coef * 0.30102999566f,
                      p.db_floor_shift);
        }
        if (EPI != MIA_EPI_NONE) {
          vmax = fmaxf(vmax, outv);
          vmin = fminf(vmin, outv);
        }
        if (valid) rowp[lane] = outv;
        rowp += p.F;
        acc_cur = acc_next;
        acc_next = 0.0f;
      };
      for (int e = 0; e < npre; ++e) emit();
      if (nchunks > 0) {
        float4 wa0 = prog[0], wa1 = prog[1], wb0 = prog[2], wb1 = prog[3];
        unsigned ctrl = __builtin_bit_cast(unsigned, prog[4].x);
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = pr[FT * j];
        for (int c = 0; c < nchunks; ++c) {
          const float4 ca0 = wa0, ca1 = wa1, cb0 = wb0, cb1 = wb1;
          const unsigned cc = ctrl;
          float cv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) cv[j] = pv[j];
          {  // next chunk's record and power values, in flight during this chunk's FMAs
            const int cn = c + 1 < nchunks ? c + 1 : c;
            const float4* __restrict__ rn = prog + cn * (MEL_REC / 4);
            wa0 = rn[0]; wa1 = rn[1]; wb0 = rn[2]; wb1 = rn[3];
            ctrl = __builtin_bit_cast(unsigned, rn[4].x);
            const float* __restrict__ qn = pr + FT * 8 * cn;
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = qn[FT * j];
          }
          const float wa[8] = {ca0.x, ca0.y, ca0.z, ca0.w, ca1.x, ca1.y, ca1.z, ca1.w};
          const float wb[8] = {cb0.x, cb0.y, cb0.z, cb0.w, cb1.x, cb1.y, cb1.z, cb1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc_cur = fmaf(wa[j], cv[j], acc_cur);
            acc_next = fmaf(wb[j], cv[j], acc_next);
            for (unsigned n = (cc >> (4 * j)) & 15u; n > 0; --n) emit();
          }
        }
      }

      if (EPI != MIA_EPI_NONE && g_maxkey != nullptr) {
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
          // the max key is kept on the log10 scale (before the (.+4)/4 map)
          if (EPI == MIA_EPI_WHISPER) vmax = fmaf(vmax, 4.0f, -4.0f);
          atomicMax(g_maxkey + b, float_to_key(vmax));
          atomicMin(g_minkey + b, float_to_key(vmin));
        }
      }
    }
    STAMP(15)
    __syncthreads();  // power tile reads done before the next tile is staged
  }
}

extern "C" void mia_debug_set_buffer(void* ptr) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbgbuf), &ptr, sizeof(ptr));
}

int launch_logmel400(const FwdParams& p0, const W400Tables& tb, hipStream_t s) {
  FwdParams p = p0;
  p.tiles = (int)((p.F + w400::FT - 1) / w400::FT);
  const long long total = p.B * p.tiles;
  if (total <= 0) return MIA_OK;
  // Persistent grid: 2 workgroups per CU, each walking a contiguous run of
  // tiles (consecutive tiles of a clip share cache lines of the output rows,
  // so one workgroup - one L2 - writes both halves of each line).
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  }
  long long grid = 2LL * cus;
  if (grid > total) grid = total;
  p.tiles_per_wg = (int)((total + grid - 1) / grid);
  grid = (total + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (p.epi == MIA_EPI_WHISPER) {
    hipLaunchKernelGGL(logmel400_kernel<MIA_EPI_WHISPER>, dim3((unsigned)grid), dim3(w400::WG), 0,
                       s, p, p.y, p.win, tb.tw2, tb.tws, tb.sweep_meta, tb.sweep_w, p.mout,
                       p.clip_max_key, p.clip_min_key);
  } else if (p.epi == MIA_EPI_DB) {
    hipLaunchKernelGGL(logmel400_kernel<MIA_EPI_DB>, dim3((unsigned)grid), dim3(w400::WG), 0, s,
                       p, p.y, p.win, tb.tw2, tb.tws, tb.sweep_meta, tb.sweep_w, p.mout,
                       p.clip_max_key, p.clip_min_key);
  } else {
    hipLaunchKernelGGL(logmel400_kernel<MIA_EPI_NONE>, dim3((unsigned)grid), dim3(w400::WG), 0, s,
                       p, p.y, p.win, tb.tw2, tb.tws, tb.sweep_meta, tb.sweep_w, p.mout,
                       p.clip_max_key, p.clip_min_key);
  }
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // namespace mia

// Fused mel front-end for n_fft = 400, hop = 160 (Whisper's framing):
// pad + frame + window + 400-point real FFT + |X|^2 + mel band projection +
// log epilogue in one persistent kernel; a 1024-thread workgroup walks a run of
// 64-frame tiles.
//
// Mapping.  LANE = frame, WAVE = role inside the FFT.  The real transform is
// split 16 x 25 on the real samples themselves:
//   pass 1: wave t (0..15) does, for its lane's frame, the REAL-input radix-25
//           DFT of x[t + 16 r], r = 0..24 (bins 0..12; the rest are conjugates),
//           and applies the step twiddle exp(-2 pi i t k1 / 400);
//   pass 2: wave k1 (0..12) does the radix-16 butterfly over t, which yields
//           X[k1 + 25 k2], k2 = 0..15.  Bins above 200 are the conjugates of
//           bins 400 - k, so the 13 butterflies give every |X[k]|^2, k = 0..200,
//           with no split step.
// Because the role is wave-uniform every twiddle, window tap and LDS offset is
// a scalar or an instruction immediate and all 64 lanes work.  Sixteen light
// waves (<= 64 VGPRs) instead of eight heavy ones put 8 waves on every SIMD.
//
// Mel stage.  Each bin k of a triangular filterbank feeds at most two
// filters: the rising edge of filter s(k) and the falling edge of filter
// s(k)-1.  Every wave owns a run of filters and sweeps their bins once in
// chunks of 8: the chunk's weights and control word arrive by scalar loads
// (two FMAs per bin with an SGPR operand), and a filter is emitted (log,
// store, running max/min) when the control word says it is complete.
//
// LDS (one 57.9 KB buffer, reused three times per tile, 2 workgroups per CU):
//   span : the tile's 10480 samples, rows of 160 samples at stride 161 dwords
//   ex   : pass-1 -> pass-2 exchange in two phases (k1 = 0..6, then 7..12),
//          per frame 16 roles x 7 complex, stride 226 dwords
//   P    : power spectrum, bin-major [201 (+8 zero rows)][64]
#include "fft_engine.h"
#include "mia_internal.h"

namespace mia {

namespace w400 {
constexpr int N = 400, HOP = 160, FT = 64, NW = W400_WAVES, WG = NW * 64;
constexpr int BINS = N / 2 + 1;              // 201
constexpr int SPAN = (FT - 1) * HOP + N;     // 10480 samples per tile
constexpr int SPAN4 = SPAN / 4;              // 2620 float4
constexpr int NPRE = (SPAN4 + WG - 1) / WG;  // 3 float4 per thread
constexpr int RS = 161;                      // span row stride (dwords), odd: conflict-free b32
constexpr int ES = 226;                      // exchange stride per frame (dwords), 2 mod 4
constexpr int LDS_DWORDS = FT * ES;          // 14464
static_assert(NW == 16, "the FFT split is 16 x 25");
static_assert(LDS_DWORDS >= ((SPAN + HOP - 1) / HOP) * RS, "span fits");
static_assert(LDS_DWORDS >= (BINS + 8) * FT, "power tile (+ zero rows) fits");
}  // namespace w400

// Profiling hooks, compiled in only with -DMIA_DEBUG_HOOKS (tools/abl.sh, tools/stamps.py):
// FwdParams::dbg bit 1 skips the mel sweep, bit 4 the output stores, bit 8 enables the
// per-phase cycle stamps of workgroup 77 (needs -DMIA_W400_STAMPS as well).
#ifdef MIA_DEBUG_HOOKS
#define MIA_DBG(p, bit) ((p).dbg & (bit))
__device__ unsigned long long* g_dbgbuf = nullptr;
#else
#define MIA_DBG(p, bit) 0
#endif
#if defined(MIA_DEBUG_HOOKS) && defined(MIA_W400_STAMPS)
#define STAMP(i)                                                                          \
  if ((p.dbg & 8) && g_dbgbuf && blockIdx.x == 77 && (threadIdx.x & 63) == 0 && it < 4)    \
    g_dbgbuf[(it * 16 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_readcyclecounter();
#else
#define STAMP(i)
#endif

// A tile whose samples all lie inside the clip (no padding) and start on a
// 16-byte boundary can be fetched as float4.
__device__ __forceinline__ bool tile_interior(const float* yb, long long s_begin, long long T) {
  return s_begin >= 0 && s_begin + w400::SPAN <= T &&
         ((reinterpret_cast<uintptr_t>(yb + s_begin) & 15) == 0);
}

// Mel sweep of one wave over the power tile.  FULL: every lane's frame exists.
template <int EPI, bool FULL>
__device__ __forceinline__ void mel_sweep(const FwdParams& p, const float* __restrict__ P,
                                          const float4* __restrict__ prog, int klo, int nchunks,
                                          int npre, float* __restrict__ rowp, int lane, bool valid,
                                          float& vmax, float& vmin) {
  using namespace w400;
  const float* __restrict__ pr = P + FT * klo + lane;
  float acc_cur = 0.0f, acc_next = 0.0f;
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
    if ((FULL || valid) && !MIA_DBG(p, 4)) rowp[lane] = outv;
    rowp += p.F;
    acc_cur = acc_next;
    acc_next = 0.0f;
  };
  for (int e = 0; e < npre; ++e) emit();
  if (nchunks <= 0) return;
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
      const float4* __restrict__ rn = prog + cn * (W400_MEL_REC / 4);
      wa0 = rn[0];
      wa1 = rn[1];
      wb0 = rn[2];
      wb1 = rn[3];
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

template <int EPI>
__global__ __launch_bounds__(w400::WG, 8) void logmel400_kernel(
    FwdParams p, const float* __restrict__ g_y, const float* __restrict__ g_win,
    const float2* __restrict__ g_tw, const int* __restrict__ g_meta,
    const float* __restrict__ g_sw, float* __restrict__ g_out, int* __restrict__ g_maxkey,
    int* __restrict__ g_minkey) {
  // Every table and the input come in as __restrict__ kernel arguments so that
  // their wave-uniform loads are scalar (s_load) loads; through a struct member
  // the compiler must assume they alias the output and falls back to vector loads.
  using namespace w400;
  __shared__ __attribute__((aligned(16))) float lds[LDS_DWORDS];

  const int tid0 = threadIdx.x;
  const int t = __builtin_amdgcn_readfirstlane(tid0 >> 6);  // role
  const int total = (int)(p.B * p.tiles);  // launcher guarantees < 2^31
  const int first = (int)blockIdx.x * p.tiles_per_wg;
  const int last = min(first + p.tiles_per_wg, total);
  if (first >= last) return;

  // ---- per-wave constants -------------------------------------------------------
  const int* __restrict__ meta = g_meta + 16 * t;
  const int klo = meta[0], nchunks = meta[1], m0 = meta[2], npre = meta[3];
  // this wave's sweep program: one 32-dword record per 8-bin chunk
  const float4* __restrict__ prog =
      reinterpret_cast<const float4*>(g_sw) + t * (W400_MEL_MAXC * W400_MEL_REC / 4);
  const bool bfly = t < 13;             // this wave owns pass-2 butterfly k1 = t
  const int bslot = t < 7 ? t : t - 7;  // its slot inside its exchange phase

  int tid = tid0, lane = tid0 & 63;
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
    // Re-derive the lane-dependent addresses every tile instead of keeping
    // loop-invariant VGPRs alive across the whole loop body.
    tid = tid0;
    asm volatile("" : "+v"(tid)::"memory");  // also keeps this tile's loads below this point
    lane = tid & 63;
    STAMP(0)

    // ---- stage the tile's samples: global -> LDS, rows of 160 at stride 161 ---------
    {
      const float* __restrict__ yb = g_y + b * p.T;
      const long long s_begin = f0 * HOP - p.pad;
      if (tile_interior(yb, s_begin, p.T)) {
        const float4* __restrict__ src = reinterpret_cast<const float4*>(yb + s_begin);
        float4 now[NPRE];
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
          const int q = tid + i * WG;
          now[i] = src[q < SPAN4 ? q : SPAN4 - 1];
        }
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
          const int q = tid + i * WG;
          if (q < SPAN4) {
            float* dst = lds + RS * (q / 40) + 4 * (q % 40);
            dst[0] = now[i].x;
            dst[1] = now[i].y;
            dst[2] = now[i].z;
            dst[3] = now[i].w;
          }
        }
      } else {
        // edge tiles of a clip (padding, or a clip start that is not 16-byte aligned)
#pragma unroll 1
        for (int s = tid; s < SPAN; s += WG) {
          lds[RS * (s / HOP) + (s % HOP)] = fetch_sample(yb, p.T, s_begin + s, p.pad_mode);
        }
      }
    }
    STAMP(1)
    __syncthreads();
    STAMP(2)

    float* __restrict__ P = lds;  // power tile, bin-major: P[k][frame]
    // ---- pass 1: real radix-25 over x[t + 16 r] of this lane's frame, then the step
    // twiddle exp(-2 pi i t k1 / 400) --------------------------------------------------
    float2 y[13];
    {
      const float* __restrict__ sp = lds + RS * lane + t;
      const float* __restrict__ wt = g_win + 32 * t;  // taps t + 16 r, r = 0..24
      float x[25];
#pragma unroll
      for (int r = 0; r < 25; ++r) x[r] = sp[16 * r + r / 10] * wt[r];
      rdft25(x, y);
      const float2* __restrict__ tw = g_tw + 16 * t;
#pragma unroll
      for (int k = 1; k < 13; ++k) y[k] = cmul(y[k], tw[k]);
      // All 13 outputs are final here: without this the scheduler sinks the second
      // half of the transform below the exchange barriers, where its temporaries
      // collide with the 16 pass-2 inputs.
#pragma unroll
      for (int k = 0; k < 13; ++k) asm volatile("" : "+v"(y[k].x), "+v"(y[k].y));
    }
    STAMP(3)
    __syncthreads();  // all span reads done; the buffer becomes the exchange area
    STAMP(4)

    float2* __restrict__ ex = reinterpret_cast<float2*>(lds);
    const int exw = (ES * lane + 14 * t) / 2;  // this thread's 7 exchange slots
    const int exr = (ES * lane) / 2 + bslot;   // pass-2 reads: + 7 * t'
    float2 v[16];

    // phase A: outputs k1 = 0..6
#pragma unroll
    for (int c = 0; c < 7; ++c) ex[exw + c] = y[c];
    STAMP(5)
    __syncthreads();
    STAMP(6)
    // Waves 0..6 take their butterfly's inputs now.  The others get placeholder
    // values (register copies, overwritten after phase B) so that v is defined on
    // every path and is never carried around the tile loop.
    if (t < 7) {
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = ex[exr + 7 * q];
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = y[q < 13 ? q : 0];
    }
    STAMP(7)
    __syncthreads();
    STAMP(8)
    // phase B: outputs k1 = 7..12 at slot k1 - 7
#pragma unroll
    for (int c = 0; c < 6; ++c) ex[exw + c] = y[7 + c];
    STAMP(9)
    __syncthreads();
    STAMP(10)
    if (t >= 7 && bfly) {
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = ex[exr + 7 * q];
    }
    // ---- pass 2: radix-16 over the roles; v[k2] = X[t + 25 k2] -----------------------
    float pw[16];
    Dft<16>::run(v);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) pw[k2] = fmaf(v[k2].x, v[k2].x, v[k2].y * v[k2].y);
    STAMP(11)
    __syncthreads();  // exchange reads done; the buffer becomes the power tile
    STAMP(12)

    if (tid < 512) P[FT * BINS + tid] = 0.0f;  // rows 201..208: read (zero weight) by a padded chunk
    if (bfly) {
      float* __restrict__ pa = P + FT * t + lane;          // bins t + 25 k2, k2 = 0..7
      float* __restrict__ pb = P + FT * (200 - t) + lane;  // bins 400 - (t + 25 k2), k2 = 8..15
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) pa[FT * 25 * k2] = pw[k2];
      pb[0] = pw[8];
      if (t != 0) {  // butterfly 0 repeats bins 175, 150, ... it already wrote
#pragma unroll
        for (int k2 = 9; k2 < 16; ++k2) pb[-FT * 25 * (k2 - 8)] = pw[k2];
      }
    }
    STAMP(13)
    __syncthreads();
    STAMP(14)

    // ---- mel sweep + log epilogue ---------------------------------------------------------
    if (!MIA_DBG(p, 1)) {
      const bool valid = f0 + lane < p.F;
      // wave-uniform row pointer + lane offset: the row step is scalar arithmetic
      float* __restrict__ rowp = g_out + ((b * p.n_mels + m0) * p.F + f0);
      float vmax = -INFINITY, vmin = INFINITY;
      if (f0 + FT <= p.F) {
        mel_sweep<EPI, true>(p, P, prog, klo, nchunks, npre, rowp, lane, true, vmax, vmin);
      } else {
        mel_sweep<EPI, false>(p, P, prog, klo, nchunks, npre, rowp, lane, valid, vmax, vmin);
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

#ifdef MIA_DEBUG_HOOKS
extern "C" void mia_debug_set_buffer(void* ptr) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbgbuf), &ptr, sizeof(ptr));
}
#endif

int launch_logmel400(const FwdParams& p0, const W400Tables& tb, hipStream_t s) {
  FwdParams p = p0;
  p.tiles = (int)((p.F + w400::FT - 1) / w400::FT);
  const long long total = p.B * p.tiles;
  if (total <= 0) return MIA_OK;
  if (total >= (1LL << 31)) {
    set_error("batch too large for one launch: %lld tiles (split the batch)", total);
    return MIA_ERR_UNSUPPORTED;
  }
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
                       s, p, p.y, tb.win16, tb.tw16, tb.sweep_meta, tb.sweep_w, p.mout,
                       p.clip_max_key, p.clip_min_key);
  } else if (p.epi == MIA_EPI_DB) {
    hipLaunchKernelGGL(logmel400_kernel<MIA_EPI_DB>, dim3((unsigned)grid), dim3(w400::WG), 0, s,
                       p, p.y, tb.win16, tb.tw16, tb.sweep_meta, tb.sweep_w, p.mout,
                       p.clip_max_key, p.clip_min_key);
  } else {
    hipLaunchKernelGGL(logmel400_kernel<MIA_EPI_NONE>, dim3((unsigned)grid), dim3(w400::WG), 0, s,
                       p, p.y, tb.win16, tb.tw16, tb.sweep_meta, tb.sweep_w, p.mout,
                       p.clip_max_key, p.clip_min_key);
  }
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // namespace mia

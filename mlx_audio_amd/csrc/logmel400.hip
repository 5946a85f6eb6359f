// Fused mel front-end for n_fft = 400, hop = 160 (Whisper's framing):
// pad + frame + window + 400-point real FFT + |X|^2 + mel band projection +
// log epilogue in one persistent kernel; a 1024-thread workgroup (one per CU)
// walks a run of 64-frame tiles.
//
// FFT.  The real transform is split 16 x 25 on the real samples themselves:
//   pass 1: role t (0..15) computes the REAL-input radix-25 DFT of x[t + 16 r],
//           r = 0..24 (bins 0..12; the rest are conjugates) and applies the step
//           twiddle exp(-2 pi i t k1 / 400);
//   pass 2: butterfly k1 (0..12) is the radix-16 DFT over the roles and yields
//           X[k1 + 25 k2], k2 = 0..15.  Bins above 200 are the conjugates of bins
//           400 - k, so the 13 butterflies give every |X[k]|^2, k = 0..200, with
//           no split step.
// A wave owns 4 frames: lane = 16 * frame + role.  It loads the 880 samples under
// the next tile's 4 frames as float4 while it transforms the current tile, and after
// the mel sweep passes them through its own LDS region to the lanes that need them.
// The pass-1 -> pass-2 exchange stays inside the wave too (no workgroup barrier).
// Window taps and twiddles are per-lane constants kept in VGPRs.
//
// Mel stage (LANE = frame).  Each bin k of a triangular filterbank feeds at most
// two filters: the rising edge of filter s(k) and the falling edge of filter
// s(k)-1.  Every wave owns a run of filters and sweeps their bins once in chunks
// of 8: the chunk's weights and control word arrive by scalar loads (two FMAs per
// bin with an SGPR operand), and a filter is emitted (log, store, running
// max/min) when the control word says it is complete.
//
// LDS (158 KB) and synchronisation.  Per wave: a power slot of 840 dwords, 832
// dwords of scratch, and a second power slot.  The exchange of tile i lives in
// slot[i & 1] (frames 0, 1) and the scratch (frames 2, 3); before that the same run
// of LDS passes the tile's samples from the lanes that loaded them to the lanes that
// transform them; once the wave has read the exchange back, its 4 columns of the
// power spectrum P[bin][frame & 3] (+ 8 zero rows) replace the slot's contents.
// The mel sweep of tile i reads slot[i & 1] of all 16 waves while waves that are done
// with it already stage and transform tile i + 1 in the other slot, so ONE barrier
// per tile (power of tile i complete = every wave is done with the sweep of tile
// i - 1) is all the workgroup needs.
#include "fft_engine.h"
#include "mia_internal.h"
// the steps of the generated sweep programs, on a context object c (SweepCtx below)
#define SW_LD(x, k) x = c.ld(k);
#define SW_SYNC() __builtin_amdgcn_sched_barrier(0);
#define SW_MUL(a, w, x) a = (w) * x;
#define SW_FMA(a, w, x) a = fmaf((w), x, a);
#define SW_EMIT(j, a) c.emit(std::integral_constant<int, (j)>{}, a);
#define SW_EMIT0(j) c.emit(std::integral_constant<int, (j)>{}, 0.0f);
#define SW_END(n) c.end_of_sweep(std::integral_constant<int, (n)>{});
#include "w400_sweep_gen.inc"

namespace mia {

namespace w400 {
constexpr int N = 400, HOP = 160, FT = 64, NW = W400_WAVES, WG = NW * 64;
constexpr int BINS = N / 2 + 1;   // 201
constexpr int EXF = 16 * 26;      // exchange dwords per frame: 16 roles x 13 complex
constexpr int SL = 840;           // power slot: 2 frames of exchange, then 4 x (201 + 8) power values
constexpr int SC = 2 * EXF;       // scratch: the exchange of frames 2, 3
constexpr int PW = 2 * SL + SC + 20;  // per-wave region, 4 mod 32: conflict-free mel reads
constexpr int WSPAN = 3 * HOP + N;    // 880 samples under a wave's 4 frames
constexpr int NQ4 = 3;                // per lane: 3 float4 (samples 0..767 of the 880) ...
constexpr int NH2 = (WSPAN - NQ4 * 256) / 2;  // ... and one float2 in the first 56 lanes (768..879)
static_assert(NH2 > 0 && NH2 <= 64, "the tail of the wave's samples fits one float2 per lane");
constexpr int RS = 176;               // staged sample rows: 160 samples at stride 176 (16 mod 32), so
                                      // 2 frames x 16 roles of a half-wave hit 32 different banks
static_assert(RS * (WSPAN / HOP) + WSPAN % HOP <= SL + SC, "staged samples fit slot + scratch");
constexpr int LDS_DWORDS = NW * PW;
static_assert(NW == 16, "16 waves x 4 frames");
static_assert(EXF % 64 == 32, "frames 0/1 of a half-wave read opposite bank halves");
static_assert(SL >= 2 * EXF && SL >= 4 * (BINS + 8) && SL % 2 == 0, "slot holds either content");
static_assert(PW % 32 == 4, "mel reads of 16 waves' regions hit 64 different banks");
static_assert(LDS_DWORDS * 4 <= 160 * 1024, "LDS budget");
}  // namespace w400

// Profiling hooks, compiled in only with -DMIA_DEBUG_HOOKS (tools/abl.sh, tools/stamps.py):
// FwdParams::dbg bit 1 skips the mel sweep, bit 2 the FFT arithmetic (LDS traffic kept),
// bit 4 the output stores, bit 16 the global loads, and in the DMA kernel bit 64 the
// exchange through LDS, bit 128 the sample reads, bit 256 the power write; bit 8 enables
// the per-phase cycle stamps of workgroup 77 (needs -DMIA_W400_STAMPS as well).
#ifdef MIA_DEBUG_HOOKS
#define MIA_DBG(p, bit) ((p).dbg & (bit))
__device__ unsigned long long* g_dbgbuf = nullptr;
#else
#define MIA_DBG(p, bit) 0
#endif
#if defined(MIA_DEBUG_HOOKS) && defined(MIA_W400_STAMPS)
#ifndef MIA_W400_STAMP_IT0
#define MIA_W400_STAMP_IT0 0  // first of the four stamped tiles of the workgroup's run
#endif
// Stamps are taken into registers by every wave (one s_memtime each) and written out once
// per tile by workgroup 77 only, so that the stamped kernel runs like the real one.
#define STAMP_DECL unsigned long long stamp_[8];
#define STAMP(i) stamp_[i] = __builtin_readcyclecounter();
#define STAMP_FLUSH                                                                        \
  if ((p.dbg & 8) && blockIdx.x == 77 && (threadIdx.x & 63) == 0 &&                        \
      it >= MIA_W400_STAMP_IT0 && it < MIA_W400_STAMP_IT0 + 4 && g_dbgbuf) {               \
    for (int i_ = 0; i_ < 8; ++i_)                                                         \
      g_dbgbuf[((it - MIA_W400_STAMP_IT0) * 16 + (threadIdx.x >> 6)) * 16 + i_] = stamp_[i_]; \
  }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

// The floor max(x, lo) of the log epilogue, written so that a NaN stays a NaN (as in the
// reference's maximum): one compare + one select.
__device__ __forceinline__ float floor_keep_nan(float x, float lo) { return x < lo ? lo : x; }

// Log epilogue of a floored power value m.
template <int EPI>
__device__ __forceinline__ float log_epilogue(const FwdParams& p, float m) {
  if (EPI == MIA_EPI_WHISPER) {
    // (log10(m) + 4) / 4 as one FMA on log2
    return fmaf(__builtin_amdgcn_logf(m), 0.25f * 0.30102999566f, 1.0f);
  }
  // coef * log10(m / ref)
  return fmaf(__builtin_amdgcn_logf(m), p.coef * 0.30102999566f, p.db_floor_shift);
}

// Mel sweep of one wave over the power tile.  FULL: every lane's frame exists.
template <int EPI, bool FULL>
__device__ __forceinline__ void mel_sweep(const FwdParams& p, const float* __restrict__ P,
                                          const float4* __restrict__ prog, int klo, int nchunks,
                                          int npre, float* __restrict__ rowp, int lane, bool valid,
                                          unsigned& mhi, unsigned& mlo) {
  using namespace w400;
  const float* __restrict__ pr = P + (lane >> 2) * PW + (lane & 3) + 4 * klo;
  float acc_cur = 0.0f, acc_next = 0.0f;
  float ninf = -INFINITY;
  asm volatile("" : "+v"(ninf));  // opaque, or the median below folds back into a canonicalising min
  unsigned voff = 4u * (unsigned)lane;
  const unsigned f4 = 4u * (unsigned)p.F;  // launcher guarantees n_mels * F * 4 < 2^32
  auto emit = [&]() {  // the filter in progress is complete
    float outv = acc_cur;
    if (EPI != MIA_EPI_NONE) {
      // The clip's extremes are tracked on the floored power itself: it is positive, so its
      // bit pattern orders like its value, the log is monotone, and a NaN (either sign) is
      // the largest pattern of all - it wins the maximum, as it does in the reference.
      const float m = floor_keep_nan(acc_cur, EPI == MIA_EPI_WHISPER ? 1e-10f : p.amin);
      outv = log_epilogue<EPI>(p, m);
      if (FULL || valid) {
        mhi = max(mhi, __builtin_bit_cast(unsigned, m));
        mlo = __builtin_bit_cast(unsigned, __builtin_amdgcn_fmed3f(__builtin_bit_cast(float, mlo), m, ninf));
      }
    }
    // wave-uniform row base + a 32-bit lane offset that steps one row per emit
    if ((FULL || valid) && !MIA_DBG(p, 4)) {
      *reinterpret_cast<float*>(reinterpret_cast<char*>(rowp) + voff) = outv;
    }
    voff += f4;
    acc_cur = acc_next;
    acc_next = 0.0f;
  };
  for (int e = 0; e < npre; ++e) emit();
  if (nchunks <= 0) return;
  float4 wa0 = prog[0], wa1 = prog[1], wb0 = prog[2], wb1 = prog[3];
  unsigned ctrl = __builtin_bit_cast(unsigned, prog[4].x);
  float pv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) pv[j] = pr[4 * j];
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
      const float* __restrict__ qn = pr + 4 * 8 * cn;
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = qn[4 * j];
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

// The same sweep for a filterbank known at build time (w400_sweep_gen.inc): the per-wave
// program is straight-line code with literal weights, so there is no per-bin control word,
// no weight load and no loop.  Arithmetic and its order equal mel_sweep's (first
// contribution a product, the rest fused multiply-adds over ascending bins), so the two
// paths give the same bits.
template <int EPI>
struct SweepCtx {
  const FwdParams& p;
  const float* __restrict__ pr;  // this lane's column of the power tile
  char* __restrict__ rowp;       // wave-uniform: first output row of the wave, first frame of the tile
  unsigned voff, f4;             // byte offset of this lane's next store; one row
  unsigned mhi, mlo;             // bit patterns of the largest / smallest floored power
  __device__ __forceinline__ float ld(int k) const { return pr[4 * k]; }
  template <int J>
  __device__ __forceinline__ void emit(std::integral_constant<int, J>, float acc) {
    float outv = acc;
    if (EPI != MIA_EPI_NONE) {
      const float m = floor_keep_nan(acc, EPI == MIA_EPI_WHISPER ? 1e-10f : p.amin);
      outv = log_epilogue<EPI>(p, m);
      // positive floats order like their bit patterns; a NaN is the largest pattern
      mhi = max(mhi, __builtin_bit_cast(unsigned, m));
      mlo = min(mlo, __builtin_bit_cast(unsigned, m));
    }
    if (!MIA_DBG(p, 4)) *reinterpret_cast<float*>(rowp + voff) = outv;
    voff += f4;
  }
  // after the wave's last emit (N = its number of emits)
  template <int N>
  __device__ __forceinline__ void end_of_sweep(std::integral_constant<int, N>) {}
};
template <int BANK, class Ctx>
__device__ __forceinline__ void compiled_sweep(int wv, Ctx& c) {
  if constexpr (BANK == 128) w400_sweep_128(wv, c);
  if constexpr (BANK == 80) w400_sweep_80(wv, c);
}
template <int BANK>
__device__ __forceinline__ int compiled_first_filter(int wv) {
  return BANK == 128 ? kW400Start128[wv] : kW400Start80[wv];
}

// The mel projection on the matrix cores instead of the sweep (BANK = W400_MFMA; kept for
// A/B measurements: fp32 MFMA runs at the vector rate on this chip, and the dense band has
// ~20 x the multiply-adds of the sparse sweep).  mel = A (16 filters x K) * P (K x 16 frames)
// by v_mfma_f32_16x16x4_f32 over the block's band of bins; wave w takes frames
// 16 (w % 4) .. + 15 of the tile and the (at most two) filter blocks of group w / 4.
// Operands per lane l: A[m = l % 16][k = l / 16] from the table (one coalesced 256-byte
// load per step), B[k = l / 16][n = l % 16] from the power tile in LDS; the result's lane l
// holds filters 4 (l / 16) + i, i = 0..3, of frame l % 16.
constexpr int W400_MFMA = -1;
typedef float w400_f4 __attribute__((ext_vector_type(4)));
template <int EPI>
__device__ __forceinline__ void mel_project_mfma(const FwdParams& p, const float* __restrict__ P,
                                                 const int* __restrict__ meta,
                                                 const float* __restrict__ amat, int wv, int lane,
                                                 float* __restrict__ outb, long long f0,
                                                 unsigned& mhi, unsigned& mlo) {
  using namespace w400;
  const int n = 16 * (wv & 3) + (lane & 15);  // this lane's frame of the tile
  const bool valid = f0 + n < p.F;
  const int* __restrict__ mt = meta + 8 * (wv >> 2);
  const float* __restrict__ bp0 = P + (n >> 2) * PW + (n & 3) + 4 * (lane >> 4);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int mb = mt[4 * e], klo = mt[4 * e + 1], steps = mt[4 * e + 2], aoff = mt[4 * e + 3];
    if (mb < 0) continue;
    const float* __restrict__ ap = amat + ((size_t)aoff * 64 + lane);
    const float* __restrict__ bp = bp0 + 4 * klo;
    w400_f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    // Steps go in chunks of 4 (the A table is zero-padded to whole chunks; the power reads of
    // the padding steps repeat the block's last step): a chunk's 16 operands are in flight
    // while the previous chunk multiplies.
    constexpr int CH = 4;
    const int nch = (steps + CH - 1) / CH;
    float a[CH], b[CH];
    auto fetch = [&](int c) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        a[j] = ap[64 * (CH * c + j)];
        b[j] = bp[16 * min(CH * c + j, steps - 1)];
      }
    };
    if (nch > 0) fetch(0);
    for (int c = 0; c < nch; ++c) {
      float ca[CH], cb[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) ca[j] = a[j], cb[j] = b[j];
      if (c + 1 < nch) fetch(c + 1);
#pragma unroll
      for (int j = 0; j < CH; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[j], cb[j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mel = 16 * mb + 4 * (lane >> 4) + i;
      float outv = acc[i];
      if (valid && mel < p.n_mels) {
        if (EPI != MIA_EPI_NONE) {
          const float m = floor_keep_nan(acc[i], EPI == MIA_EPI_WHISPER ? 1e-10f : p.amin);
          outv = log_epilogue<EPI>(p, m);
          mhi = max(mhi, __builtin_bit_cast(unsigned, m));
          mlo = min(mlo, __builtin_bit_cast(unsigned, m));
        }
        if (!MIA_DBG(p, 4)) outb[(long long)mel * p.F + f0 + n] = outv;
      }
    }
  }
}

// Four consecutive samples.  Clips need not start on a 16-byte boundary (odd clip
// lengths): the type only promises float alignment.
struct __attribute__((packed, aligned(4))) float4_a4 {
  float x, y, z, w;
};

// This lane's share of the 880 samples under the wave's 4 frames: 14 registers.
struct __attribute__((packed, aligned(4))) float2_a4 {
  float x, y;
};
struct Staged {
  float4 q[w400::NQ4];
  float2 h;
};
// Fetch it.  Runs that lie wholly inside the clip are one vector load; the few that touch
// a clip boundary go sample by sample through the padding rule.
__device__ __forceinline__ void wave_fetch(Staged& r, const float* __restrict__ yb, long long T,
                                           long long s_wave, int pad_mode, int lane) {
  using namespace w400;
#pragma unroll
  for (int i = 0; i < NQ4; ++i) {
    const long long s0 = s_wave + 4 * (lane + 64 * i);
    if (s0 >= 0 && s0 + 4 <= T) {
      const float4_a4 v = *reinterpret_cast<const float4_a4*>(yb + s0);
      r.q[i] = {v.x, v.y, v.z, v.w};
    } else {
      r.q[i] = {fetch_sample(yb, T, s0, pad_mode), fetch_sample(yb, T, s0 + 1, pad_mode),
                fetch_sample(yb, T, s0 + 2, pad_mode), fetch_sample(yb, T, s0 + 3, pad_mode)};
    }
  }
  const long long s0 = s_wave + NQ4 * 256 + 2 * (lane < NH2 ? lane : NH2 - 1);
  if (s0 >= 0 && s0 + 2 <= T) {
    const float2_a4 v = *reinterpret_cast<const float2_a4*>(yb + s0);
    r.h = {v.x, v.y};
  } else {
    r.h = {fetch_sample(yb, T, s0, pad_mode), fetch_sample(yb, T, s0 + 1, pad_mode)};
  }
}
// Write it to the wave's staging area: sample s at RS * (s / 160) + s % 160.
__device__ __forceinline__ void wave_stage(float* __restrict__ stage, const Staged& r, int lane) {
  using namespace w400;
#pragma unroll
  for (int i = 0; i < NQ4; ++i) {
    const int q = lane + 64 * i;
    *reinterpret_cast<float4*>(stage + RS * (q / 40) + 4 * (q % 40)) = r.q[i];
  }
  if (lane < NH2) {
    const int s = NQ4 * 256 + 2 * lane;
    *reinterpret_cast<float2*>(stage + RS * (s / HOP) + s % HOP) = r.h;
  }
}

template <int EPI, int BANK>
__global__ __launch_bounds__(w400::WG, 4) void logmel400_kernel(
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
  const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);  // wave: frames 4 wv .. 4 wv + 3
  const int total = (int)(p.B * p.tiles);  // launcher guarantees < 2^31
  const int first = (int)blockIdx.x * p.tiles_per_wg;
  const int last = min(first + p.tiles_per_wg, total);
  if (first >= last) return;

  // ---- mel sweep constants of this wave -------------------------------------------
  int klo = 0, nchunks = 0, m0 = 0, npre = 0;
  const float4* __restrict__ prog = nullptr;
  if constexpr (BANK == 0) {
    const int* __restrict__ meta = g_meta + 16 * wv;
    klo = meta[0], nchunks = meta[1], m0 = meta[2], npre = meta[3];
    prog = reinterpret_cast<const float4*>(g_sw) + wv * (W400_MEL_MAXC * W400_MEL_REC / 4);
  } else if constexpr (BANK > 0) {
    m0 = compiled_first_filter<BANK>(wv);
  }

  // ---- FFT constants of this lane's role: window taps t + 16 r and the step
  // twiddles exp(-2 pi i t k / 400), held in VGPRs for the whole run of tiles -------
  const int role0 = tid0 & 15;
  float wnd[25];
  float2 twd[13];
#pragma unroll
  for (int r = 0; r < 25; ++r) wnd[r] = g_win[32 * role0 + r];
#pragma unroll
  for (int k = 1; k < 13; ++k) twd[k] = g_tw[16 * role0 + k];
  // opaque from here on: never re-fetched inside the loop
#pragma unroll
  for (int r = 0; r < 25; ++r) asm volatile("" : "+v"(wnd[r]));
#pragma unroll
  for (int k = 1; k < 13; ++k) asm volatile("" : "+v"(twd[k].x), "+v"(twd[k].y));

  int cur_b = first / p.tiles;             // clip and tile-in-clip of the current tile,
  int cur_tile = first - cur_b * p.tiles;  // advanced incrementally

  // Running extremes of the clip this wave is in; they leave the wave (one atomic
  // pair) only when the clip or the run of tiles ends.
  unsigned mhi = 0u, mlo = 0x7F800000u;  // bit patterns of the largest / smallest floored power
  float* __restrict__ mine = lds + wv * PW;  // this wave's region: slot 0 | scratch | slot 1
  // (slot + scratch is one run of LDS whichever slot a tile uses)

  // ---- the first tile's samples ---------------------------------------------------------
  {
    const float* __restrict__ yb = g_y + (long long)cur_b * p.T;
    const long long s_wave = (long long)cur_tile * (FT * HOP) - p.pad + 4 * wv * HOP;
    Staged regs;
    wave_fetch(regs, yb, p.T, s_wave, p.pad_mode, tid0 & 63);
    wave_stage(mine, regs, tid0 & 63);
  }

  int it = 0;
  for (int idx = first; idx < last; ++idx, ++it) {
    const int b = cur_b;
    const int tile = cur_tile;
    if (++cur_tile == p.tiles) {
      cur_tile = 0;
      ++cur_b;
    }
    const bool more = idx + 1 < last;
    // Re-derive the lane-dependent addresses every tile instead of keeping
    // loop-invariant VGPRs alive across the whole loop body.
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int role = lane & 15;  // pass 1: role t; pass 2: butterfly k1
    const int fi = lane >> 4;    // frame of the wave's four
    STAMP_DECL
    STAMP(0)

    // ---- fetch the next tile's samples; they land during the FFT --------------------------
    Staged regs;
#pragma unroll
    for (int i = 0; i < NQ4; ++i) regs.q[i] = {0.0f, 0.0f, 0.0f, 0.0f};  // defined on every path
    regs.h = {0.0f, 0.0f};
    if (more && !MIA_DBG(p, 16)) {  // (ablation bit 16: no global loads)
      const float* __restrict__ ynb = g_y + (long long)cur_b * p.T;  // cur_* already advanced
      const long long sn = (long long)cur_tile * (FT * HOP) - p.pad + 4 * wv * HOP;
      wave_fetch(regs, ynb, p.T, sn, p.pad_mode, lane);
    }
    STAMP(1)

    float* __restrict__ slot = mine + (it & 1) * (SL + SC);
    {
      const float* __restrict__ stage = mine + (it & 1) * SL;
      // exchange area of this lane's frame: frames 0, 1 in the slot, 2, 3 in the scratch
      float* __restrict__ exf = (fi < 2 ? slot : mine + SL - 2 * EXF) + EXF * fi;
      // The staging and the exchange are private to the wave: LDS operations of one
      // wave complete in order, so only the compiler has to be kept from moving them.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      // ---- pass 1: real radix-25 over x[t + 16 r] of frame 4 wv + fi, then the step twiddle --
      {
        float2 y[13];
        float x[25];
        const float* __restrict__ sp = stage + RS * fi + role;
#pragma unroll
        for (int r = 0; r < 25; ++r) x[r] = sp[16 * r + (RS - HOP) * (r / 10)] * wnd[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (MIA_DBG(p, 2)) {  // ablation: no pass-1 arithmetic
#pragma unroll
          for (int k = 0; k < 13; ++k) y[k] = {x[2 * k], x[(2 * k + 1) % 25]};
        } else {
          rdft25(x, y);
#pragma unroll
          for (int k = 1; k < 13; ++k) y[k] = cmul(y[k], twd[k]);
        }
        float2* __restrict__ exw = reinterpret_cast<float2*>(exf + 26 * role);
#pragma unroll
        for (int c = 0; c < 13; ++c) exw[c] = y[c];
      }
      STAMP(2)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

      // ---- pass 2: radix-16 over the roles; v[k2] = X[k1 + 25 k2] (lanes k1 = 13..15 idle along) --
      float pw[16];
      {
        float2 v[16];
        const int k1r = role < 13 ? role : 12;
        const float2* __restrict__ exr = reinterpret_cast<const float2*>(exf + 2 * k1r);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = exr[13 * q];
        if (!MIA_DBG(p, 2)) Dft<16>::run(v);  // (ablation: no pass-2 arithmetic)
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) pw[k2] = fmaf(v[k2].x, v[k2].x, v[k2].y * v[k2].y);
        if constexpr (EPI != MIA_EPI_WHISPER) {
          if (p.power_mode == 1) {  // power = 1: the filterbank is applied to magnitudes
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) pw[k2] = sqrtf(pw[k2]);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      STAMP(3)

      // ---- power columns of the wave's 4 frames: slot[4 * bin + fi] ---------------------------
      if (lane < 32) slot[4 * BINS + lane] = 0.0f;  // bins 201..208: read (zero weight) by a padded chunk
      if (role < 13) {
        float* __restrict__ pa = slot + 4 * role + fi;          // bins k1 + 25 k2, k2 = 0..7
        float* __restrict__ pb = slot + 4 * (200 - role) + fi;  // bins 400 - (k1 + 25 k2), k2 = 8..15
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) pa[4 * 25 * k2] = pw[k2];
        pb[0] = pw[8];
        if (role != 0) {  // butterfly 0 repeats bins 175, 150, ... it already wrote
#pragma unroll
          for (int k2 = 9; k2 < 16; ++k2) pb[-4 * 25 * (k2 - 8)] = pw[k2];
        }
      }
    }
    // The fetched samples are waited for HERE, a whole FFT after their loads and after
    // the previous sweep's stores; behind the sweep the wait would include the
    // acknowledgements of the stores just issued (the counter retires in order).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    STAMP(4)
    // The power of this tile is complete, and every wave is done with the previous sweep.
    __syncthreads();
    STAMP(5)

    // ---- mel sweep + log epilogue (lane = frame) --------------------------------------------
    if (!MIA_DBG(p, 1)) {
      const long long f0 = (long long)tile * FT;
      const bool valid = f0 + lane < p.F;
      const float* __restrict__ P = lds + (it & 1) * (SL + SC);
      // wave-uniform row pointer + lane offset: the row step is scalar arithmetic
      float* __restrict__ rowp = g_out + (((long long)b * p.n_mels + m0) * p.F + f0);
      if constexpr (BANK == W400_MFMA) {  // g_meta / g_sw carry the projection's tables
        mel_project_mfma<EPI>(p, P, g_meta, g_sw, wv, lane, g_out + (long long)b * p.n_mels * p.F, f0,
                              mhi, mlo);
      } else if (BANK != 0) {
        if (valid) {  // lanes past the clip's last frame sit the sweep out
          SweepCtx<EPI> c{p, P + (lane >> 2) * PW + (lane & 3), reinterpret_cast<char*>(rowp),
                          4u * (unsigned)lane, 4u * (unsigned)p.F, mhi, mlo};
          compiled_sweep<BANK>(wv, c);
          mhi = c.mhi;
          mlo = c.mlo;
        }
      } else if (f0 + FT <= p.F) {
        mel_sweep<EPI, true>(p, P, prog, klo, nchunks, npre, rowp, lane, true, mhi, mlo);
      } else {
        mel_sweep<EPI, false>(p, P, prog, klo, nchunks, npre, rowp, lane, valid, mhi, mlo);
      }
      if (EPI != MIA_EPI_NONE && g_maxkey != nullptr && (cur_tile == 0 || !more)) {
        unsigned rhi = mhi, rlo = mlo;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          rhi = max(rhi, (unsigned)__shfl_xor((int)rhi, off));
          rlo = min(rlo, (unsigned)__shfl_xor((int)rlo, off));  // positive floats order like their bits
        }
        if (lane == 0 && rhi != 0u) {  // (a wave without filters has nothing to report)
          // the stored values of the extreme powers, by the same instructions that stored them
          float rmax = log_epilogue<EPI>(p, __builtin_bit_cast(float, rhi));
          const float rmin = log_epilogue<EPI>(p, __builtin_bit_cast(float, rlo));
          // the max key is kept on the log10 scale (before the (.+4)/4 map)
          if (EPI == MIA_EPI_WHISPER) rmax = fmaf(rmax, 4.0f, -4.0f);
          atomicMax(g_maxkey + b, float_to_key_nan(rmax));
          atomicMin(g_minkey + b, float_to_key(rmin));
        }
        mhi = 0u;
        mlo = 0x7F800000u;
      }
    }
    STAMP(6)

    // ---- the next tile's samples pass through its (now unused) exchange area to reach
    // the lanes that transform them --------------------------------------------------------
    if (more) wave_stage(mine + ((it + 1) & 1) * SL, regs, lane);
    STAMP(7)
    STAMP_FLUSH
  }
}


// =============================================================================================
// The same front-end with the samples brought into LDS by LDS-DMA loads (global_load_lds_dwordx4)
// instead of through registers, for the compiled filterbanks (clip rows at any float
// alignment).  Differences from logmel400_kernel:
//   * a wave's 880 samples have a FIXED staging area of their own at the end of the wave's
//     region (it doubles as the exchange area of frames 2, 3), so the loads of tile i + 1 can be
//     issued as soon as pass 2 of tile i has read its exchange back - before the barrier and
//     before the sweep's stores, which therefore never stand in front of a load in the
//     vector-memory queue; no staging registers, no ds_write pass;
//   * bank conflicts between the two frames of a half-wave are avoided by an XOR swizzle
//     (bit 4 of the position of every sample in an even 160-sample row) that is applied to
//     the SOURCE address of each DMA lane; the reading lanes fold it into their offsets;
//   * the workgroup barrier is a bare s_barrier (a __syncthreads() would drain the loads in
//     flight), and the wait for the DMA sits at the end of the sweep, counted so that the
//     sweep's own stores stay in flight.
//   * the 16 x 25 split is the prime-factor (Good-Thomas) one: 16 and 25 are coprime, so with
//     role n1 taking the samples (25 n1 + 16 n2) mod 400 and butterfly k2 yielding the bins
//     k = k1 (mod 16), k = k2 (mod 25), no twiddle factors stand between the two passes (12
//     complex products and 24 registers per lane less than the Cooley-Tukey split); the
//     window multiply is folded into the first butterflies.  Each lane keeps the LDS addresses
//     of its 25 samples in registers instead (the swizzle and the index wrap cost nothing).
// The mel sweep's arithmetic equals logmel400_kernel's; the transform differs in rounding
// (both are within 2e-6 of the float64 result on the (log10 + 4) / 4 scale).
#ifndef W400D_SWEEP_PRIO
#define W400D_SWEEP_PRIO 1
#endif
#ifndef W400D_STAGGER
#define W400D_STAGGER 0
#endif
namespace w400d {
constexpr int N = 400, HOP = 160, FT = 64, NW = W400_WAVES, WG = NW * 64;
constexpr int BINS = N / 2 + 1;
constexpr int EXF = 16 * 26;       // exchange dwords per frame: 16 roles x 13 complex
constexpr int SL = 2 * EXF;        // power slot (4 x 201 used) = exchange of frames 0, 1
constexpr int WSPAN = 3 * HOP + N; // 880 samples under a wave's 4 frames
constexpr int ST = WSPAN + 4;      // staging area = exchange of frames 2, 3 (the + 4: PW = 4 mod 8)
constexpr int PW = 2 * SL + ST;    // per-wave region: slot 0 | slot 1 | staging
constexpr int OFF_ST = 2 * SL;
constexpr int NDMA = (WSPAN / 4 + 63) / 64;        // 4 wave-wide 16-byte loads
constexpr int LAST_LANES = WSPAN / 4 - 64 * (NDMA - 1);  // lanes of the last one (28)
static_assert(ST >= 2 * EXF && SL >= 4 * BINS, "areas hold either content");
static_assert(PW % 8 == 4, "mel reads of 8 waves' regions hit 32 different banks");
static_assert((PW * 4) % 16 == 0 && (OFF_ST * 4) % 16 == 0, "DMA destinations are 16-byte aligned");
static_assert(NW * PW * 4 <= 160 * 1024, "LDS budget");
static_assert(EXF % 64 == 32, "frames 0/1 of a half-wave read opposite bank halves");
// position of sample s of the wave's span in the staging area, and back (an involution)
__device__ __forceinline__ constexpr int swz(int s) { return s ^ (((s / HOP) & 1) ? 0 : 16); }
// Slot of bin k in a frame's power column: butterfly k2 = k mod 25 (0..12; bins with a larger
// residue are the conjugates of bin 400 - k) leaves output k1 = k mod 16 at 13 k1 + k2.
__device__ __forceinline__ constexpr int bin_slot(int k) {
  int k1 = k % 16, k2 = k % 25;
  if (k2 > 12) {
    k1 = (16 - k1) % 16;
    k2 = 25 - k2;
  }
  return 13 * k1 + k2;
}
}  // namespace w400d

// One wave-wide LDS-DMA load: lane L's 16 bytes at gbase + voff land at lds_byte_addr + 16 L.
// The compiler does not count this load (its s_waitcnt bookkeeping never sees it): the
// caller waits for it with wait_vmcnt<N>().
__device__ __forceinline__ void lds_dma16(const float* gbase, unsigned voff_bytes,
                                          unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff_bytes), "s"(gbase), "s"(lds_byte_addr)
      : "memory");
}
// The same with a per-lane 64-bit source address.
__device__ __forceinline__ void lds_dma16_ptr(const float* gptr, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gptr), "s"(lds_byte_addr)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
// SW_END of the generated sweeps: every older vector-memory operation than the wave's
// last N (the stores of its sweep) is complete - in particular the next tile's DMA.
// Sweep context of the DMA kernel.  Its stores cover whole 128-byte lines: output rows
// start at arbitrary 4-byte phases (a row is F floats), so for row j the wave does not store
// the tile's 64 frames but the 64 frames that start sg_j = (row start + tile start) mod 32
// frames earlier, where that window begins on a line boundary.  The first sg_j of them are
// the last outputs of the previous tile, kept in prev[j]; a lane rotation (ds_bpermute)
// brings every value to the lane that stores it.  The first tile of a clip (or of the
// workgroup's run) has no predecessor and skips those lanes; the last one also stores the
// frames its window leaves over.  Unaligned 256-byte pieces cost 18 % of the step: partial
// lines are written to memory as read-modify-write pairs.
template <int EPI, int MAXR, bool FULL>
struct SweepCtxDma {
  const FwdParams& p;
  const float* __restrict__ pr;  // this lane's column of the power tile
  char* __restrict__ rowp;       // wave-uniform: 32 floats before row 0 of the wave, frame f0
  float (&prev)[MAXR];           // the previous tile's outputs of this lane's frame slot
  unsigned lane4;                // 4 * lane
  unsigned mhi, mlo;             // bit patterns of the largest / smallest floored power
  unsigned sg4;                  // wave-uniform: 4 * frames the next row's window starts early (0..31)
  unsigned fm4;                  // 4 * (F mod 32): sg4's step from row to row
  unsigned roff;                 // wave-uniform: byte offset of the next row from rowp, + 128
  unsigned f4;                   // one row in bytes
  // FULL: the tile has a predecessor in this wave's run, 64 frames of its clip, and is not
  // the last of the clip or the run - every lane stores, nothing is left over.
  int has_prev, flush, nvalid;   // (!FULL) wave-uniform; frames of the clip in this tile
  float pend_y;                  // (FULL) the previous emit's rotated value, stored one emit
  unsigned pend_off;             // later so that the rotation's latency is not waited for
  __device__ __forceinline__ float ld(int k) const { return pr[4 * w400d::bin_slot(k)]; }
  __device__ __forceinline__ void store(unsigned off, float v) const {
    // nontemporal: the rows are not read again by this launch (1 % of the step on the 30 s batch)
    if (!MIA_DBG(p, 4)) __builtin_nontemporal_store(v, reinterpret_cast<float*>(rowp + (size_t)off));
  }
  template <int J>
  __device__ __forceinline__ void emit(std::integral_constant<int, J>, float acc) {
    static_assert(J < MAXR, "row index within the wave");
    // max(acc, 1e-10) that keeps a NaN: acc is a sum of products of non-negative numbers,
    // so it is +0, positive or NaN, and those order like their bit patterns
    const unsigned mb = max(__builtin_bit_cast(unsigned, acc), __builtin_bit_cast(unsigned, 1e-10f));
    const float outv = log_epilogue<EPI>(p, __builtin_bit_cast(float, mb));
    mhi = max(mhi, mb);
    mlo = min(mlo, mb);
    // value for window position (lane + sg) mod 64: this tile's frame, or for the last sg
    // lanes the previous tile's
    const float x = lane4 < 256u - sg4 ? outv : prev[J];
    prev[J] = outv;
    const unsigned t = lane4 - sg4;  // byte offset of this lane's window slot from f0
    const float y = __builtin_bit_cast(
        float, __builtin_amdgcn_ds_bpermute((int)(t & 252u), __builtin_bit_cast(int, x)));
    const unsigned off = t + roff;
    if (FULL) {
      if (J > 0) store(pend_off, pend_y);
      pend_y = y;
      pend_off = off;
    } else {
      const int l = (int)(lane4 >> 2), sg = (int)(sg4 >> 2);
      if ((has_prev || l >= sg) && l - sg < nvalid) store(off, y);
      if (flush) {  // frames 64 - sg .. nvalid - 1 of the tile: this tile's values of lanes 64 - sg + l
        const float y2 = __builtin_bit_cast(
            float, __builtin_amdgcn_ds_bpermute((int)(t & 252u), __builtin_bit_cast(int, outv)));
        if (l < nvalid - 64 + sg) store(off + 256u, y2);
      }
    }
    roff += f4;
    sg4 = (sg4 + fm4) & 124u;
  }
  template <int N>
  __device__ __forceinline__ void end_of_sweep(std::integral_constant<int, N>) {
    // The next tile's DMA is older than this sweep's stores; in a full tile there are N of
    // them, otherwise (masked or doubled stores) wait for everything.
    if (FULL) {
      if (N > 0) store(pend_off, pend_y);
      wait_vmcnt<N>();
    } else {
      wait_vmcnt<0>();
    }
  }
};

// ---- pass 2 across the lanes (P2 = 1 of logmel400d_kernel; kept for A/B measurements) -------
// The 16 roles of a frame are the 16 lanes of one DPP row, so the radix-16 transform over the
// roles can run as four radix-2 decimation-in-frequency stages whose partner operand comes
// through the DPP crossbar instead of an exchange through LDS: lane n and lane n ^ h
// (h = 8, 4, 2, 1) form (own + partner) in the lane with bit h clear and
// (partner - own) * W_{2h}^(n mod h) in the other.  After the four stages lane n holds output
// k1 = bit-reversed n.  Per complex value: 4 instructions per stage + 4 per twiddle layer
// (three layers) = 28, against 10 for the register-resident Dft<16> (160 / 16): the form costs
// more arithmetic than the exchange it removes (see the A/B in the profiles directory).
// lanes n and n ^ 4: no single DPP pattern swaps banks of four lanes, so the lanes with bit 2
// clear take lane n + 4 (row_ror:12, banks 0 and 2) and add, the others take lane n - 4
// (row_ror:4, banks 1 and 3) and subtract.  (s_nop: a DPP read of a register the previous
// vector instruction wrote needs two wait states, which the compiler does not see here.)
__device__ __forceinline__ float2 dpp_bfly4(float2 y) {
  float2 t;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %2 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
      "v_sub_f32_dpp %0, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %1, %3, %3 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
      "v_sub_f32_dpp %1, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xa"
      : "=&v"(t.x), "=&v"(t.y)
      : "v"(y.x), "v"(y.y));
  return t;
}
// partner + sgn * own for the lane pairs of one DPP pattern (sgn = +1 / -1 per lane).  Written
// as v_add_f32_dpp by hand: the compiler leaves a v_mov_b32_dpp in front of every addition.
#define MIA_DPP_BFLY(NAME, PATTERN)                                                        \
  __device__ __forceinline__ float2 NAME(float2 y, float sgn) {                            \
    const float zx = sgn * y.x, zy = sgn * y.y;                                            \
    float2 t;                                                                              \
    asm("s_nop 1\n\t"                                                                      \
        "v_add_f32_dpp %0, %2, %4 " PATTERN " row_mask:0xf bank_mask:0xf\n\t"              \
        "v_add_f32_dpp %1, %3, %5 " PATTERN " row_mask:0xf bank_mask:0xf"                  \
        : "=&v"(t.x), "=&v"(t.y)                                                           \
        : "v"(y.x), "v"(y.y), "v"(zx), "v"(zy));                                           \
    return t;                                                                              \
  }
MIA_DPP_BFLY(dpp_bfly8, "row_ror:8")               // lane ^ 8
MIA_DPP_BFLY(dpp_bfly2, "quad_perm:[2,3,0,1]")     // lane ^ 2
MIA_DPP_BFLY(dpp_bfly1, "quad_perm:[1,0,3,2]")     // lane ^ 1
#undef MIA_DPP_BFLY
struct DppTwiddles {
  float2 w8, w4, w2;  // the lane's stage twiddles (1 in the lanes that add)
  float s8, s2, s1;   // -1 in the lanes that subtract
};
__device__ __forceinline__ DppTwiddles dpp_twiddles(int n) {  // n = role = lane of the row
  DppTwiddles t;
  auto w = [](int k, int N, bool on) -> float2 {  // exp(-2 pi i k / N) or 1
    float s, c;
    sincospif(-2.0f * (float)k / (float)N, &s, &c);
    return on ? float2{c, s} : float2{1.0f, 0.0f};
  };
  t.w8 = w(n & 7, 16, (n & 8) != 0);
  t.w4 = w(n & 3, 8, (n & 4) != 0);
  t.w2 = w(n & 1, 4, (n & 2) != 0);
  t.s8 = (n & 8) ? -1.0f : 1.0f;
  t.s2 = (n & 2) ? -1.0f : 1.0f;
  t.s1 = (n & 1) ? -1.0f : 1.0f;
  return t;
}
// X[bitrev4(lane), .] of the row's 16 values y
__device__ __forceinline__ float2 dpp_dft16(float2 y, const DppTwiddles& t) {
  y = cmul(dpp_bfly8(y, t.s8), t.w8);
  y = cmul(dpp_bfly4(y), t.w4);
  y = cmul(dpp_bfly2(y, t.s2), t.w2);
  return dpp_bfly1(y, t.s1);
}

// Issue the loads of the wave's 880 samples starting at sample s_wave of clip row yb.
__device__ __forceinline__ void dma_wave_samples(const float* __restrict__ yb, long long T,
                                                 long long s_wave, int pad_mode, int lane,
                                                 const unsigned (&soff)[w400d::NDMA],
                                                 float* __restrict__ stage, unsigned stage_bytes) {
  using namespace w400d;
  if (s_wave >= 0 && s_wave + WSPAN <= T) {  // wave-uniform: the whole span lies inside the clip
    const float* __restrict__ gbase = yb + s_wave;
#pragma unroll
    for (int i = 0; i < NDMA - 1; ++i) lds_dma16(gbase, 4u * soff[i], stage_bytes + 1024u * i);
    if (lane < LAST_LANES) lds_dma16(gbase, 4u * soff[NDMA - 1], stage_bytes + 1024u * (NDMA - 1));
  } else {
    // a clip boundary inside the span: lanes whose four samples exist load them, the others
    // build theirs by the padding rule and write them
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int pos = 4 * (lane + 64 * i);
      if (pos < WSPAN) {
        const long long a = s_wave + (long long)soff[i];
        if (a >= 0 && a + 4 <= T) {
          lds_dma16_ptr(yb + a, stage_bytes + 1024u * i);
        } else {
          const float4 v = {fetch_sample(yb, T, a, pad_mode), fetch_sample(yb, T, a + 1, pad_mode),
                            fetch_sample(yb, T, a + 2, pad_mode), fetch_sample(yb, T, a + 3, pad_mode)};
          *reinterpret_cast<float4*>(stage + pos) = v;
        }
      }
    }
  }
}

template <int EPI, int BANK, int P2>
__global__ __launch_bounds__(w400d::WG, 4) void logmel400d_kernel(
    FwdParams p, const float* __restrict__ g_y, const float* __restrict__ g_win,
    float* __restrict__ g_out, int* __restrict__ g_maxkey, int* __restrict__ g_minkey) {
  using namespace w400d;
  static_assert(BANK != 0, "the DMA kernel exists for the compiled filterbanks");
  __shared__ __attribute__((aligned(16))) float lds[NW * PW];

  const int tid0 = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);  // wave: frames 4 wv .. 4 wv + 3
  const int total = (int)(p.B * p.tiles);  // launcher guarantees < 2^31
  const int first = (int)blockIdx.x * p.tiles_per_wg;
  const int last = min(first + p.tiles_per_wg, total);
  if (first >= last) return;
  const int m0 = compiled_first_filter<BANK>(wv);

  // ---- per-lane constants, held in VGPRs for the whole run of tiles ----------------------
  const int role0 = tid0 & 15;
  float wnd[25];       // window taps of this role's samples (25 n1 + 16 n2) mod 400
  unsigned sadr[25];   // ... and their byte offsets in the staging area (swizzle applied)
#pragma unroll
  for (int r = 0; r < 25; ++r) {
    wnd[r] = g_win[32 * role0 + r];
    const int j = (25 * role0 + 16 * r) % N;
    sadr[r] = 4u * (unsigned)swz(HOP * ((tid0 & 63) >> 4) + j);
  }
  unsigned soff[NDMA];  // sample (relative to the wave's span) behind this lane's DMA chunks
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int pos = 4 * ((tid0 & 63) + 64 * i);
    soff[i] = (unsigned)swz(pos < WSPAN ? pos : 0);
  }
#pragma unroll
  for (int r = 0; r < 25; ++r) asm volatile("" : "+v"(wnd[r]), "+v"(sadr[r]));
#pragma unroll
  for (int i = 0; i < NDMA; ++i) asm volatile("" : "+v"(soff[i]));
  DppTwiddles dtw = dpp_twiddles(role0);  // (P2 = 1) stage twiddles of the cross-lane pass 2
  if (P2 == 1) {
    asm volatile("" : "+v"(dtw.w8.x), "+v"(dtw.w8.y), "+v"(dtw.w4.x), "+v"(dtw.w4.y), "+v"(dtw.w2.x),
                 "+v"(dtw.w2.y), "+v"(dtw.s8), "+v"(dtw.s2), "+v"(dtw.s1));
  }

  int cur_b = first / p.tiles;
  int cur_tile = first - cur_b * p.tiles;
  unsigned mhi = 0u, mlo = 0x7F800000u;
  constexpr int MAXR = BANK == 128 ? W400_BANK128_MAXROWS : W400_BANK80_MAXROWS;
  float prev[MAXR];  // the previous tile's outputs (see SweepCtxDma)
#pragma unroll
  for (int j = 0; j < MAXR; ++j) prev[j] = 0.0f;
  STAMP_DECL
  float* __restrict__ mine = lds + wv * PW;  // slot 0 | slot 1 | staging
  float* __restrict__ stage = mine + OFF_ST;
  const unsigned stage_bytes = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned)(unsigned long long)(stage));  // LDS byte address (low half of the flat one)

  // Tiles k = 0 .. n - 1 of this workgroup's run.  Two cursors walk them, each advanced by
  // additions once per period (no per-tile 64-bit multiplies): `sw` is the tile being swept
  // (tile k), `ld` the tile whose samples are being loaded (tile k + 2; k starts at -1).
  const int n_run = last - first;
  struct SweepCursor {
    int b, tile;      // clip, tile of the clip
    int left;         // frames of the clip from this tile's first on (clamped to int)
    char* rp;         // 32 floats before (row m0 of clip b, frame 64 tile)
    unsigned sg4;     // 4 * ((row start of (b, m0)) mod 32)
  } sw;
  struct LoadCursor {
    int tile;
    const float* yb;  // clip row
    long long s;      // first sample of the wave's span in that tile (may be negative: padding)
  } ld;
  const long long clip_out_bytes = 4ll * p.n_mels * p.F;
  const unsigned clip_sg4 = 4u * (unsigned)(((long long)p.n_mels * p.F) & 31);
  const long long s_first = (long long)4 * wv * HOP - p.pad;
  {
    const long long row0 = (long long)cur_b * p.n_mels + m0;
    sw.b = cur_b;
    sw.tile = cur_tile;
    sw.left = (int)min(p.F - (long long)cur_tile * FT, (long long)0x7fffffff);
    sw.rp = reinterpret_cast<char*>(g_out + (row0 * p.F + (long long)cur_tile * FT - 32));
    sw.sg4 = 4u * (unsigned)((row0 * p.F) & 31);
    int lb = cur_b, lt = cur_tile + 1;  // tile 1 of the run
    if (lt == p.tiles) lt = 0, ++lb;
    ld.tile = lt;
    ld.yb = g_y + (long long)lb * p.T;
    ld.s = (long long)lt * (FT * HOP) + s_first;
  }
  auto advance_sweep = [&]() {
    if (++sw.tile == p.tiles) {
      sw.rp += clip_out_bytes - 4ll * FT * (p.tiles - 1);
      sw.sg4 = (sw.sg4 + clip_sg4) & 124u;
      sw.tile = 0;
      ++sw.b;
      sw.left = (int)min(p.F, (long long)0x7fffffff);
    } else {
      sw.rp += 4 * FT;
      sw.left -= FT;
    }
  };
  auto advance_load = [&]() {
    if (++ld.tile == p.tiles) {
      ld.tile = 0;
      ld.yb += p.T;
      ld.s = s_first;
    } else {
      ld.s += FT * HOP;
    }
  };

  // ---- transform of tile kf (its samples are staged), P into slot[kf & 1]; issues the
  // loads of tile kf + 1 = clip bn, tile tn ----------------------------------------------
  auto transform = [&](int kf, bool load_next) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int role = lane & 15;  // pass 1: role t; pass 2: butterfly k1
    const int fi = lane >> 4;    // frame of the wave's four
    float* __restrict__ slot = mine + (kf & 1) * SL;
    // exchange area of this lane's frame: frames 0, 1 in the slot, 2, 3 in the staging area
    float* __restrict__ exf = (fi < 2 ? slot : stage - 2 * EXF) + EXF * fi;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- pass 1: windowed real radix-25 over this role's samples: Y[k2], k2 = 0..12 ---------
    float pw[16];
    {
      float2 y[13];
      float x[25];
      const char* __restrict__ sb = reinterpret_cast<const char*>(stage);
      if (MIA_DBG(p, 128)) {  // ablation: no sample reads
#pragma unroll
        for (int r = 0; r < 25; ++r) x[r] = wnd[r];
      } else {
#pragma unroll
        for (int r = 0; r < 25; ++r) x[r] = *reinterpret_cast<const float*>(sb + sadr[r]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (MIA_DBG(p, 2)) {
#pragma unroll
        for (int k = 0; k < 13; ++k) y[k] = {x[2 * k], x[(2 * k + 1) % 25]};
      } else {
        rdft25w(x, wnd, y);
        // An infinite sample must leave NaN in every bin of its frame, as the cancellations
        // of the reference's transform do (and as the twiddle products of the Cooley-Tukey
        // split did); with additions only, same-signed infinities would survive as +inf.
        // y[0] is the plain sum of the role's windowed samples: it is finite unless one of
        // them is not.  The check is wave-uniform and false for audio.
        const bool bad = !(__builtin_fabsf(y[0].x) < INFINITY);
        if (__builtin_amdgcn_ballot_w64(bad) != 0) {
          if (bad) {
#pragma unroll
            for (int k = 0; k < 13; ++k) y[k] = {__builtin_nanf(""), __builtin_nanf("")};
          }
        }
      }
      if constexpr (P2 == 1) {
        // ---- pass 2 across the lanes: this lane ends up with X[k1 = bitrev(role), k2 = c] ----
#pragma unroll
        for (int c = 0; c < 13; ++c) {
          const float2 v = dpp_dft16(y[c], dtw);
          pw[c] = fmaf(v.x, v.x, v.y * v.y);
        }
      } else {
        float2* __restrict__ exw = reinterpret_cast<float2*>(exf + 26 * role);
        if (!MIA_DBG(p, 64)) {  // (ablation bit 64: no exchange)
#pragma unroll
          for (int c = 0; c < 13; ++c) exw[c] = y[c];
        } else {
#pragma unroll
          for (int c = 0; c < 13; ++c) asm volatile("" ::"v"(y[c].x), "v"(y[c].y));
        }
      }
    }
    STAMP(2)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ---- pass 2: butterfly k2 = role (0..12): radix-16 over the roles n1; v[k1] = X[k1, k2] ----
    if constexpr (P2 == 0) {
      float2 v[16];
      const int k1r = role < 13 ? role : 12;
      const float2* __restrict__ exr = reinterpret_cast<const float2*>(exf + 2 * k1r);
      if (MIA_DBG(p, 64)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = {wnd[q], wnd[q + 1]};
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = exr[13 * q];
      }
      if (!MIA_DBG(p, 2)) Dft<16>::run(v);
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) pw[k1] = fmaf(v[k1].x, v[k1].x, v[k1].y * v[k1].y);
    }
    // opaque: the exchange has been read back (the power values depend on it) before
    // anything below is issued
#pragma unroll
    for (int k2 = 0; k2 < (P2 == 1 ? 13 : 16); ++k2) asm volatile("" : "+v"(pw[k2]));

    STAMP(3)
    // ---- the next tile's samples: the staging area is free from here on ------------------
    if (load_next && !MIA_DBG(p, 16)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      dma_wave_samples(ld.yb, p.T, ld.s, p.pad_mode, lane, soff, stage, stage_bytes);
    }

    STAMP(4)
    // ---- power columns of the wave's 4 frames: X[k1, k2] at slot[4 (13 k1 + k2) + fi] --------
    if constexpr (P2 == 1) {
      const int k1 = (int)(__builtin_bitreverse32((unsigned)role) >> 28);
      float* __restrict__ pa = slot + 4 * 13 * k1 + fi;
#pragma unroll
      for (int c = 0; c < 13; ++c) pa[4 * c] = pw[c];
    } else if (role < 13 && !MIA_DBG(p, 256)) {  // (ablation bit 256: no power write)
      float* __restrict__ pa = slot + 4 * role + fi;
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) pa[4 * 13 * k1] = pw[k1];
    }
  };

  // ---- mel sweep + log epilogue of tile ks = clip b, tile `tile` (lane = frame); ends with
  // the wait that retires the loads in flight ----------------------------------------------
  auto sweep = [&](int ks, bool last_of_run) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int b = sw.b;
    const bool valid = lane < sw.left;
    const bool last_of_clip = sw.tile + 1 == p.tiles;
    const float* __restrict__ P = lds + (ks & 1) * SL;
    if (!MIA_DBG(p, 1)) {
      const int has_prev = sw.tile > 0 && ks > 0;
      const int flush = last_of_clip || last_of_run;
      const int nvalid = sw.left < FT ? sw.left : FT;
      const int full = has_prev && !flush && nvalid == FT;
      const unsigned mhi0 = mhi, mlo0 = mlo;
      const float* __restrict__ pcol = P + (lane >> 2) * PW + (lane & 3);
      char* __restrict__ rp = sw.rp;
      const unsigned sg0 = sw.sg4, fm = 4u * (unsigned)(p.F & 31);
      const unsigned f4 = 4u * (unsigned)p.F;
      unsigned nhi, nlo;
      if (full) {
        SweepCtxDma<EPI, MAXR, true> c{p, pcol, rp, prev, 4u * (unsigned)lane, mhi, mlo, sg0, fm, 128u,
                                       f4, 1, 0, FT, 0.0f, 0u};
        compiled_sweep<BANK>(wv, c);
        nhi = c.mhi;
        nlo = c.mlo;
      } else {
        SweepCtxDma<EPI, MAXR, false> c{p, pcol, rp, prev, 4u * (unsigned)lane, mhi, mlo, sg0, fm, 128u,
                                        f4, has_prev, flush, nvalid, 0.0f, 0u};
        compiled_sweep<BANK>(wv, c);
        nhi = c.mhi;
        nlo = c.mlo;
      }
      // lanes past the clip's last frame swept padding: they do not count
      mhi = valid ? nhi : mhi0;
      mlo = valid ? nlo : mlo0;
    } else {
      wait_vmcnt<0>();  // no sweep, no stores: the DMA is the youngest operation
    }
    if (EPI != MIA_EPI_NONE && g_maxkey != nullptr && (last_of_clip || last_of_run)) {
      unsigned rhi = mhi, rlo = mlo;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        rhi = max(rhi, (unsigned)__shfl_xor((int)rhi, off));
        rlo = min(rlo, (unsigned)__shfl_xor((int)rlo, off));
      }
      if (lane == 0 && rhi != 0u) {
        float rmax = log_epilogue<EPI>(p, __builtin_bit_cast(float, rhi));
        const float rmin = log_epilogue<EPI>(p, __builtin_bit_cast(float, rlo));
        if (EPI == MIA_EPI_WHISPER) rmax = fmaf(rmax, 4.0f, -4.0f);
        atomicMax(g_maxkey + b, float_to_key_nan(rmax));
        atomicMin(g_minkey + b, float_to_key(rmin));
      }
      mhi = 0u;
      mlo = 0x7F800000u;
    }
  };

  // ---- the first tile's samples --------------------------------------------------------------
  dma_wave_samples(g_y + (long long)cur_b * p.T, p.T, (long long)cur_tile * (FT * HOP) + s_first,
                   p.pad_mode, tid0 & 63, soff, stage, stage_bytes);
  wait_vmcnt<0>();

  // ---- period k = -1 .. n - 1: the sweep of tile k (k >= 0) and the transform of tile k + 1
  // (k + 1 < n), then ONE barrier ("P of tile k + 1 is complete, everyone is done sweeping tile
  // k").  The two pieces do not depend on each other; with W400D_STAGGER the waves of a SIMD
  // take them in opposite orders (waves 4..7 and 12..15 transform first). ---------------------
  const bool transform_first = W400D_STAGGER && (wv & 4) != 0;
  for (int k = -1; k < n_run; ++k) {
    const bool has_next = k + 1 < n_run;
    STAMP(0)
    if (transform_first && has_next) transform(k + 1, k + 2 < n_run);
    if (k >= 0) {
      __builtin_amdgcn_s_setprio(W400D_SWEEP_PRIO);
      sweep(k, !has_next);
      __builtin_amdgcn_s_setprio(0);
      advance_sweep();
    }
    STAMP(1)
    if (!transform_first && has_next) transform(k + 1, k + 2 < n_run);
    // Every sweep ends with the wait for the loads in flight; the opening period has no
    // sweep, and a transform-first wave reads tile 1's samples right after it.
    if (k < 0) wait_vmcnt<0>();
    STAMP(5)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    STAMP(6)
    {
      [[maybe_unused]] const int it = k;
      STAMP_FLUSH
    }
    advance_load();
  }
}

#ifdef MIA_DEBUG_HOOKS
extern "C" void mia_debug_set_buffer(void* ptr) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbgbuf), &ptr, sizeof(ptr));
}
#endif

// 80 / 128 if `basis` (n_mels x bins) equals that compiled-in Whisper bank bit for bit, else 0.
int w400_compiled_bank(const float* basis, int n_mels, int bins) {
  if (bins != w400::BINS || (n_mels != 80 && n_mels != 128)) return 0;
  const int nnz = n_mels == 128 ? W400_BANK128_NNZ : W400_BANK80_NNZ;
  const unsigned short(*idx)[2] = n_mels == 128 ? kW400Bank128Idx : kW400Bank80Idx;
  const unsigned* bits = n_mels == 128 ? kW400Bank128Bits : kW400Bank80Bits;
  int seen = 0;
  for (int m = 0; m < n_mels; ++m) {
    for (int k = 0; k < bins; ++k) {
      const unsigned b = __builtin_bit_cast(unsigned, basis[(size_t)m * bins + k]);
      if (seen < nnz && idx[seen][0] == m && idx[seen][1] == k) {
        if (b != bits[seen]) return 0;
        ++seen;
      } else if ((b & 0x7FFFFFFFu) != 0u) {  // a non-zero the compiled bank does not have
        return 0;
      }
    }
  }
  return seen == nnz ? n_mels : 0;
}

template <int EPI, int BANK>
static void launch_w400d(const FwdParams& p, const W400Tables& tb, unsigned grid, hipStream_t s) {
  if (tb.force_dpp_pass2) {
    hipLaunchKernelGGL((logmel400d_kernel<EPI, BANK, 1>), dim3(grid), dim3(w400d::WG), 0, s, p, p.y,
                       tb.winp, p.mout, p.clip_max_key, p.clip_min_key);
  } else {
    hipLaunchKernelGGL((logmel400d_kernel<EPI, BANK, 0>), dim3(grid), dim3(w400d::WG), 0, s, p, p.y,
                       tb.winp, p.mout, p.clip_max_key, p.clip_min_key);
  }
}

template <int EPI, int BANK>
static void launch_w400(const FwdParams& p, const W400Tables& tb, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((logmel400_kernel<EPI, BANK>), dim3(grid), dim3(w400::WG), 0, s, p, p.y,
                     tb.win16, tb.tw16, BANK == W400_MFMA ? tb.mfma_meta : tb.sweep_meta,
                     BANK == W400_MFMA ? tb.mfma_a : tb.sweep_w, p.mout, p.clip_max_key,
                     p.clip_min_key);
}

int launch_logmel400(const FwdParams& p0, const W400Tables& tb, hipStream_t s) {
  FwdParams p = p0;
  p.tiles = (int)((p.F + w400::FT - 1) / w400::FT);
  const long long total = p.B * p.tiles;
  if (total <= 0) return MIA_OK;
  if (total >= (1LL << 31) || p.n_mels * p.F >= (1LL << 30)) {
    set_error("batch too large for one launch: %lld tiles of %lld frames (split the batch)", total, p.F);
    return MIA_ERR_UNSUPPORTED;
  }
  // Persistent grid: one workgroup per CU (158 KB of LDS), each walking a
  // contiguous run of tiles (consecutive tiles of a clip share cache lines of the
  // output rows, so one workgroup - one L2 - writes both halves of each line).
  long long grid = tb.cus > 0 ? tb.cus : 256;  // the plan's device
  if (grid > total) grid = total;
  p.tiles_per_wg = (int)((total + grid - 1) / grid);
  grid = (total + p.tiles_per_wg - 1) / p.tiles_per_wg;
  // The straight-line sweep exists for Whisper's two banks with the Whisper epilogue.
  const int bank = (p.epi == MIA_EPI_WHISPER && !tb.force_table_sweep) ? tb.compiled_bank : 0;
  // LDS-DMA staging: a lane's 16-byte source only has to be float-aligned
  // (global_load_lds_dwordx4 reads rows at any 4-byte phase at the same rate), so odd clip
  // lengths stay on this kernel.
  const bool dma = bank != 0 && !tb.force_vgpr_staging;
  if (tb.force_mfma && tb.mfma_a != nullptr && p.epi == MIA_EPI_WHISPER) {
    launch_w400<MIA_EPI_WHISPER, W400_MFMA>(p, tb, (unsigned)grid, s);
  } else if (dma) {
    if (bank == 128) launch_w400d<MIA_EPI_WHISPER, 128>(p, tb, (unsigned)grid, s);
    else launch_w400d<MIA_EPI_WHISPER, 80>(p, tb, (unsigned)grid, s);
  } else if (p.epi == MIA_EPI_WHISPER) {
    if (bank == 128) launch_w400<MIA_EPI_WHISPER, 128>(p, tb, (unsigned)grid, s);
    else if (bank == 80) launch_w400<MIA_EPI_WHISPER, 80>(p, tb, (unsigned)grid, s);
    else launch_w400<MIA_EPI_WHISPER, 0>(p, tb, (unsigned)grid, s);
  } else if (p.epi == MIA_EPI_DB) {
    launch_w400<MIA_EPI_DB, 0>(p, tb, (unsigned)grid, s);
  } else {
    launch_w400<MIA_EPI_NONE, 0>(p, tb, (unsigned)grid, s);
  }
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // namespace mia

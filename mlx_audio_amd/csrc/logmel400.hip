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

template <int EPI>
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
  const int* __restrict__ meta = g_meta + 16 * wv;
  const int klo = meta[0], nchunks = meta[1], m0 = meta[2], npre = meta[3];
  const float4* __restrict__ prog =
      reinterpret_cast<const float4*>(g_sw) + wv * (W400_MEL_MAXC * W400_MEL_REC / 4);

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
    STAMP(0)

    // ---- fetch the next tile's samples; they land during the FFT --------------------------
    Staged regs;
#pragma unroll
    for (int i = 0; i < NQ4; ++i) regs.q[i] = {0.0f, 0.0f, 0.0f, 0.0f};  // defined on every path
    regs.h = {0.0f, 0.0f};
    if (more) {
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
        rdft25(x, y);
#pragma unroll
        for (int k = 1; k < 13; ++k) y[k] = cmul(y[k], twd[k]);
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
        Dft<16>::run(v);
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) pw[k2] = fmaf(v[k2].x, v[k2].x, v[k2].y * v[k2].y);
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
      if (f0 + FT <= p.F) {
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
  if (total >= (1LL << 31) || p.n_mels * p.F >= (1LL << 30)) {
    set_error("batch too large for one launch: %lld tiles of %lld frames (split the batch)", total, p.F);
    return MIA_ERR_UNSUPPORTED;
  }
  // Persistent grid: one workgroup per CU (158 KB of LDS), each walking a
  // contiguous run of tiles (consecutive tiles of a clip share cache lines of the
  // output rows, so one workgroup - one L2 - writes both halves of each line).
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  }
  long long grid = 1LL * cus;
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

// Forward mel path for n_fft = 1024 with the transform's exchange kept inside a wave: fused
// pad + frame + window + real FFT + power + mel band projection + log epilogue, frame-major
// output (B, F, n_mels).
//
// The half-size complex transform (M = n_fft / 2 = 512 points of z[n] = x[2n] + i x[2n+1]) is
// split M = 32 x 16 over the 16 lanes that own a frame (a wave holds 4 consecutive frames):
//   staging the samples under the wave's 4 frames (3 hops + n_fft) come from global memory
//           straight into the wave's LDS staging area by LDS-DMA loads (1 KiB per wave
//           instruction; rows at any float alignment), issued as soon as pass A of the previous
//           step has read its samples, so that they land behind a whole step of arithmetic;
//   pass A  lane n1 (0..15) takes z[n1 + 16 n2], n2 = 0..31, times the window, runs a radix-32
//           transform in registers and applies the step twiddle W_M^(n1 k2);
//   exchange through the wave's own LDS region, [k2][n1] with an odd row pitch (no conflicts on
//           either side), in two halves of 16 columns: every lane owns one column of each half;
//   pass B  lane c takes the columns k2 = c and 32 - c (lane 0: 0 and 16) and runs a radix-16
//           transform over n1 on each: Z[k2 + 32 k1], k1 = 0..15.  A column and its mirror hold
//           the bin pairs (k, M - k) the real-input split step combines, so the split runs in
//           registers.
// The power spectrum of the wave's frames then replaces the exchange data; the wave projects it
// on the mel bands (lane = (band, frame), the 16 bands of one step adjacent so that their
// lengths are alike), applies the epilogue and stores 64-byte pieces of four consecutive
// (frame, band) rows - consecutive steps complete the lines in L2.  LDS operations of one wave
// complete in order, so nothing in the loop synchronises workgroup-wide: the eight waves of a
// workgroup share only the read-only tables.
//
// Against the four-pass kernel of spectral_fwd.hip (a frame spread over a whole wave, eight
// points per lane and pass): a quarter of the LDS stores, no workgroup barriers, 10 wide loads
// per 4 frames instead of 32, radix-32 / radix-16 butterflies with compile-time inner twiddles
// instead of four layers of table twiddles.
#include "fft_engine.h"
#include "mia_internal.h"

#ifndef LANE_ABL
#define LANE_ABL 0   // ablation bits for timing runs (results are garbage): 1 no projection, 2 no global loads
#endif
namespace mia {
namespace lanefft {

constexpr int L = 16;     // lanes per frame
constexpr int FPW = 4;    // frames per wave and step
constexpr int RA = 32;    // pass A radix (points per lane)
constexpr int RB = 16;    // pass B radix
constexpr int M = RA * RB, N = 2 * M;
constexpr int PITCH = L + 1;                 // complex elements per k2 row of the exchange
constexpr int FRAME_C = (RA / 2) * PITCH;    // per frame: 16 columns, 272 complex = 2176 bytes (32 banks
                                             // mod 64: the two frames of a half-wave take complementary banks)
constexpr int FRAME_F = 2 * FRAME_C;         // the same in floats: the frame's power row lives here later
constexpr int WAVES = 8, WG = 64 * WAVES;
constexpr int NPAIR = 17;                    // split pairs per lane (the 17th: lane 0's self-paired bin M/2)
constexpr int P_OFF = 8;                     // a frame's power row starts 8 * frame floats into its region
static_assert(FRAME_F % 64 == 32, "frame regions alternate bank halves");
static_assert(M + 1 + (FPW - 1) * P_OFF <= FRAME_F, "the power row fits the frame's exchange region");

// bin of split pair p of lane c (its partner is M - bin)
__host__ __device__ constexpr int pair_bin(int p, int c) {
  if (c != 0) return c + 32 * p;
  if (p < 8) return 32 * p;
  if (p < 16) return 16 + 32 * (p - 8);
  return M / 2;
}

// floats of the staging area: the samples under 4 consecutive frames, in whole 16-byte pieces
__host__ __device__ inline int span_floats(int hop) { return ((FPW - 1) * hop + N + 3) & ~3; }
__host__ inline size_t lds_bytes(int hop, int nnz, int n_mels) {
  return sizeof(float2) * (size_t)(M + RA * L + NPAIR * L + WAVES * FPW * FRAME_C) +
         sizeof(float) * (size_t)WAVES * span_floats(hop) + sizeof(float) * (size_t)nnz +
         3 * sizeof(int) * (size_t)n_mels;
}

// One wave-wide LDS-DMA load: lane l's 16 bytes at gptr land at lds_byte_addr + 16 l.  The
// compiler does not count it; the kernel waits with an explicit s_waitcnt vmcnt.
__device__ __forceinline__ void dma16(const float* gptr, unsigned lds_byte_addr) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(gptr), "s"(lds_byte_addr)
      : "memory");
}

__global__ __launch_bounds__(WG, 2) void fwd_lane_kernel(FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lane_smem[];
  const int span = span_floats(p.hop);
  float2* const s_win = reinterpret_cast<float2*>(lane_smem);  // [M] window taps of the sample pairs
  float2* const s_twA = s_win + M;                             // [RA][L] W_M^(n1 k2)
  float2* const s_tws = s_twA + RA * L;                        // [NPAIR][L] split twiddles W_N^bin
  float2* const s_ex = s_tws + NPAIR * L;                      // [WAVES][FPW][FRAME_C]
  float* const s_st = reinterpret_cast<float*>(s_ex + WAVES * FPW * FRAME_C);  // [WAVES][span]
  float* const s_wts = s_st + WAVES * span;
  int* const s_rows = reinterpret_cast<int*>(s_wts + p.nnz);   // row_lo | row_len | row_off

  const int tid = threadIdx.x;
  {
    const float2* __restrict__ win2 = reinterpret_cast<const float2*>(p.win);
    for (int i = tid; i < M; i += WG) s_win[i] = win2[i];
    for (int i = tid; i < RA * L; i += WG) s_twA[i] = p.tw[2 * (i % L) * (i / L)];
    for (int i = tid; i < NPAIR * L; i += WG) s_tws[i] = p.tw[pair_bin(i / L, i % L)];
    for (int i = tid; i < p.nnz; i += WG) s_wts[i] = p.wts[i];
    for (int i = tid; i < p.n_mels; i += WG) {
      s_rows[i] = p.row_lo[i];
      s_rows[p.n_mels + i] = p.row_len[i];
      s_rows[2 * p.n_mels + i] = p.row_off[i];
    }
  }
  __syncthreads();

  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane >> 4;   // frame of the wave's four
  const int n1 = lane & 15;   // pass A: residue of the lane's points; pass B / split: column c
  const bool t0 = n1 == 0;
  float2* const exw = s_ex + wv * FPW * FRAME_C;   // the wave's exchange region
  float2* const exf = exw + fi * FRAME_C;          // this lane's frame
  float* const regw = reinterpret_cast<float*>(exw);
  float* const stw = s_st + wv * span;             // the wave's staged samples
  const unsigned stw_bytes = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned)(unsigned long long)(stw));    // LDS byte address (low half of the flat one)

  // Steps (4 consecutive frames of one clip) of this wave: a contiguous run.
  const unsigned spc = (unsigned)((p.F + FPW - 1) / FPW);       // steps per clip
  const long long total = p.B * (long long)spc;                  // launcher keeps it below 2^31
  const long long nwaves = (long long)gridDim.x * WAVES;
  const long long per = (total + nwaves - 1) / nwaves;
  const long long first = ((long long)blockIdx.x * WAVES + wv) * per;
  const long long last = first + per < total ? first + per : total;
  if (first >= last) return;

  // Exchanges stay inside the wave, and the LDS serves one wave's operations in issue order:
  // a later operation of the wave sees every earlier one, whatever lane issued it.  Only the
  // compiler has to be kept from moving LDS accesses across an exchange point - no s_waitcnt,
  // no barrier (a workgroup-scope fence would drain the LDS queue a dozen times per step).
  auto wave_sync = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  };

  // ---- the samples of step q into the staging area.  A step whose span lies inside the clip is
  // ten wave-wide LDS-DMA loads; one that touches the clip's ends (padding, frames past the
  // last) is built sample by sample by the padding rule (a rolled loop: it is rare) ------------
  // (part, nparts): the DMA pieces go out in nparts groups, spread over the step's phases - a
  // wave blocks on each LDS-DMA issue for a few hundred clocks when they go back to back
  auto stage_step = [&](long long q, int part = 0, int nparts = 1) {
    if (q >= last) return;
    const unsigned b = (unsigned)q / spc;
    const long long f0 = (long long)((unsigned)q - b * spc) * FPW;
    const float* __restrict__ yb = p.y + (long long)b * p.T;
    const long long s_first = f0 * p.hop - p.pad;
    if (LANE_ABL & 2) return;
    if (s_first >= 0 && s_first + span <= p.T && f0 + FPW <= p.F) {   // wave-uniform
      const float* __restrict__ src = yb + s_first + 4 * lane;
      const int pieces = span / 4;
      for (int i = part; 64 * i < pieces; i += nparts) {
        if (64 * i + lane < pieces) dma16(src + 256 * i, stw_bytes + 1024u * i);
      }
    } else if (part == 0) {
#pragma unroll 1
      for (int i = lane; i < span; i += 64) {
        stw[i] = fetch_sample(yb, p.T, s_first + i, p.pad_mode);   // (frames past the last are not stored)
      }
    }
  };
  stage_step(first);

  int kmax = kKeyMinusInf;   // running extremes of the clip this wave is in
  float vmin = INFINITY;
  auto flush_keys = [&](unsigned b) {
    if (p.epi == MIA_EPI_NONE || p.clip_max_key == nullptr) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      kmax = max(kmax, __shfl_xor(kmax, off));
      vmin = fminf(vmin, __shfl_xor(vmin, off));
    }
    if (lane == 0 && kmax != kKeyMinusInf) {
      atomicMax(p.clip_max_key + b, kmax);
      atomicMin(p.clip_min_key + b, float_to_key(vmin));
    }
    kmax = kKeyMinusInf;
    vmin = INFINITY;
  };

  // The bands a lane projects are the same in every step: band (lane / 4) + 16 it in turn it.
  // Their table rows stay in registers for the first RIT turns (64 bands).
  constexpr int RIT = 4;
  int r_lo[RIT], r_len[RIT], r_off[RIT];
#pragma unroll
  for (int it = 0; it < RIT; ++it) {
    const int m = (lane >> 2) + 16 * it;
    const bool have = m < p.n_mels;
    r_lo[it] = have ? s_rows[m] : 0;
    r_len[it] = have ? s_rows[p.n_mels + m] : 0;
    r_off[it] = have ? s_rows[2 * p.n_mels + m] : 0;
  }

  unsigned cur_b = (unsigned)first / spc;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first step's samples
  for (long long q = first; q < last; ++q) {
    const unsigned b = (unsigned)q / spc;
    const long long f0 = (long long)((unsigned)q - b * spc) * FPW;
    if (b != cur_b) {
      flush_keys(cur_b);
      cur_b = b;
    }
    const int nf = p.F - f0 < FPW ? (int)(p.F - f0) : FPW;   // frames of the clip in this step
    wave_sync();

    // ---- pass A: window, radix 32 over n2, step twiddle -----------------------------------------
    float2 z[RA];
    {
      const float2* __restrict__ sp = reinterpret_cast<const float2*>(stw + fi * p.hop) + n1;
#pragma unroll
      for (int n2 = 0; n2 < RA; ++n2) {
        const float2 x = sp[L * n2];
        const float2 w = s_win[n1 + L * n2];
        z[n2] = {x.x * w.x, x.y * w.y};
      }
    }
    // the samples have been read (values in registers) before the loads that overwrite them go out:
    // the DMA writes LDS from the memory side, outside the wave's LDS queue
#pragma unroll
    for (int n2 = 0; n2 < RA; ++n2) asm volatile("" : "+v"(z[n2].x), "+v"(z[n2].y));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stage_step(q + 1, 0, 4);   // lands behind the rest of this step
    {
      float2 twa[RA];      // in flight during the transform
#pragma unroll
      for (int k2 = 1; k2 < RA; ++k2) twa[k2] = s_twA[k2 * L + n1];
      Dft<RA>::run(z);
#pragma unroll
      for (int k2 = 1; k2 < RA; ++k2) z[k2] = cmul(z[k2], twa[k2]);
    }

    // ---- exchange in two halves + pass B: radix 16 over n1 on the columns c and 32 - c
    // (lane 0: 0 and 16) -------------------------------------------------------------------------
    float2 A[RB], Bv[RB];
#pragma unroll
    for (int k2 = 0; k2 < RA / 2; ++k2) exf[k2 * PITCH + n1] = z[k2];
    wave_sync();
    stage_step(q + 1, 1, 4);
#pragma unroll
    for (int j = 0; j < RB; ++j) A[j] = exf[n1 * PITCH + j];
    wave_sync();
#pragma unroll
    for (int k2 = RA / 2; k2 < RA; ++k2) exf[(k2 - RA / 2) * PITCH + n1] = z[k2];
    wave_sync();
    {
      const int cb = t0 ? 0 : 16 - n1;   // column 32 - c (lane 0: 16) of the second half
#pragma unroll
      for (int j = 0; j < RB; ++j) Bv[j] = exf[cb * PITCH + j];
    }
    float2 tws[16];
#pragma unroll
    for (int pp = 0; pp < 16; ++pp) tws[pp] = s_tws[pp * L + n1];
    stage_step(q + 1, 2, 4);
    Dft<RB>::run(A);
    stage_step(q + 1, 3, 4);
    Dft<RB>::run(Bv);
    wave_sync();  // the exchange has been read: the power rows may overwrite it

    // ---- split step in registers: pair p gives bins k and M - k; powers to LDS ------------------
    float* const prow = reinterpret_cast<float*>(exf) + P_OFF * fi;
    {
      auto emit = [&](int k, float2 X) { prow[k] = 0.25f * (X.x * X.x + X.y * X.y); };
      static_for<0, 16>([&](auto pc) {
        constexpr int pp = decltype(pc)::value;
        // lane 0: pairs inside column 0 (0,0), (1,15) .. (7,9), then inside column 16 (j, 15 - j)
        constexpr int u0 = pp < 8 ? pp : pp - 8;
        constexpr int v0 = pp < 8 ? (16 - pp) % 16 : 15 - (pp - 8);
        const float2 U0 = pp < 8 ? A[u0] : Bv[u0];
        const float2 V0 = pp < 8 ? A[v0] : Bv[v0];
        const float2 U = t0 ? U0 : A[pp];
        const float2 V = t0 ? V0 : Bv[15 - pp];
        const int k = t0 ? pair_bin(pp, 0) : n1 + 32 * pp;
        const float2 w = tws[pp];
        // E = U + conj(V);  O = -i (U - conj(V));  X[k] = (E + w O) / 2,  X[M-k] = conj(E - w O) / 2
        const float2 E = {U.x + V.x, U.y - V.y};
        const float2 O = {U.y + V.y, V.x - U.x};
        const float2 WO = cmul(O, w);
        emit(k, {E.x + WO.x, E.y + WO.y});
        emit(M - k, {E.x - WO.x, WO.y - E.y});
      });
      if (t0) {  // the self-paired bin M/2: X = conj(Z[M/2]) (times 2, as the pairs above)
        const float2 Z = A[8];
        emit(M / 2, {2.0f * Z.x, -2.0f * Z.y});
      }
    }
    wave_sync();
    if (p.power_mode != 0) {  // magnitudes or |X|^p instead of powers: one pass over the rows
      for (int i = lane; i < FPW * (M + 1); i += 64) {
        const int r = i / (M + 1), k = i - r * (M + 1);
        float* __restrict__ v = regw + r * FRAME_F + P_OFF * r + k;
        const float mag = sqrtf(*v);
        *v = p.power_mode == 1 ? mag : powf(mag, p.power);
      }
      wave_sync();
    }

    // The next step's samples have had pass B and the split to land; waiting here, in front of
    // this step's stores, keeps the stores' acknowledgements out of the wait (the counter
    // retires in order).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- mel projection + epilogue: lane = (band, frame), frame fastest --------------------------
    {
      const int nm = p.n_mels;
      const int fl = lane & 3;
      const float* __restrict__ pr = regw + fl * FRAME_F + P_OFF * fl;
      float* __restrict__ orow = p.mout + ((long long)b * p.F + f0 + fl) * nm;
      auto band = [&](int lo, int len, int off) -> float {
        const float* __restrict__ w = s_wts + off;
        const float* __restrict__ x = pr + lo;
        float acc = 0.0f;
        int j = 0;
        for (; j + 8 <= len; j += 8) {   // eight taps' operands in flight together; same order of the sum
          float wv[8], xv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) wv[u] = w[j + u], xv[u] = x[j + u];
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = fmaf(wv[u], xv[u], acc);
        }
        if (j + 4 <= len) {
          float wv[4], xv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) wv[u] = w[j + u], xv[u] = x[j + u];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc = fmaf(wv[u], xv[u], acc);
          j += 4;
        }
        for (; j < len; ++j) acc = fmaf(w[j], x[j], acc);
        float outv = acc;
        if (p.epi == MIA_EPI_WHISPER) {
          const float yv = acc <= 1e-10f ? -10.0f : __log10f(acc);
          outv = (yv + 4.0f) * 0.25f;
          if (fl < nf) {
            kmax = max(kmax, float_to_key_nan(yv));
            vmin = fminf(vmin, outv);
          }
        } else if (p.epi == MIA_EPI_DB) {
          outv = acc <= p.amin ? p.db_floor : p.coef * __log10f(acc * p.inv_ref);
          if (fl < nf) {
            kmax = max(kmax, float_to_key_nan(outv));
            vmin = fminf(vmin, outv);
          }
        }
        return outv;
      };
      if (!(LANE_ABL & 1)) {
        // The first RIT turns' values wait in registers until every turn has read the power
        // rows; they are then staged over those rows and leave as whole 16-byte pieces of
        // consecutive (frame, band) rows.  Later turns (more than 16 RIT bands) store directly.
        float val[RIT];
#pragma unroll
        for (int it = 0; it < RIT; ++it) {
          val[it] = 16 * it < nm ? band(r_lo[it], r_len[it], r_off[it]) : 0.0f;
        }
        for (int m = (lane >> 2) + 16 * RIT; m < nm; m += 16) {
          const float v = band(s_rows[m], s_rows[nm + m], s_rows[2 * nm + m]);
          if (fl < nf) orow[m] = v;
        }
        wave_sync();
        const int nst = nm < 16 * RIT ? nm : 16 * RIT;   // staged bands per frame
        float* __restrict__ stage = regw + fl * FRAME_F;
#pragma unroll
        for (int it = 0; it < RIT; ++it) {
          if (16 * it < nm) stage[(lane >> 2) + 16 * it] = val[it];
        }
        wave_sync();
        const int per4 = nst / 4, n4 = nf * per4;
        for (int i = lane; i < n4; i += 64) {
          const int r = i / per4, c4 = i - r * per4;
          *reinterpret_cast<float4*>(p.mout + ((long long)b * p.F + f0 + r) * nm + 4 * c4) =
              *reinterpret_cast<const float4*>(regw + r * FRAME_F + 4 * c4);
        }
        wave_sync();   // the next step's exchange overwrites the staged rows
      }
    }
  }
  flush_keys(cur_b);
}

}  // namespace lanefft

// Whether the wave-private kernel serves this mel plan (frame-major output only).
bool forward_lane_ok(int n_fft, int hop, int n_mels, int nnz) {
  return n_fft == lanefft::N && hop >= 2 && hop % 2 == 0 && hop <= lanefft::N && n_mels >= 16 &&
         n_mels % 16 == 0 && lanefft::lds_bytes(hop, nnz, n_mels) <= 160 * 1024;
}

int launch_forward_lane(const FwdParams& p0, int cus, hipStream_t s) {
  using namespace lanefft;
  FwdParams p = p0;
  const long long steps = p.B * ((p.F + FPW - 1) / FPW);
  if (steps <= 0) return MIA_OK;
  if (steps >= (1LL << 31)) {
    set_error("%lld steps do not fit one launch (split the batch)", steps);
    return MIA_ERR_UNSUPPORTED;
  }
  const size_t lds = lds_bytes(p.hop, p.nnz, p.n_mels);
  static bool attr_set[16] = {false};  // per device: the attribute belongs to the device's copy of the kernel
  int dev = 0;
  MIA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16 || !attr_set[dev]) {
    MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_lane_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (dev >= 0 && dev < 16) attr_set[dev] = true;
  }
  long long grid = cus > 0 ? cus : 256;   // one 8-wave workgroup per CU
  if (grid * WAVES > steps) grid = (steps + WAVES - 1) / WAVES;
  hipLaunchKernelGGL(fwd_lane_kernel, dim3((unsigned)grid), dim3(WG), lds, s, p);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // namespace mia

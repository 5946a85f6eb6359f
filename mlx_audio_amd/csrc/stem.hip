// Whisper encoder stem: what the model does with the log-mel spectrogram before its transformer
// blocks (models/whisper/layers/encoder.py:72-109): Conv1d(n_mels -> n_state, k = 3, pad 1) ->
// GELU -> Conv1d(n_state -> n_state, k = 3, stride 2, pad 1) -> GELU -> + positional embedding.
//
// Both convolutions run as one GEMM kernel on the matrix cores (v_mfma_f32_32x32x16_f16, float32
// accumulation).  With the activations stored channels-last and ONE ZERO ROW in front of and
// behind every clip, the three taps of output frame t are 3 * C_in contiguous halves starting at
// row stride * t of the clip, so the "im2col" matrix is just an overlapping strided view of the
// activation buffer: row m of the GEMM starts at (clip * rows + stride * t) * C_in, K = 3 * C_in,
// and the weights in the reference's own (out, tap, in) order are the other operand as they are.
//
//   stem_prep_kernel   log-mel (clips, n_mels, frames) float32, frames contiguous
//                      -> (clips, frames + 2, C_pad) float16, channels contiguous, zero rows/pad
//   stem_gemm_kernel   tile = 256 output channels x 256 output frames x 64 of K with 8 waves, each
//                      128 x 64 as 4 x 2 accumulators of 32 x 32 (half the operand bytes per
//                      multiply-add of a 128 x 128 tile, which serves small batches and channel
//                      counts 256 does not divide); both operand tiles arrive by
//                      LDS-DMA (global_load_lds_dwordx4) into a double buffer, XOR-swizzled on the
//                      source address so that the ds_read_b128 fragment reads are conflict-free;
//                      epilogue: + bias, GELU, (+ positional embedding), float16 rows staged
//                      through LDS and stored as whole 256-byte runs (or float32 from registers).
// The weight operand is the MFMA's A (rows = output channels), so a lane of the result holds
// four consecutive channels of one frame: bias and embedding are float4 loads, the float16
// result one 8-byte LDS store.
//
// Two operand precisions share the kernel.  MIA_STEM_F16: weights, spectrogram and the
// conv1 -> GELU intermediate rounded to float16 (the arithmetic of a float16 checkpoint).
// MIA_STEM_F32: everything stays float32 and the products run on v_mfma_f32_32x32x2_f32, whose
// result equals a chain of fused multiply-adds in float32 - the arithmetic of the reference's
// nn.Conv1d on a float32 checkpoint (its default), at 1/16 of the float16 rate.  The LDS tiles
// are the same bytes either way (128-byte rows = 64 halves or 32 floats of K); a lane's 16-byte
// fragment is then four consecutive k of its row, fed to four successive MFMAs (the two lane
// halves take different chunks, which only permutes the order of the sum over k).
#include "mia_internal.h"

namespace mia {
namespace stem {
constexpr int BKB = 128;  // bytes of K per step: the operand rows in LDS
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct Gemm {
  const void* act;      // (clips, rows_in, cin): zero row, frames, zero row
  long long rows_in;
  int cin;              // elements per row, a multiple of a 128-byte K-step (64 halves / 32 floats)
  int stride;           // 1 or 2
  int t_out;            // output frames per clip
  int m_total;          // clips * t_out
  const void* w;        // (n, 3 * cin)
  const float* bias;    // (n)
  const float* pos;     // (t_out, n) or null
  int n;                // a multiple of 128
  void* out;            // rows of n elements
  long long out_rows;   // rows per clip in out
  int out_row0;         // row of a clip's first frame in out (1 when out carries the zero rows)
};

// One wave-wide LDS-DMA load: lane L's 16 bytes at gptr land at lds_byte_addr + 16 L.  The
// compiler does not count it; the kernel waits with an explicit s_waitcnt vmcnt(0).
__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_byte_addr) {
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

// x * Phi(x) with Phi from the complementary error function in the form
// erfc(z) = t (a1 + t (a2 + ... a5 t)) exp(-z^2), t = 1 / (1 + p z) (absolute error 1.5e-7,
// no cancellation in the negative tail).
__device__ __forceinline__ float gelu(float x) {
  const float z = __builtin_fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float q = fmaf(t, 1.061405429f, -1.453152027f);
  q = fmaf(t, q, 1.421413741f);
  q = fmaf(t, q, -0.284496736f);
  q = fmaf(t, q, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);
  const float half_erfc = 0.5f * t * q * e;  // Phi(-|x|)
  return x * (x < 0.0f ? half_erfc : 1.0f - half_erfc);
}
}  // namespace stem

// Tile shape: NWR x NWC waves, each RN x RM accumulators of 32 channels x 32 frames.
template <int NWR_, int NWC_, int RN_, int RM_, int OCC_>
struct StemTile {
  static constexpr int NWR = NWR_, NWC = NWC_, RN = RN_, RM = RM_, OCC = OCC_;
  static constexpr int NW = NWR * NWC, WG = 64 * NW;
  static constexpr int BN = 32 * RN * NWR, BM = 32 * RM * NWC;       // channels x frames of a tile
  static constexpr int W_BYTES = BN * 128, A_BYTES = BM * 128;       // operand tiles of one K-step
  static constexpr int BUF_BYTES = W_BYTES + A_BYTES;
  static constexpr int PW = BN / (8 * NW), PA = BM / (8 * NW);       // DMA pieces per wave and K-step
  static constexpr int STAGE_PITCH = BN * 2 + 16;                    // float16 output rows in LDS
  static constexpr int LDS_BYTES = 2 * BUF_BYTES > BM * STAGE_PITCH ? 2 * BUF_BYTES : BM * STAGE_PITCH;
  static_assert(BN % (8 * NW) == 0 && BM % (8 * NW) == 0, "whole DMA pieces per wave");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};
using StemTile128 = StemTile<2, 2, 2, 2, 2>;  // 128 x 128, 4 waves, two workgroups per CU
using StemTile256 = StemTile<2, 4, 4, 2, 1>;  // 256 x 256, 8 waves, one workgroup per CU

template <class TL, class ET, bool OUT_F16>
__global__ __launch_bounds__(TL::WG, TL::OCC) void stem_gemm_kernel(stem::Gemm p) {
  using namespace stem;
  constexpr int BN = TL::BN, BM = TL::BM, WG = TL::WG;
  constexpr bool F32 = sizeof(ET) == 4;
  constexpr int ES = (int)sizeof(ET);  // bytes per operand element
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  // Workgroups go to the eight XCDs in turn; consecutive LOGICAL tiles (the column tiles that
  // share one activation tile) are given to one XCD so that its L2 serves the re-reads.
  const unsigned nb = gridDim.x, id = blockIdx.x;
  const unsigned q8 = nb / 8, r8 = nb % 8, xcd = id % 8;
  const unsigned logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + id / 8;
  const int nt = p.n / BN;
  const int n0 = (int)(logical % (unsigned)nt) * BN;
  const int m0 = (int)(logical / (unsigned)nt) * BM;
  const int K = 3 * p.cin;

  // ---- sources of this thread's DMA pieces: piece s = (P wv + i) * 64 + lane of a tile is
  // LDS slot s = row * 8 + c'; it holds the 16 bytes of chunk c = c' ^ ((row >> 1) & 7) ----
  const char* wsrc[TL::PW];
  const char* asrc[TL::PA];
#pragma unroll
  for (int i = 0; i < TL::PW; ++i) {
    const int s = (TL::PW * wv + i) * 64 + lane;
    const int row = s >> 3, c = (s & 7) ^ ((row >> 1) & 7);
    wsrc[i] = static_cast<const char*>(p.w) + ((size_t)(n0 + row) * K * ES + 16 * c);
  }
#pragma unroll
  for (int i = 0; i < TL::PA; ++i) {
    const int s = (TL::PA * wv + i) * 64 + lane;
    const int row = s >> 3, c = (s & 7) ^ ((row >> 1) & 7);
    int m = m0 + row;
    m = m < p.m_total ? m : p.m_total - 1;  // rows past the end repeat the last one; never stored
    const int clip = m / p.t_out, t = m - clip * p.t_out;
    asrc[i] = static_cast<const char*>(p.act) +
              (((size_t)clip * p.rows_in + (size_t)p.stride * t) * p.cin * ES + 16 * c);
  }
  const unsigned lds0 = (unsigned)(unsigned long long)(smem);
  const unsigned wdst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + 1024u * TL::PW * wv));
  const unsigned adst =
      (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + TL::W_BYTES + 1024u * TL::PA * wv));
  auto issue = [&](int ks, int buf) {
#pragma unroll
    for (int i = 0; i < TL::PW; ++i) dma16(wsrc[i] + ks * BKB, wdst + buf * TL::BUF_BYTES + 1024u * i);
#pragma unroll
    for (int i = 0; i < TL::PA; ++i) dma16(asrc[i] + ks * BKB, adst + buf * TL::BUF_BYTES + 1024u * i);
  };

  // ---- fragment reads: lane l holds elements k = 8 (l >> 5) + j of row l & 31 -------------
  const int wr = wv / TL::NWC, wc = wv % TL::NWC;  // the wave's block of channels / of frames
  const int r31 = lane & 31;
  const int e = (lane >> 5) ^ ((r31 >> 1) & 7);
  const unsigned char* wbase = smem + (wr * 32 * TL::RN + r31) * 128;
  const unsigned char* abase = smem + TL::W_BYTES + (wc * 32 * TL::RM + r31) * 128;

  f16v acc[TL::RN][TL::RM];
#pragma unroll
  for (int a = 0; a < TL::RN; ++a)
#pragma unroll
    for (int b = 0; b < TL::RM; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

  const int nk = K * ES / BKB;
  // One K-step from LDS buffer `buf`: four sub-steps of 16.  Measured on this loop (64 clips,
  // 1.46 ms for both convolutions): without the LDS-DMA loads 1.09 ms, without the MFMAs 1.22 ms
  // - the two overlap only in part, and neither alone is near the matrix peak (0.5 ms).  What
  // did NOT move it: operands made L2-resident (same time: not a memory-latency problem),
  // fragments of sub-step kk + 1 read into a second register set ahead of the MFMAs of kk,
  // 16 waves of 64 x 64, the activation tile through registers instead of LDS-DMA, a
  // weight-stationary tile order (8 % slower).  The remaining lever is the schedule itself:
  // LDS-DMA kept in flight across the barrier with counted waits and the loads, fragment reads
  // and MFMAs of neighbouring steps interleaved phase by phase.
  auto compute = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int off = buf * TL::BUF_BYTES + (((2 * kk) ^ e) << 4);
      if constexpr (F32) {
        // the lane's 16 bytes are k = 4 (2 kk + half) + j, j = 0..3, of its row: MFMA j takes
        // element j from both operands (a permutation of k, the same for both)
        float4 a[TL::RN], b[TL::RM];
#pragma unroll
        for (int i = 0; i < TL::RN; ++i) a[i] = *reinterpret_cast<const float4*>(wbase + off + i * 32 * 128);
#pragma unroll
        for (int i = 0; i < TL::RM; ++i) b[i] = *reinterpret_cast<const float4*>(abase + off + i * 32 * 128);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ni = 0; ni < TL::RN; ++ni)
#pragma unroll
            for (int mi = 0; mi < TL::RM; ++mi) {
              const float av = j == 0 ? a[ni].x : (j == 1 ? a[ni].y : (j == 2 ? a[ni].z : a[ni].w));
              const float bv = j == 0 ? b[mi].x : (j == 1 ? b[mi].y : (j == 2 ? b[mi].z : b[mi].w));
              acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ni][mi], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
      } else {
        h8 a[TL::RN], b[TL::RM];
#pragma unroll
        for (int i = 0; i < TL::RN; ++i) a[i] = *reinterpret_cast<const h8*>(wbase + off + i * 32 * 128);
#pragma unroll
        for (int i = 0; i < TL::RM; ++i) b[i] = *reinterpret_cast<const h8*>(abase + off + i * 32 * 128);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < TL::RN; ++ni)
#pragma unroll
          for (int mi = 0; mi < TL::RM; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ni], b[mi], acc[ni][mi], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }
  };
  {
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
      const int buf = ks & 1;
      if (ks + 1 < nk) issue(ks + 1, buf ^ 1);
      compute(buf);
      // the next tile has landed, and every wave is done reading this one
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue: acc[ni][mi][reg] = channel n0 + 32 (RN wr + ni) + (reg & 3) + 8 (reg >> 2)
  // + 4 (lane >> 5) of frame m0 + 32 (RM wc + mi) + (lane & 31) --------------------------------
  const int h = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < TL::RM; ++mi) {
    const int ml = 32 * (TL::RM * wc + mi) + r31;
    const int m = m0 + ml;
    const bool live = m < p.m_total;
    const int mc = live ? m : p.m_total - 1;
    const int clip = mc / p.t_out, t = mc - clip * p.t_out;
    const float* __restrict__ prow = p.pos ? p.pos + (size_t)t * p.n : nullptr;
    float* __restrict__ orow =
        reinterpret_cast<float*>(p.out) + ((size_t)clip * p.out_rows + p.out_row0 + t) * p.n;
#pragma unroll
    for (int ni = 0; ni < TL::RN; ++ni) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = 32 * (TL::RN * wr + ni) + 8 * g + 4 * h;
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n0 + nl);
        float4 v = {gelu(acc[ni][mi][4 * g] + bv.x), gelu(acc[ni][mi][4 * g + 1] + bv.y),
                    gelu(acc[ni][mi][4 * g + 2] + bv.z), gelu(acc[ni][mi][4 * g + 3] + bv.w)};
        if (prow) {
          const float4 pv = *reinterpret_cast<const float4*>(prow + n0 + nl);
          v.x += pv.x, v.y += pv.y, v.z += pv.z, v.w += pv.w;
        }
        if (OUT_F16) {
          const h4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
          *reinterpret_cast<h4*>(smem + ml * TL::STAGE_PITCH + nl * 2) = hv;
        } else if (live) {
          *reinterpret_cast<float4*>(orow + n0 + nl) = v;
        }
      }
    }
  }
  if (OUT_F16) {
    __syncthreads();
    // BM rows x BN / 8 pieces of 16 bytes: BN / 8 consecutive lanes store one row's run
    constexpr int PIECES = BN / 8;
#pragma unroll 4
    for (int idx = tid; idx < BM * PIECES; idx += WG) {
      const int row = idx / PIECES, cp = idx % PIECES;
      const int m = m0 + row;
      if (m < p.m_total) {
        const int clip = m / p.t_out, t = m - clip * p.t_out;
        const uint4 v = *reinterpret_cast<const uint4*>(smem + row * TL::STAGE_PITCH + cp * 16);
        _Float16* __restrict__ orow =
            reinterpret_cast<_Float16*>(p.out) + ((size_t)clip * p.out_rows + p.out_row0 + t) * p.n;
        *reinterpret_cast<uint4*>(orow + n0 + 8 * cp) = v;
      }
    }
  }
}

template <class TL, class ET, bool OUT_F16>
static int launch_stem_gemm(const stem::Gemm& g, hipStream_t s) {
  const long long tiles = ((long long)g.m_total + TL::BM - 1) / TL::BM * (g.n / TL::BN);
  if (tiles >= (1LL << 31)) {
    set_error("batch too large for one launch");
    return MIA_ERR_UNSUPPORTED;
  }
  auto kern = stem_gemm_kernel<TL, ET, OUT_F16>;
  static bool attr_set = false;  // (per instantiation; the attribute is a property of the function)
  if (!attr_set) {
    MIA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, TL::LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(TL::WG), TL::LDS_BYTES, s, g);
  return MIA_OK;
}
// The 256 x 256 tile moves half the operand bytes per multiply-add; it needs enough tiles to
// fill the chip and a channel count it divides.
template <class ET, bool OUT_F16>
static int dispatch_stem_gemm(const stem::Gemm& g, int cus, hipStream_t s) {
  const long long tiles256 = ((long long)g.m_total + 255) / 256 * (g.n / 256);
  if (g.n % 256 == 0 && tiles256 >= 2LL * cus) return launch_stem_gemm<StemTile256, ET, OUT_F16>(g, s);
  return launch_stem_gemm<StemTile128, ET, OUT_F16>(g, s);
}

// log-mel (clips, C, T) float32 -> (clips, T + 2, CP) of ET, channels contiguous; rows 0 and
// T + 1 and channels >= C are zero.  grid (ceil((T + 2) / 64), clips): a workgroup writes rows
// 64 bx .. 64 bx + 63 of the padded clip (frame = row - 1).
template <class ET>
__global__ __launch_bounds__(256) void stem_prep_kernel(const float* __restrict__ mel, int C, int T, int CP,
                                                        ET* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char prep_raw[];
  ET* tile = reinterpret_cast<ET*>(prep_raw);  // [64][CP + 16 bytes]
  constexpr int PER16 = 16 / (int)sizeof(ET);  // elements per 16-byte piece
  const int pitch = CP + PER16;
  const long long clip = blockIdx.y;
  const int r0 = blockIdx.x * 64;
  const int tid = threadIdx.x, tl = tid & 63;
  const int frame = r0 + tl - 1;
  const bool in = frame >= 0 && frame < T;
  const float* __restrict__ src = mel + clip * (long long)C * T + (in ? frame : 0);
  for (int c = tid >> 6; c < CP; c += 4) {
    const float v = (in && c < C) ? src[(long long)c * T] : 0.0f;
    tile[tl * pitch + c] = (ET)v;
  }
  __syncthreads();
  const int pieces = CP / PER16;  // 16-byte pieces per row
  ET* __restrict__ dst = out + (clip * (long long)(T + 2) + r0) * CP;
  for (int idx = tid; idx < 64 * pieces; idx += 256) {
    const int row = idx / pieces, cp = idx - row * pieces;
    if (r0 + row < T + 2) {
      *reinterpret_cast<uint4*>(dst + (long long)row * CP + PER16 * cp) =
          *reinterpret_cast<const uint4*>(tile + row * pitch + PER16 * cp);
    }
  }
}

// rows 0 and rows - 1 of every clip of a (clips, rows, n) array := 0
template <class ET>
__global__ __launch_bounds__(256) void stem_zero_rows_kernel(ET* __restrict__ x, long long rows, int n) {
  const long long clip = blockIdx.y;
  ET* __restrict__ a = x + clip * rows * n;
  ET* __restrict__ b = a + (rows - 1) * n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    a[i] = (ET)0.0f;
    b[i] = (ET)0.0f;
  }
}

// (n, 3, c) float32 -> (n, 3, cp) of ET, zero-padded channels
template <class ET>
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ w, long long rows, int c,
                                                        int cp, ET* __restrict__ out) {
  const long long total = rows * cp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / cp;
    const int ch = (int)(i - r * cp);
    out[i] = ch < c ? (ET)w[r * c + ch] : (ET)0.0f;
  }
}

// channels padded to whole 128-byte K-steps
static int stem_cpad(int c, size_t es) {
  const int q = (int)(128 / es);
  return (c + q - 1) / q * q;
}
static size_t stem_align(size_t x) { return (x + 255) & ~(size_t)255; }
static bool stem_precision_ok(int precision) { return precision == MIA_STEM_F16 || precision == MIA_STEM_F32; }
static size_t stem_es(int precision) { return precision == MIA_STEM_F32 ? 4 : 2; }

template <class ET>
static int stem_pack_t(const float* conv1_w, const float* conv2_w, int n_mels, int n_state, void* packed,
                       hipStream_t s) {
  const int cp = stem_cpad(n_mels, sizeof(ET));
  ET* w1 = static_cast<ET*>(packed);
  ET* w2 = reinterpret_cast<ET*>(static_cast<unsigned char*>(packed) +
                                 stem_align((size_t)n_state * 3 * cp * sizeof(ET)));
  hipLaunchKernelGGL(stem_pack_kernel<ET>, dim3(1024), dim3(256), 0, s, conv1_w, (long long)n_state * 3,
                     n_mels, cp, w1);
  hipLaunchKernelGGL(stem_pack_kernel<ET>, dim3(4096), dim3(256), 0, s, conv2_w, (long long)n_state * 3,
                     n_state, n_state, w2);
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

template <class ET>
static int stem_run_t(const float* mel, int64_t clips, int64_t frames, int n_mels, int n_state,
                      const void* packed, const float* conv1_b, const float* conv2_b, const float* pos,
                      void* out, int out_f16, void* workspace, hipStream_t s) {
  const int64_t t_out = (frames - 1) / 2 + 1;
  const int cp = stem_cpad(n_mels, sizeof(ET));
  const ET* w1 = static_cast<const ET*>(packed);
  const ET* w2 = reinterpret_cast<const ET*>(static_cast<const unsigned char*>(packed) +
                                             stem_align((size_t)n_state * 3 * cp * sizeof(ET)));
  ET* x0 = static_cast<ET*>(workspace);
  ET* h1 = reinterpret_cast<ET*>(static_cast<unsigned char*>(workspace) +
                                 stem_align((size_t)clips * (frames + 2) * cp * sizeof(ET)));
  constexpr int PER16 = 16 / (int)sizeof(ET);
  const unsigned gy = 32768;  // clips per launch of the two small kernels (grid.y limit)
  for (int64_t c0 = 0; c0 < clips; c0 += gy) {
    const unsigned nc = (unsigned)(clips - c0 < gy ? clips - c0 : gy);
    hipLaunchKernelGGL(stem_prep_kernel<ET>, dim3((unsigned)((frames + 2 + 63) / 64), nc), dim3(256),
                       (size_t)64 * (cp + PER16) * sizeof(ET), s, mel + c0 * n_mels * frames, n_mels,
                       (int)frames, cp, x0 + c0 * (frames + 2) * cp);
    hipLaunchKernelGGL(stem_zero_rows_kernel<ET>, dim3(2, nc), dim3(256), 0, s,
                       h1 + c0 * (frames + 2) * n_state, (long long)(frames + 2), n_state);
  }
  int dev = 0, cus = 256;
  MIA_CHECK_HIP(hipGetDevice(&dev));
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  constexpr bool H1_F16 = sizeof(ET) == 2;  // the intermediate has the operands' type
  stem::Gemm g1{x0, frames + 2, cp, 1, (int)frames, (int)(clips * frames), w1, conv1_b, nullptr, n_state,
                h1, frames + 2, 1};
  int rc = dispatch_stem_gemm<ET, H1_F16>(g1, cus, s);
  if (rc != MIA_OK) return rc;
  stem::Gemm g2{h1, frames + 2, n_state, 2, (int)t_out, (int)(clips * t_out), w2, conv2_b, pos, n_state,
                out, t_out, 0};
  rc = out_f16 ? dispatch_stem_gemm<ET, true>(g2, cus, s) : dispatch_stem_gemm<ET, false>(g2, cus, s);
  if (rc != MIA_OK) return rc;
  MIA_CHECK_HIP(hipGetLastError());
  return MIA_OK;
}

}  // namespace mia

using namespace mia;

extern "C" {

size_t mia_whisper_stem_weight_bytes(int n_mels, int n_state, int precision) {
  if (n_mels <= 0 || n_state <= 0 || !stem_precision_ok(precision)) return 0;
  const size_t es = stem_es(precision);
  return stem_align((size_t)n_state * 3 * stem_cpad(n_mels, es) * es) +
         stem_align((size_t)n_state * 3 * n_state * es);
}

int mia_whisper_stem_pack(const float* conv1_w, const float* conv2_w, int n_mels, int n_state,
                          int precision, void* packed, void* stream) {
  MIA_REQUIRE(conv1_w && conv2_w && packed, "null argument");
  MIA_REQUIRE(stem_precision_ok(precision), "unknown precision %d", precision);
  MIA_REQUIRE(n_mels > 0 && n_state > 0 && n_state % 128 == 0,
              "n_state must be a positive multiple of 128 (got %d)", n_state);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return precision == MIA_STEM_F32 ? stem_pack_t<float>(conv1_w, conv2_w, n_mels, n_state, packed, s)
                                   : stem_pack_t<_Float16>(conv1_w, conv2_w, n_mels, n_state, packed, s);
}

size_t mia_whisper_stem_workspace_bytes(int64_t clips, int64_t frames, int n_mels, int n_state,
                                        int precision) {
  if (clips <= 0 || frames <= 0 || n_mels <= 0 || n_state <= 0 || !stem_precision_ok(precision)) return 0;
  const size_t es = stem_es(precision);
  return stem_align((size_t)clips * (frames + 2) * stem_cpad(n_mels, es) * es) +
         stem_align((size_t)clips * (frames + 2) * n_state * es);
}

int mia_whisper_stem(const float* mel, int64_t clips, int64_t frames, int n_mels, int n_state,
                     int precision, const void* packed, const float* conv1_b, const float* conv2_b,
                     const float* pos, void* out, int out_f16, void* workspace, size_t workspace_bytes,
                     void* stream) {
  MIA_REQUIRE(clips >= 0 && frames > 0, "bad shape");
  MIA_REQUIRE(stem_precision_ok(precision), "unknown precision %d", precision);
  if (clips == 0) return MIA_OK;
  MIA_REQUIRE(mel && packed && conv1_b && conv2_b && out && workspace, "null argument");
  MIA_REQUIRE(n_mels > 0 && n_state > 0 && n_state % 128 == 0,
              "n_state must be a positive multiple of 128 (got %d)", n_state);
  MIA_REQUIRE(workspace_bytes >= mia_whisper_stem_workspace_bytes(clips, frames, n_mels, n_state, precision),
              "workspace too small");
  MIA_REQUIRE(clips * frames < (1LL << 31) - 256, "batch too large for one call: %lld frames (split the batch)",
              (long long)(clips * frames));
  MIA_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(conv1_b) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(conv2_b) & 15) == 0 && (reinterpret_cast<uintptr_t>(pos) & 15) == 0,
              "pointers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (precision == MIA_STEM_F32) {
    return stem_run_t<float>(mel, clips, frames, n_mels, n_state, packed, conv1_b, conv2_b, pos, out,
                             out_f16, workspace, s);
  }
  return stem_run_t<_Float16>(mel, clips, frames, n_mels, n_state, packed, conv1_b, conv2_b, pos, out,
                              out_f16, workspace, s);
}

}  // extern "C"

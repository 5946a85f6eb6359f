// extern "C" surface of libmia_hip.so: error reporting, host-side table
// builders, plan management and the launch wrappers.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>

#include "mia_internal.h"

namespace mia {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

constexpr int64_t kMaxGridRows = 65535;  // gridDim.y of the row-per-block-row launches
int launch_overlap_add(const float*, const float*, long long, long long, int, int, long long,
                       float, float*, hipStream_t);
int launch_pad(const float*, long long, long long, long long, int, float*, hipStream_t);
int launch_frame(const float*, long long, long long, long long, int, int, float*, hipStream_t);
int launch_to_db(const float*, long long, long long, float, float, float, float, int, float*,
                 int*, hipStream_t);
int launch_floor_clips(float*, long long, long long, int*, const int*, int, float, int,
                       hipStream_t);
int launch_dct2(const float*, long long, int, long long, const float*, int, float*, const int*,
                long long, float, hipStream_t);
int launch_magnitude(const float2*, long long, float*, hipStream_t);
int launch_from_db(const float*, long long, float, float, float*, hipStream_t);
int launch_reduce_keys(int*, long long, hipStream_t);
int launch_phase(const float2*, long long, float*, hipStream_t);

// --------------------------- table builders --------------------------------

constexpr double kSlaneyFMin = 0.0;
constexpr double kSlaneyFSp = 200.0 / 3;
constexpr double kSlaneyMinLogHz = 1000.0;
static const double kSlaneyLogstep = std::log(6.4) / 27.0;
constexpr double kSlaneyMinLogMel = (kSlaneyMinLogHz - kSlaneyFMin) / kSlaneyFSp;
constexpr double kHtkFactor = 2595.0, kHtkBase = 700.0;
constexpr double kFilterbankEps = 1e-10;

static double hz_to_mel1(double f, bool htk) {
  if (htk) return kHtkFactor * std::log10(1.0 + f / kHtkBase);
  if (f < kSlaneyMinLogHz) return (f - kSlaneyFMin) / kSlaneyFSp;
  return kSlaneyMinLogMel + std::log(f / kSlaneyMinLogHz) / kSlaneyLogstep;
}
static double mel_to_hz1(double m, bool htk) {
  if (htk) return kHtkBase * (std::pow(10.0, m / kHtkFactor) - 1.0);
  if (m < kSlaneyMinLogMel) return kSlaneyFMin + kSlaneyFSp * m;
  return kSlaneyMinLogHz * std::exp(kSlaneyLogstep * (m - kSlaneyMinLogMel));
}

// numpy.linspace(a, b, n): a + i*step with step = (b-a)/(n-1), last point = b.
static std::vector<double> linspace(double a, double b, int n) {
  std::vector<double> v((size_t)n);
  if (n == 1) {
    v[0] = a;
    return v;
  }
  const double step = (b - a) / double(n - 1);
  for (int i = 0; i < n; ++i) v[(size_t)i] = a + double(i) * step;
  v[(size_t)n - 1] = b;
  return v;
}

static int window_kind(const char* type) {
  static const struct { const char* name; int kind; } names[] = {
      {"hann", 0}, {"hanning", 0}, {"hamming", 1}, {"blackman", 2}, {"bartlett", 3},
      {"triangular", 3}, {"rectangular", 4}, {"boxcar", 4}, {"ones", 4}};
  for (auto& e : names) {
    if (strcasecmp(type, e.name) == 0) return e.kind;
  }
  return -1;
}

}  // namespace mia

using namespace mia;

extern "C" {

const char* mia_last_error(void) { return g_last_error.c_str(); }
int mia_version(void) { return 100; }

int mia_generate_window(const char* type, int length, int periodic, float* out) {
  MIA_REQUIRE(type != nullptr && out != nullptr, "null argument");
  MIA_REQUIRE(length > 0, "Window length must be positive, got %d", length);
  const int kind = window_kind(type);
  MIA_REQUIRE(kind >= 0,
              "Unknown window type: '%s'. Supported: bartlett, blackman, boxcar, hamming, "
              "hann, hanning, ones, rectangular, triangular", type);
  const int n = periodic ? length + 1 : length;
  for (int k = 0; k < length; ++k) {
    double w;
    if (kind == 4 || n <= 1) {
      w = 1.0;
    } else if (kind == 3) {
      w = 1.0 - std::fabs(2.0 * k / double(n - 1) - 1.0);
    } else {
      const double d = double(n - 1);
      if (kind == 0) {
        w = 0.5 + (-1.0) * 0.5 * std::cos(2.0 * 1 * M_PI * k / d);
      } else if (kind == 1) {
        w = 0.54 + (-1.0) * 0.46 * std::cos(2.0 * 1 * M_PI * k / d);
      } else {
        w = 0.42 + (-1.0) * 0.5 * std::cos(2.0 * 1 * M_PI * k / d);
        w = w + 1.0 * 0.08 * std::cos(2.0 * 2 * M_PI * k / d);
        w = std::fmax(w, 0.0);
      }
    }
    out[k] = (float)w;
  }
  return MIA_OK;
}

int mia_hz_to_mel(const double* hz, int n, int htk, double* out) {
  MIA_REQUIRE(hz && out && n >= 0, "bad argument");
  for (int i = 0; i < n; ++i) out[i] = hz_to_mel1(hz[i], htk != 0);
  return MIA_OK;
}
int mia_mel_to_hz(const double* mels, int n, int htk, double* out) {
  MIA_REQUIRE(mels && out && n >= 0, "bad argument");
  for (int i = 0; i < n; ++i) out[i] = mel_to_hz1(mels[i], htk != 0);
  return MIA_OK;
}

int mia_mel_filterbank(int sr, int n_fft, int n_mels, double fmin, double fmax, int htk,
                       int norm, float* out) {
  MIA_REQUIRE(out != nullptr, "null output");
  MIA_REQUIRE(n_mels > 0, "n_mels must be positive, got %d", n_mels);
  MIA_REQUIRE(fmin >= 0, "fmin must be non-negative, got %g", fmin);
  MIA_REQUIRE(fmin < fmax, "fmin (%g) must be less than fmax (%g)", fmin, fmax);
  MIA_REQUIRE(fmax <= sr / 2.0, "fmax (%g) cannot exceed Nyquist frequency (%g)", fmax,
              sr / 2.0);
  MIA_REQUIRE(norm == 0 || norm == 1, "Unknown norm code %d", norm);
  const int n_freqs = 1 + n_fft / 2;
  const std::vector<double> fft_freqs = linspace(0.0, sr / 2.0, n_freqs);
  const std::vector<double> mel_pts =
      linspace(hz_to_mel1(fmin, htk != 0), hz_to_mel1(fmax, htk != 0), n_mels + 2);
  std::vector<double> hz((size_t)n_mels + 2);
  for (int i = 0; i < n_mels + 2; ++i) hz[(size_t)i] = mel_to_hz1(mel_pts[(size_t)i], htk != 0);
  for (int m = 0; m < n_mels; ++m) {
    const double lo = hz[(size_t)m], ce = hz[(size_t)m + 1], hi = hz[(size_t)m + 2];
    const double enorm = 2.0 / (hi - lo);
    for (int k = 0; k < n_freqs; ++k) {
      const double f = fft_freqs[(size_t)k];
      const double lower = (f - lo) / (ce - lo + kFilterbankEps);
      const double upper = (hi - f) / (hi - ce + kFilterbankEps);
      float w = (float)std::fmax(0.0, std::fmin(lower, upper));
      if (norm == 1) w = (float)(double(w) * enorm);  // float32 array *= float64 scale
      out[(size_t)m * n_freqs + k] = w;
    }
  }
  return MIA_OK;
}

int mia_dct_matrix(int n_out, int n_in, int ortho, float* out) {
  MIA_REQUIRE(out != nullptr && n_out > 0 && n_in > 0, "bad argument");
  for (int k = 0; k < n_out; ++k) {
    for (int n = 0; n < n_in; ++n) {
      double v = std::cos(M_PI * double(k) * double(2 * n + 1) / double(2 * n_in));
      if (ortho) v *= (k == 0) ? 1.0 / std::sqrt(double(n_in)) : std::sqrt(2.0 / double(n_in));
      out[(size_t)k * n_in + n] = (float)v;
    }
  }
  return MIA_OK;
}

// ------------------------------- plans -------------------------------------

int mia_fft_size_supported(int n_fft) { return fft_size_supported(n_fft) ? 1 : 0; }

}  // extern "C"

// Clamp keys of mia_melspectrogram_static: the two key arrays (per-clip maximum and
// minimum, MIA_STATIC_MAX_CLIPS + 1 entries each) that mia_melspectrogram takes from
// its workspace argument.  Every GPU the library runs on holds its own copy of this
// array at its own address until the process ends.  In host code the name is only a
// handle for hipGetSymbolAddress: kernels and memsets get the device address that
// scratch_resolve() stored for the device, never the name.
static constexpr size_t kStaticKeyBytes =
    (sizeof(int) * ((size_t)MIA_STATIC_MAX_CLIPS + 1) + 255) & ~(size_t)255;
static __device__ __align__(256) unsigned char scratch[2 * kStaticKeyBytes];

static constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static void* g_scratch_ptr[kMaxDevices] = {};

// Device address of `scratch` on `device`, which must be the current device; looked up
// once per device (mia_plan_create does it, so every plan's device is covered).
static int scratch_resolve(int device, void** out) {
  MIA_REQUIRE(device >= 0 && device < kMaxDevices, "device index %d out of range", device);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[device] == nullptr) {
    int cur = -1;
    MIA_CHECK_HIP(hipGetDevice(&cur));
    MIA_REQUIRE(cur == device, "device %d is not current (current: %d)", device, cur);
    void* ptr = nullptr;
    MIA_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    MIA_REQUIRE(ptr != nullptr, "hipGetSymbolAddress returned a null address");
    g_scratch_ptr[device] = ptr;
  }
  *out = g_scratch_ptr[device];
  return MIA_OK;
}

static void plan_free(mia_plan* p) {
  if (!p) return;
  (void)hipFree(p->d_window);
  (void)hipFree(p->d_tw);
  (void)hipFree(p->d_row_lo);
  (void)hipFree(p->d_row_len);
  (void)hipFree(p->d_row_off);
  (void)hipFree(p->d_wts);
  (void)hipFree(p->d_dct);
  (void)hipFree(p->d_win16);
  (void)hipFree(p->d_tw16);
  (void)hipFree(p->d_winp);
  (void)hipFree(p->d_mfma_meta);
  (void)hipFree(p->d_mfma_a);
  (void)hipFree(p->d_sweep_meta);
  (void)hipFree(p->d_sweep_w);
  (void)hipFree(p->d_chirp);
  (void)hipFree(p->d_bspec);
  delete p;
}

template <class Tp>
static int upload(Tp** dst, const std::vector<Tp>& src) {
  const size_t bytes = sizeof(Tp) * (src.empty() ? 1 : src.size());
  MIA_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(dst), bytes));
  if (!src.empty()) {
    MIA_CHECK_HIP(hipMemcpy(*dst, src.data(), sizeof(Tp) * src.size(), hipMemcpyHostToDevice));
  }
  return MIA_OK;
}


// Tables of the lane-per-frame n_fft=400 / hop=160 kernel (logmel400.hip).
// Returns MIA_OK and leaves the plan without them when the filterbank is not
// of the "two taps per bin" triangular form the sweep needs.
static int build_w400_tables(mia_plan* p, const float* window, const float* basis, int n_mels,
                             int bins) {
  // segment s(k) of every bin: the bin feeds the rising edge of filter s and
  // the falling edge of filter s-1
  std::vector<int> seg((size_t)bins);
  std::vector<float> wr((size_t)bins, 0.0f), wf((size_t)bins, 0.0f);
  int cur = 0;
  for (int k = 0; k < bins; ++k) {
    int nz[3], n = 0;
    for (int m = 0; m < n_mels && n < 3; ++m) {
      if (basis[(size_t)m * bins + k] != 0.0f) nz[n++] = m;
    }
    if (n == 0) {
      seg[(size_t)k] = cur;
    } else if (n == 1) {
      const int s = nz[0] >= cur ? nz[0] : nz[0] + 1;
      if (s < cur) return MIA_OK;
      cur = s;
      seg[(size_t)k] = s;
    } else if (n == 2 && nz[1] == nz[0] + 1 && nz[1] >= cur) {
      cur = nz[1];
      seg[(size_t)k] = cur;
    } else {
      return MIA_OK;
    }
    const int s = seg[(size_t)k];
    if (s < n_mels) wr[(size_t)k] = basis[(size_t)s * bins + k];
    if (s >= 1 && s - 1 < n_mels) wf[(size_t)k] = basis[(size_t)(s - 1) * bins + k];
  }
  for (int m = 0; m < n_mels; ++m) {  // exact reconstruction or no fast path
    for (int k = 0; k < bins; ++k) {
      const int s = seg[(size_t)k];
      const float v = (m == s) ? wr[(size_t)k] : ((m == s - 1) ? wf[(size_t)k] : 0.0f);
      if (v != basis[(size_t)m * bins + k]) return MIA_OK;
    }
  }
  // Per-segment bin runs.  Segment s holds the bins between the centres of
  // filters s-1 and s; it may be empty (filters narrower than the bin spacing).
  std::vector<int> seg_count((size_t)n_mels + 1, 0), seg_first((size_t)n_mels + 1, -1);
  for (int k = 0; k < bins; ++k) {
    const int s = seg[(size_t)k];
    if (s > n_mels) return MIA_OK;
    if (seg_first[(size_t)s] < 0) seg_first[(size_t)s] = k;
    ++seg_count[(size_t)s];
  }
  auto bins_of = [&](int m0, int m1, int& klo) {  // bins with m0 <= s(k) <= m1
    int lo = -1, n = 0;
    for (int s = m0; s <= m1; ++s) {
      if (seg_count[(size_t)s] > 0 && lo < 0) lo = seg_first[(size_t)s];
      n += seg_count[(size_t)s];
    }
    klo = lo < 0 ? 0 : lo;
    return n;
  };
  // Deal contiguous runs of filters to the waves with even sweep cost
  // (about 2 instructions per bin, 7 per filter, 10 per 8-bin chunk): the
  // smallest bound for which a greedy left-to-right deal places every filter.
  constexpr int kWaves = W400_WAVES, kMaxChunks = W400_MEL_MAXC, kChunk = 8, kRec = W400_MEL_REC;
  auto cost = [&](int m0, int m1) {  // filters [m0, m1)
    int klo = 0;
    const int nb = bins_of(m0, m1, klo);
    return 12.0 * (m1 - m0) + 9.0 * nb + 80.0 * ((nb + kChunk - 1) / kChunk);
  };
  // Waves 4 j .. 4 j + 3 are the j-th oldest on their SIMDs and the issue arbiter
  // favours older waves, so the younger ones get a smaller share of the sweep.
  const double share[4] = {1.1, 1.05, 0.95, 0.85};
  auto deal = [&](double bound, int* starts) {
    int m = 0;
    for (int w = 0; w < kWaves; ++w) {
      starts[w] = m;
      const double bw = bound * share[w / 4];
      while (m < n_mels) {
        int klo = 0;
        if (bins_of(starts[w], m + 1, klo) > kMaxChunks * kChunk) break;
        if (cost(starts[w], m + 1) > bw) break;
        ++m;
      }
    }
    starts[kWaves] = m;
    return m == n_mels;
  };
  int starts[kWaves + 1];
  double lo_b = 0.0, hi_b = 1e9;
  if (!deal(hi_b, starts)) return MIA_OK;  // does not fit the per-wave limits
  for (int iter = 0; iter < 60; ++iter) {
    const double mid = 0.5 * (lo_b + hi_b);
    if (deal(mid, starts)) hi_b = mid; else lo_b = mid;
  }
  deal(hi_b, starts);
  // Per-wave sweep program.  The wave walks the bins of segments m0..m1 in
  // chunks of 8; record c holds, for its 8 bins, the weight into the filter
  // being completed (wA), the weight into the next filter (wB) and a control
  // word with, per bin, the number of filters that are complete after it.
  //   segment m0     : rising edge of m0          -> wA (the filter in progress)
  //   segment m0+j   : falling edge of m0+j-1     -> wA, rising edge of m0+j -> wB
  //   segment m1     : falling edge of m1-1       -> wA, nothing            -> wB
  std::vector<int> meta(kWaves * 16, 0);
  std::vector<float> sw((size_t)kWaves * kMaxChunks * kRec, 0.0f);
  for (int w = 0; w < kWaves; ++w) {
    const int m0 = starts[w];
    const int m1 = starts[w + 1];
    int* mt = &meta[(size_t)w * 16];
    mt[2] = m0;
    if (m1 == m0) continue;  // no filters: nchunks = npre = 0
    int klo = 0;
    const int nb = bins_of(m0, m1, klo);
    mt[0] = klo;
    mt[1] = (nb + kChunk - 1) / kChunk;
    float* rec = &sw[(size_t)w * kMaxChunks * kRec];
    int i = 0;        // bins emitted into the program so far
    int pending = 0;  // filters complete before the next bin
    int npre = 0;
    for (int s = m0; s <= m1; ++s) {
      for (int j = 0; j < seg_count[(size_t)s]; ++j, ++i) {
        const int k = seg_first[(size_t)s] + j;
        if (k != klo + i) return MIA_OK;
        float* r = rec + (size_t)(i / kChunk) * kRec;
        r[i % kChunk] = (s == m0) ? wr[(size_t)k] : wf[(size_t)k];
        r[kChunk + i % kChunk] = (s == m0 || s == m1) ? 0.0f : wr[(size_t)k];
      }
      // the end of segment s > m0 completes filter s-1
      if (s > m0) {
        if (i == 0) {
          ++npre;
        } else {
          float* r = rec + (size_t)((i - 1) / kChunk) * kRec;
          int ctrl;
          std::memcpy(&ctrl, &r[2 * kChunk], sizeof(int));
          const int sh = 4 * ((i - 1) % kChunk);
          if (((ctrl >> sh) & 15) == 15) return MIA_OK;
          ctrl += 1 << sh;
          std::memcpy(&r[2 * kChunk], &ctrl, sizeof(int));
        }
      }
      (void)pending;
    }
    mt[3] = npre;
    if (i != nb) return MIA_OK;
  }
  // FFT tables: role t's window taps and its step twiddles
  std::vector<float> win16((size_t)kWaves * 32, 0.0f);
  std::vector<float2> tw16((size_t)kWaves * 16, float2{1.0f, 0.0f});
  for (int t = 0; t < kWaves; ++t) {
    for (int r = 0; r < 25; ++r) win16[(size_t)t * 32 + r] = window[t + 16 * r];
    for (int k1 = 0; k1 < 13; ++k1) {
      const double a = -2.0 * M_PI * double((t * k1) % 400) / 400.0;
      tw16[(size_t)t * 16 + k1] = {(float)std::cos(a), (float)std::sin(a)};
    }
  }
  int rc;
  std::vector<float> winp((size_t)kWaves * 32, 0.0f);  // the prime-factor split's sample order
  for (int n1 = 0; n1 < kWaves; ++n1) {
    for (int n2 = 0; n2 < 25; ++n2) winp[(size_t)n1 * 32 + n2] = window[(25 * n1 + 16 * n2) % 400];
  }
  if ((rc = upload(&p->d_winp, winp)) != MIA_OK) return rc;
  if ((rc = upload(&p->d_win16, win16)) != MIA_OK) return rc;
  if ((rc = upload(&p->d_tw16, tw16)) != MIA_OK) return rc;
  if ((rc = upload(&p->d_sweep_meta, meta)) != MIA_OK) return rc;
  if ((rc = upload(&p->d_sweep_w, sw)) != MIA_OK) return rc;
  // Matrix-core projection (the A/B alternative to the sweep): the filterbank in blocks of 16
  // filters, each a 16 x K band of the matrix, K in steps of 4 bins (one
  // v_mfma_f32_16x16x4_f32 each).  Wave w projects frames 16 (w % 4) .. + 15 for the blocks of
  // group w / 4; the (at most 8) blocks are dealt to the 4 groups so that the steps balance.
  if (n_mels <= 128) {
    const int nmb = (n_mels + 15) / 16;
    struct Blk { int mb, klo, steps; };
    std::vector<Blk> blks;
    for (int mb = 0; mb < nmb; ++mb) {
      int lo = bins, hi = -1;
      for (int m = 16 * mb; m < std::min(n_mels, 16 * mb + 16); ++m) {
        for (int k = 0; k < bins; ++k) {
          if (basis[(size_t)m * bins + k] != 0.0f) lo = std::min(lo, k), hi = std::max(hi, k);
        }
      }
      blks.push_back({mb, hi < 0 ? 0 : lo, hi < 0 ? 0 : (hi - lo + 4) / 4});
    }
    std::sort(blks.begin(), blks.end(), [](const Blk& a, const Blk& b) { return a.steps > b.steps; });
    std::vector<int> mm(4 * 2 * 4, 0);
    for (int i = 0; i < 8; ++i) mm[(size_t)4 * i] = -1;
    int load[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
    std::vector<float> amat;
    for (const Blk& bk : blks) {
      int g = -1;
      for (int c = 0; c < 4; ++c) {
        if (cnt[c] < 2 && (g < 0 || load[c] < load[g])) g = c;
      }
      int* e = &mm[(size_t)(2 * g + cnt[g]) * 4];
      e[0] = bk.mb, e[1] = bk.klo, e[2] = bk.steps, e[3] = (int)(amat.size() / 64);
      ++cnt[g];
      load[g] += bk.steps;
      for (int st = 0; st < (bk.steps + 3) / 4 * 4; ++st) {  // whole chunks of 4 steps, zero-padded
        for (int lane = 0; lane < 64; ++lane) {  // A[m = lane % 16][k = lane / 16]
          const int m = 16 * bk.mb + lane % 16, k = bk.klo + 4 * st + lane / 16;
          amat.push_back(st < bk.steps && m < n_mels && k < bins ? basis[(size_t)m * bins + k] : 0.0f);
        }
      }
    }
    if (amat.empty()) amat.push_back(0.0f);
    if ((rc = upload(&p->d_mfma_meta, mm)) != MIA_OK) return rc;
    if ((rc = upload(&p->d_mfma_a, amat)) != MIA_OK) return rc;
  }
  return MIA_OK;
}

extern "C" {

int mia_plan_create(const mia_plan_desc* d, mia_plan** out) {
  MIA_REQUIRE(d != nullptr && out != nullptr, "null argument");
  *out = nullptr;
  MIA_REQUIRE(d->n_fft > 0, "n_fft must be positive, got %d", d->n_fft);
  MIA_REQUIRE(d->n_fft <= (1 << 24), "n_fft=%d is beyond the supported 2^24", d->n_fft);
  MIA_REQUIRE(d->hop_length > 0, "hop_length must be positive, got %d", d->hop_length);
  MIA_REQUIRE(d->window != nullptr, "window is required");
  MIA_REQUIRE(d->pad_mode >= 0 && d->pad_mode <= 2, "bad pad_mode %d", d->pad_mode);
  mia_plan* p = new mia_plan();
  p->generic = fft_size_supported(d->n_fft) ? 0 : 1;  // no compiled kernel: the any-length engine
  {
    int np = 0, radix[6];
    p->wave_fft = p->generic && forward_wave_plan(d->n_fft, &np, radix) ? 1 : 0;
  }
  int rc = MIA_OK;
  auto fail = [&](int code) {
    plan_free(p);
    return code;
  };
  if (hipGetDevice(&p->device) != hipSuccess) {
    set_error("hipGetDevice failed (no GPU?)");
    return fail(MIA_ERR_HIP);
  }
  if (hipDeviceGetAttribute(&p->cus, hipDeviceAttributeMultiprocessorCount, p->device) !=
          hipSuccess || p->cus <= 0) {
    p->cus = 256;
  }
  {
    void* keys = nullptr;  // the static clamp keys of this device (mia_melspectrogram_static)
    if ((rc = scratch_resolve(p->device, &keys)) != MIA_OK) return fail(rc);
  }
  p->n_fft = d->n_fft;
  p->hop = d->hop_length;
  p->center = d->center ? 1 : 0;
  p->pad_mode = d->pad_mode;
  p->power = d->power;
  const int N = d->n_fft, bins = N / 2 + 1;
  if (p->generic && (rc = generic_build_tables(N, &p->d_chirp, &p->d_bspec)) != MIA_OK) return fail(rc);
  {
    std::vector<float> w(d->window, d->window + N);
    if ((rc = upload(&p->d_window, w)) != MIA_OK) return fail(rc);
    std::vector<float2> tw((size_t)N);
    for (int k = 0; k < N; ++k) {
      // exact octant symmetry is not needed here; float64 libm then one rounding
      const double a = -2.0 * M_PI * double(k) / double(N);
      tw[(size_t)k] = {(float)std::cos(a), (float)std::sin(a)};
    }
    if ((rc = upload(&p->d_tw, tw)) != MIA_OK) return fail(rc);
  }
  if (d->n_mels > 0) {
    if (d->mel_basis == nullptr) {
      set_error("mel_basis is required when n_mels > 0");
      return fail(MIA_ERR_INVALID);
    }
    p->n_mels = d->n_mels;
    std::vector<int> lo((size_t)d->n_mels), len((size_t)d->n_mels), off((size_t)d->n_mels);
    std::vector<float> wts;
    for (int m = 0; m < d->n_mels; ++m) {
      const float* row = d->mel_basis + (size_t)m * bins;
      int first = -1, last = -1;
      for (int k = 0; k < bins; ++k) {
        if (row[k] != 0.0f) {
          if (first < 0) first = k;
          last = k;
        }
      }
      lo[(size_t)m] = first < 0 ? 0 : first;
      len[(size_t)m] = first < 0 ? 0 : last - first + 1;
      off[(size_t)m] = (int)wts.size();
      for (int k = 0; k < len[(size_t)m]; ++k) wts.push_back(row[first + k]);
      if (len[(size_t)m] > p->max_row_len) p->max_row_len = len[(size_t)m];
    }
    p->nnz = (int)wts.size();
    if ((rc = upload(&p->d_row_lo, lo)) != MIA_OK) return fail(rc);
    if ((rc = upload(&p->d_row_len, len)) != MIA_OK) return fail(rc);
    if ((rc = upload(&p->d_row_off, off)) != MIA_OK) return fail(rc);
    if ((rc = upload(&p->d_wts, wts)) != MIA_OK) return fail(rc);
    if (N == 400 && d->hop_length == 160 && (d->power == 2.0f || d->power == 1.0f)) {
      rc = build_w400_tables(p, d->window, d->mel_basis, d->n_mels, bins);
      if (rc != MIA_OK) return fail(rc);
      if (p->d_sweep_meta != nullptr) p->w400_bank = w400_compiled_bank(d->mel_basis, d->n_mels, bins);
    }
  }
  if (d->n_mfcc > 0) {
    if (d->dct_basis == nullptr || d->n_mels <= 0) {
      set_error("dct_basis and a mel stage are required when n_mfcc > 0");
      return fail(MIA_ERR_INVALID);
    }
    p->n_mfcc = d->n_mfcc;
    std::vector<float> c((size_t)d->n_mfcc * d->n_mels);
    for (int k = 0; k < d->n_mfcc; ++k) {
      const float l = d->lifter ? d->lifter[k] : 1.0f;
      for (int n = 0; n < d->n_mels; ++n) {
        c[(size_t)k * d->n_mels + n] = d->dct_basis[(size_t)k * d->n_mels + n] * l;
      }
    }
    if ((rc = upload(&p->d_dct, c)) != MIA_OK) return fail(rc);
  }
  *out = p;
  return MIA_OK;
}

int mia_plan_set_option(mia_plan* plan, const char* key, int value) {
  MIA_REQUIRE(plan != nullptr && key != nullptr, "null argument");
  if (strcmp(key, "w400_table_sweep") == 0) {
    plan->opt_table_sweep = value != 0;
    return MIA_OK;
  }
  if (strcmp(key, "w400_vgpr_staging") == 0) {
    plan->opt_vgpr_staging = value != 0;
    return MIA_OK;
  }
  if (strcmp(key, "w400_mfma_mel") == 0) {
    plan->opt_mfma_mel = value != 0;
    return MIA_OK;
  }
  if (strcmp(key, "w400_dpp_pass2") == 0) {
    plan->opt_dpp_pass2 = value != 0;
    return MIA_OK;
  }
  if (strcmp(key, "wave_fft") == 0) {  // 0: an any-length plan back on the whole-row engine (tests, A/B runs)
    int np = 0, radix[6];
    plan->wave_fft = value != 0 && plan->generic && forward_wave_plan(plan->n_fft, &np, radix) ? 1 : 0;
    return MIA_OK;
  }
  if (strcmp(key, "wave_istft_fused") == 0) {  // 0: inverse transform and overlap-add as two kernels again
    plan->opt_two_kernel_istft = value == 0;
    return MIA_OK;
  }
  if (strcmp(key, "fwd_lane") == 0) {
    plan->opt_lane = value != 0;
    return MIA_OK;
  }
  set_error("unknown plan option '%s'", key);
  return MIA_ERR_INVALID;
}

static bool plan_istft_fused(const mia_plan* plan);
int mia_plan_get_info(const mia_plan* plan, const char* key) {
  if (plan == nullptr || key == nullptr) return MIA_ERR_INVALID;
  if (strcmp(key, "w400_bank") == 0) return plan->w400_bank;
  if (strcmp(key, "w400_tables") == 0) return plan->d_sweep_meta != nullptr;
  if (strcmp(key, "w400_mfma_tables") == 0) return plan->d_mfma_a != nullptr;
  if (strcmp(key, "compute_units") == 0) return plan->cus;
  if (strcmp(key, "wave_fixed_plan") == 0) return plan->wave_fft ? forward_wave_fixed_plan(plan->n_fft) : -1;
  if (strcmp(key, "istft_fused") == 0) return plan_istft_fused(plan) ? 1 : 0;
  set_error("unknown plan info '%s'", key);
  return MIA_ERR_INVALID;
}

int mia_plan_destroy(mia_plan* plan) {
  plan_free(plan);
  return MIA_OK;
}

int64_t mia_plan_num_frames(const mia_plan* plan, int64_t T) {
  if (!plan) {
    set_error("null plan");
    return MIA_ERR_INVALID;
  }
  const int64_t padded = T + (plan->center ? 2 * (int64_t)(plan->n_fft / 2) : 0);
  if (padded < plan->n_fft) {
    set_error("Signal length (%lld) must be >= frame_length (%d). Consider padding the signal.",
              (long long)padded, plan->n_fft);
    return MIA_ERR_INVALID;
  }
  return 1 + (padded - plan->n_fft) / plan->hop;
}

// ----------------------------- hot path -------------------------------------

}  // extern "C"

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int fill_fwd(const mia_plan* plan, const float* y, int64_t B, int64_t T, FwdParams& p) {
  MIA_REQUIRE(plan != nullptr, "null plan");
  MIA_REQUIRE(y != nullptr || B == 0, "null input");
  MIA_REQUIRE(B >= 0 && T >= 0, "negative shape");
  const int64_t F = mia_plan_num_frames(plan, T);
  if (F < 0) return (int)F;
  const int pad = plan->center ? plan->n_fft / 2 : 0;
  if (plan->pad_mode == MIA_PAD_REFLECT && pad > 0) {
    MIA_REQUIRE(pad <= T - 1, "reflect padding of %d needs a signal longer than %d samples",
                pad, pad);
  }
  if (plan->pad_mode == MIA_PAD_EDGE && pad > 0) MIA_REQUIRE(T >= 1, "empty signal");
  std::memset(&p, 0, sizeof(p));
  p.y = y;
  p.B = B;
  p.T = T;
  p.F = F;
  p.hop = plan->hop;
  p.pad = pad;
  p.pad_mode = plan->pad_mode;
  p.win = plan->d_window;
  p.tw = plan->d_tw;
  p.scale = 1.0f;
#ifdef MIA_DEBUG_HOOKS
  { const char* e = getenv("MIA_DBG"); p.dbg = e ? atoi(e) : 0; }  // profiling builds only
#endif
  return MIA_OK;
}

extern "C" {

int mia_stft(const mia_plan* plan, const float* y, int64_t B, int64_t T, float scale, void* out,
             void* stream) {
  FwdParams p;
  int rc = fill_fwd(plan, y, B, T, p);
  if (rc != MIA_OK) return rc;
  MIA_REQUIRE(out != nullptr || B == 0, "null output");
  if (plan->generic && !plan->wave_fft) {
    set_error("n_fft=%d has no compiled kernel: call mia_stft_ws with mia_stft_workspace_bytes() "
              "bytes of workspace", plan->n_fft);
    return MIA_ERR_INVALID;
  }
  p.scale = scale;
  p.cout = reinterpret_cast<float2*>(out);
  if (plan->wave_fft) return launch_forward_wave(plan, p, 0, (hipStream_t)stream);
  return launch_forward(plan, p, 0, (hipStream_t)stream);
}

int mia_stft_moments(const mia_plan* plan, const float* y, int64_t B, int64_t T, float scale, void* out,
                     int group, int bins_counted, int64_t frame_lo, int64_t frame_hi, void* sums,
                     void* stream) {
  FwdParams p;
  int rc = fill_fwd(plan, y, B, T, p);
  if (rc != MIA_OK) return rc;
  MIA_REQUIRE(group >= 1 && B % group == 0, "the %lld rows do not split into groups of %d",
              (long long)B, group);
  MIA_REQUIRE(bins_counted >= 0 && bins_counted <= plan->n_fft / 2 + 1 && frame_lo >= 0 &&
                  frame_lo <= frame_hi,
              "bins_counted must be in [0, n_fft/2 + 1] and 0 <= frame_lo <= frame_hi");
  if (plan->generic) {
    set_error("n_fft=%d has no compiled kernel: run mia_stft_ws and mia_normalize_cac", plan->n_fft);
    return MIA_ERR_UNSUPPORTED;
  }
  if (B == 0) return MIA_OK;
  MIA_REQUIRE(out && sums, "null pointer");
  MIA_CHECK_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)(B / group), (hipStream_t)stream));
  p.scale = scale;
  p.cout = reinterpret_cast<float2*>(out);
  p.mom_sums = reinterpret_cast<double*>(sums);
  p.mom_group = group;
  p.mom_bins = bins_counted;
  p.mom_f_lo = frame_lo;
  p.mom_f_hi = frame_hi;
  return launch_forward(plan, p, 5, (hipStream_t)stream);
}

int mia_stft_energy_supported(const mia_plan* plan) {
  return plan && !plan->generic && !plan->wave_fft &&
                 (plan->n_fft == 1024 || plan->n_fft == 2048 || plan->n_fft == 4096)
             ? 1 : 0;
}

int mia_stft_energy(const mia_plan* plan, const float* y, int64_t B, int64_t T, float scale, void* out,
                    float* energy, void* stream) {
  FwdParams p;
  int rc = fill_fwd(plan, y, B, T, p);
  if (rc != MIA_OK) return rc;
  if (!mia_stft_energy_supported(plan)) {
    set_error("n_fft=%d has no spectrum + energy kernel (1024, 2048, 4096 have): run mia_stft and "
              "mia_frame_energy_spec", plan->n_fft);
    return MIA_ERR_UNSUPPORTED;
  }
  if (B == 0) return MIA_OK;
  MIA_REQUIRE(out && energy, "null pointer");
  p.scale = scale;
  p.cout = reinterpret_cast<float2*>(out);
  p.energy = energy;
  return launch_forward(plan, p, 7, (hipStream_t)stream);
}

size_t mia_stft_workspace_bytes(const mia_plan* plan, int64_t B, int64_t T) {
  if (!plan || !plan->generic || plan->wave_fft || B <= 0) return 0;
  const int64_t F = mia_plan_num_frames(plan, T);
  return F <= 0 ? 0 : generic_fft_bytes(plan->n_fft, B * F);
}

int mia_stft_ws(const mia_plan* plan, const float* y, int64_t B, int64_t T, float scale, void* out,
                void* workspace, size_t workspace_bytes, void* stream) {
  MIA_REQUIRE(plan != nullptr, "null plan");
  if (!plan->generic || plan->wave_fft) return mia_stft(plan, y, B, T, scale, out, stream);
  FwdParams p;
  int rc = fill_fwd(plan, y, B, T, p);
  if (rc != MIA_OK) return rc;
  MIA_REQUIRE(out != nullptr || B == 0, "null output");
  p.scale = scale;
  p.cout = reinterpret_cast<float2*>(out);
  return launch_forward_generic(plan, p, 0, workspace, workspace_bytes, (hipStream_t)stream);
}

int mia_irfft_frames(const mia_plan* plan, const void* spec, int64_t B, int64_t F,
                     int64_t stride_b, int64_t stride_bin, int64_t stride_frame, float* frames,
                     void* stream) {
  MIA_REQUIRE(plan != nullptr, "null plan");
  MIA_REQUIRE(B >= 0 && F >= 0, "negative shape");
  MIA_REQUIRE((spec && frames) || B == 0 || F == 0, "null buffer");
  if (plan->generic && !plan->wave_fft) {
    set_error("n_fft=%d has no compiled kernel: use mia_istft (it takes the workspace the "
              "any-length engine needs)", plan->n_fft);
    return MIA_ERR_UNSUPPORTED;
  }
  InvParams p{};
  p.spec = reinterpret_cast<const float2*>(spec);
  p.B = B;
  p.F = F;
  p.sb = stride_b;
  p.sk = stride_bin;
  p.sf = stride_frame;
  p.tw = plan->d_tw;
  p.frames = frames;
  if (plan->wave_fft) return launch_inverse_wave(plan, p, (hipStream_t)stream);
  return launch_inverse(plan, p, (hipStream_t)stream);
}

int mia_overlap_add(const float* frames, const float* window, int64_t B, int64_t F, int n_fft,
                    int hop, int64_t out_len, float scale, float* out, void* stream) {
  MIA_REQUIRE(n_fft > 0, "n_fft must be positive, got %d", n_fft);
  MIA_REQUIRE(hop > 0, "hop_length must be positive, got %d", hop);
  MIA_REQUIRE(B >= 0 && F >= 0 && out_len >= 0, "negative shape");
  MIA_REQUIRE((frames && window && out) || B == 0 || out_len == 0, "null buffer");
  for (int64_t b0 = 0; b0 < B; b0 += kMaxGridRows) {  // a launch takes at most 65535 rows
    const int64_t nb = B - b0 < kMaxGridRows ? B - b0 : kMaxGridRows;
    const int rc = launch_overlap_add(frames + (size_t)b0 * F * n_fft, window, nb, F, n_fft, hop,
                                      out_len, scale, out + (size_t)b0 * out_len,
                                      (hipStream_t)stream);
    if (rc != MIA_OK) return rc;
  }
  return MIA_OK;
}

// Inverse transform and overlap-add in one launch for this plan's framing?
static bool plan_istft_fused(const mia_plan* plan) {
  if (mia::istft_fused_supported(plan->n_fft, plan->hop)) {
    return !(plan->wave_fft && plan->opt_two_kernel_istft);
  }
  // the small compiled sizes: their inverse on the per-wave transform's fused kernel
  return !plan->generic && !plan->opt_two_kernel_istft &&
         mia::istft_wave_fused_supported(plan->n_fft, plan->hop);
}

size_t mia_istft_workspace_bytes(const mia_plan* plan, int64_t B, int64_t F) {
  if (!plan || B <= 0 || F <= 0) return 0;
  if (plan->generic && !plan->wave_fft) {  // frames + the any-length engine's row buffers
    return align256(sizeof(float) * (size_t)B * (size_t)F * (size_t)plan->n_fft) +
           generic_fft_bytes(plan->n_fft, B * F);
  }
  if (plan_istft_fused(plan)) return 0;  // overlap kept in LDS
  return sizeof(float) * (size_t)B * (size_t)F * (size_t)plan->n_fft;
}

int mia_istft(const mia_plan* plan, const void* spec, int64_t B, int64_t F, int64_t stride_b,
              int64_t stride_bin, int64_t stride_frame, int64_t out_len, float scale, float* out,
              void* workspace, size_t workspace_bytes, void* stream) {
  MIA_REQUIRE(plan != nullptr, "null plan");
  if (plan->generic && !plan->wave_fft) {
    MIA_REQUIRE(B >= 0 && F >= 0 && out_len >= 0, "negative shape");
    MIA_REQUIRE((spec && out) || B == 0 || F == 0 || out_len == 0, "null buffer");
    if (B == 0 || out_len == 0) return MIA_OK;
    MIA_REQUIRE(workspace_bytes >= mia_istft_workspace_bytes(plan, B, F),
                "workspace too small: %zu < %zu", workspace_bytes,
                mia_istft_workspace_bytes(plan, B, F));
    float* frames = reinterpret_cast<float*>(workspace);
    const size_t fbytes = align256(sizeof(float) * (size_t)B * (size_t)F * (size_t)plan->n_fft);
    InvParams p{};
    p.spec = reinterpret_cast<const float2*>(spec);
    p.B = B;
    p.F = F;
    p.sb = stride_b;
    p.sk = stride_bin;
    p.sf = stride_frame;
    p.frames = frames;
    int rc = launch_inverse_generic(plan, p, reinterpret_cast<char*>(workspace) + fbytes,
                                    workspace_bytes - fbytes, (hipStream_t)stream);
    if (rc != MIA_OK) return rc;
    return mia_overlap_add(frames, plan->d_window, B, F, plan->n_fft, plan->hop, out_len, scale,
                           out, stream);
  }
  if (plan_istft_fused(plan)) {
    MIA_REQUIRE(B >= 0 && F >= 0 && out_len >= 0, "negative shape");
    MIA_REQUIRE((spec && out) || B == 0 || F == 0 || out_len == 0, "null buffer");
    if (B == 0 || out_len == 0) return MIA_OK;
    if (F == 0) {
      MIA_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * (size_t)out_len,
                                   (hipStream_t)stream));
      return MIA_OK;
    }
    InvParams p{};
    p.spec = reinterpret_cast<const float2*>(spec);
    p.B = B;
    p.F = F;
    p.sb = stride_b;
    p.sk = stride_bin;
    p.sf = stride_frame;
    p.tw = plan->d_tw;
    p.win = plan->d_window;
    p.hop = plan->hop;
    p.out_len = out_len;
    p.scale = scale;
    p.out = out;
    if (mia::istft_fused_supported(plan->n_fft, plan->hop)) {
      return launch_istft_fused(plan, p, (hipStream_t)stream);
    }
    return launch_istft_wave_fused(plan, p, (hipStream_t)stream);  // a compiled size on an inverse-only plan
  }
  MIA_REQUIRE(workspace_bytes >= mia_istft_workspace_bytes(plan, B, F),
              "workspace too small: %zu < %zu", workspace_bytes,
              mia_istft_workspace_bytes(plan, B, F));
  float* frames = reinterpret_cast<float*>(workspace);
  int rc = mia_irfft_frames(plan, spec, B, F, stride_b, stride_bin, stride_frame, frames, stream);
  if (rc != MIA_OK) return rc;
  return mia_overlap_add(frames, plan->d_window, B, F, plan->n_fft, plan->hop, out_len, scale,
                         out, stream);
}

int mia_istft_gated_supported(const mia_plan* plan, int size_freq, int size_freq2, int size_time) {
  return plan && !plan->generic && !plan->wave_fft &&
                 mia::istft_gated_supported(plan->n_fft, plan->hop, size_freq, size_freq2, size_time)
             ? 1 : 0;
}

int mia_istft_gated(const mia_plan* plan, const void* spec, int64_t B, int64_t F, int64_t out_len,
                    float scale, const float* thr, int64_t rows_thr, float gain, int keep,
                    float prop_decrease, int size_freq, int size_freq2, int size_time, float* out,
                    void* stream) {
  MIA_REQUIRE(plan != nullptr, "null plan");
  MIA_REQUIRE(B >= 0 && F >= 0 && out_len >= 0, "negative shape");
  MIA_REQUIRE(F < (1LL << 31), "too many frames per row: %lld", (long long)F);
  MIA_REQUIRE(rows_thr == 1 || rows_thr == B, "thresholds for %lld rows, the spectrum has %lld",
              (long long)rows_thr, (long long)B);
  if (!mia_istft_gated_supported(plan, size_freq, size_freq2, size_time)) {
    set_error("no gated inverse for n_fft=%d hop=%d and windows of %d / %d bins, %d frames",
              plan->n_fft, plan->hop, size_freq, size_freq2, size_time);
    return MIA_ERR_UNSUPPORTED;
  }
  MIA_REQUIRE((spec && out && thr) || B == 0 || F == 0 || out_len == 0, "null buffer");
  if (B == 0 || out_len == 0) return MIA_OK;
  if (F == 0) {
    MIA_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * (size_t)out_len,
                                 (hipStream_t)stream));
    return MIA_OK;
  }
  const long long bins = plan->n_fft / 2 + 1;
  InvParams p{};
  p.spec = reinterpret_cast<const float2*>(spec);
  p.B = B;
  p.F = F;
  p.sb = F * bins;
  p.sk = 1;
  p.sf = bins;
  p.tw = plan->d_tw;
  p.win = plan->d_window;
  p.hop = plan->hop;
  p.out_len = out_len;
  p.scale = scale;
  p.out = out;
  GateParams g{};
  g.thr = thr;
  g.thr_stride = rows_thr == 1 ? 0 : bins;
  g.gain = gain;
  g.prop = prop_decrease;
  g.keep = keep ? 1 : 0;
  g.s1 = size_freq;
  g.s2 = size_freq2;
  g.st = size_time;
  return launch_istft_gated(plan, p, g, (hipStream_t)stream);
}

}  // extern "C"

extern "C" size_t mia_mel_workspace_bytes(const mia_plan* plan, int64_t B, int64_t T) {
  if (!plan || B <= 0) return 0;
  const int64_t F = mia_plan_num_frames(plan, T);
  if (F <= 0) return 0;
  const size_t keys = 2 * align256(sizeof(int) * (size_t)(B + 1));
  return plan->generic && !plan->wave_fft ? keys + generic_fft_bytes(plan->n_fft, B * F) : keys;
}

// The n_fft = 400 kernel when the plan has its tables (and a clip's output fits the
// kernel's 32-bit offsets), else the generic one.
static int forward_mel_dispatch(const mia_plan* plan, const FwdParams& p, hipStream_t s,
                                void* fft_ws = nullptr, size_t fft_ws_bytes = 0) {
  if (plan->wave_fft) return launch_forward_wave(plan, p, 1, s);
  if (plan->generic) return launch_forward_generic(plan, p, 1, fft_ws, fft_ws_bytes, s);
  // (magnitudes instead of powers: the n_fft = 400 kernel's linear and dB epilogues take them, its
  // Whisper epilogue does not)
  if (plan->d_sweep_meta != nullptr && p.n_mels * p.F < (1LL << 30) &&
      (p.power_mode == 0 || (p.power_mode == 1 && p.epi != MIA_EPI_WHISPER))) {
    W400Tables tb{plan->d_win16, plan->d_tw16, plan->d_winp, plan->d_sweep_meta, plan->d_sweep_w,
                  plan->w400_bank, plan->opt_table_sweep, plan->opt_vgpr_staging, plan->cus,
                  plan->d_mfma_meta, plan->d_mfma_a, plan->opt_mfma_mel, plan->opt_dpp_pass2};
    return launch_logmel400(p, tb, s);
  }
  return launch_forward(plan, p, 1, s);
}

static bool mel_frame_major_ok(const mia_plan* plan) {
  if (plan->wave_fft) return forward_wave_frame_major_ok(plan->n_fft, plan->n_mels);
  return !plan->generic && plan->d_sweep_meta == nullptr &&
         forward_frame_major_ok(plan->n_fft, plan->n_mels);
}

static int mel_forward(const mia_plan* plan, const float* y, int64_t B, int64_t T, int epilogue,
                       float ref, float amin, float coef, float top_db, int per_clip, float* out,
                       void* workspace, size_t workspace_bytes, hipStream_t s, int frame_major = 0) {
  FwdParams p;
  int rc = fill_fwd(plan, y, B, T, p);
  if (rc != MIA_OK) return rc;
  MIA_REQUIRE(plan->n_mels > 0, "plan has no mel stage");
  if (frame_major && !mel_frame_major_ok(plan)) {
    set_error("this plan writes (B, n_mels, F) only (frame-major rows need one of the compiled "
              "generic kernels and n_mels %% 4 == 0)");
    return MIA_ERR_UNSUPPORTED;
  }
  p.frame_major = frame_major ? 1 : 0;
  MIA_REQUIRE(out != nullptr || B == 0, "null output");
  MIA_REQUIRE(epilogue >= 0 && epilogue <= 2, "bad epilogue %d", epilogue);
  p.n_mels = plan->n_mels;
  p.row_lo = plan->d_row_lo;
  p.row_len = plan->d_row_len;
  p.row_off = plan->d_row_off;
  p.wts = plan->d_wts;
  p.nnz = plan->nnz;
  p.lds_tables = 0;
  p.power = plan->power;
  p.power_mode = plan->power == 2.0f ? 0 : (plan->power == 1.0f ? 1 : 2);
  p.mout = out;
  p.epi = epilogue;
  p.amin = amin;
  p.inv_ref = 1.0f / fmaxf(ref, amin);
  p.coef = coef;
  p.db_floor = coef * log10f(amin / fmaxf(ref, amin));
  p.db_floor_shift = coef * log10f(1.0f / fmaxf(ref, amin));
  const bool clamp = epilogue == MIA_EPI_WHISPER || (epilogue == MIA_EPI_DB && top_db >= 0.0f);
  int* max_keys = nullptr;
  int* min_keys = nullptr;
  if (clamp && B > 0) {
    MIA_REQUIRE(workspace != nullptr && workspace_bytes >= mia_mel_workspace_bytes(plan, B, T),
                "workspace too small");
    max_keys = reinterpret_cast<int*>(workspace);
    min_keys = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) +
                                      align256(sizeof(int) * (size_t)(B + 1)));
    if ((rc = launch_init_keys(max_keys, min_keys, B + 1, s)) != MIA_OK) return rc;
    p.clip_max_key = max_keys;
    p.clip_min_key = min_keys;
  }
#ifdef MIA_DEBUG_HOOKS
  { const char* e = getenv("MIA_DBG"); p.dbg = e ? atoi(e) : 0; }  // profiling builds only
#endif
  const size_t keys_bytes = 2 * align256(sizeof(int) * (size_t)(B + 1));
  if (plan->generic && !plan->wave_fft && B > 0) {
    MIA_REQUIRE(workspace != nullptr && workspace_bytes >= mia_mel_workspace_bytes(plan, B, T),
                "workspace too small (any-length plans need mia_mel_workspace_bytes() bytes)");
  }
  rc = forward_mel_dispatch(plan, p, s, reinterpret_cast<char*>(workspace) + keys_bytes,
                            workspace_bytes > keys_bytes ? workspace_bytes - keys_bytes : 0);
  if (rc != MIA_OK) return rc;
#ifdef MIA_DEBUG_HOOKS
  if (p.dbg != 0) return rc;  // ablation runs time the main kernel alone (their data is garbage)
#endif
  if (clamp && B > 0) {
    const float top = epilogue == MIA_EPI_WHISPER ? 8.0f : top_db;
    rc = launch_floor_clips(out, B, (long long)plan->n_mels * p.F, max_keys, min_keys, epilogue,
                            top, per_clip, s);
  }
  return rc;
}

extern "C" {

int mia_melspectrogram(const mia_plan* plan, const float* y, int64_t B, int64_t T, int epilogue,
                       float ref, float amin, float coef, float top_db, int per_clip, float* out,
                       void* workspace, size_t workspace_bytes, void* stream) {
  return mel_forward(plan, y, B, T, epilogue, ref, amin, coef, top_db, per_clip, out, workspace,
                     workspace_bytes, (hipStream_t)stream);
}

int mia_mel_frames_supported(const mia_plan* plan) {
  return plan != nullptr && plan->n_mels > 0 && mel_frame_major_ok(plan) ? 1 : 0;
}

int mia_melspectrogram_frames(const mia_plan* plan, const float* y, int64_t B, int64_t T, int epilogue,
                              float ref, float amin, float coef, float top_db, int per_clip, float* out,
                              void* workspace, size_t workspace_bytes, void* stream) {
  return mel_forward(plan, y, B, T, epilogue, ref, amin, coef, top_db, per_clip, out, workspace,
                     workspace_bytes, (hipStream_t)stream, 1);
}

int mia_melspectrogram_static(const mia_plan* plan, const float* y, int64_t B, int64_t T,
                              int epilogue, float ref, float amin, float coef, float top_db,
                              int per_clip, float* out, void* stream) {
  MIA_REQUIRE(plan != nullptr, "null plan");
  MIA_REQUIRE(!plan->generic || plan->wave_fft, "n_fft=%d has no compiled kernel: its any-length plan needs the "
              "workspace of mia_melspectrogram", plan->n_fft);
  MIA_REQUIRE(B <= MIA_STATIC_MAX_CLIPS,
              "mia_melspectrogram_static holds clamp keys for at most %d clips, got %lld "
              "(use mia_melspectrogram with a workspace)", MIA_STATIC_MAX_CLIPS, (long long)B);
  void* keys = nullptr;
  int rc = scratch_resolve(plan->device, &keys);  // a table lookup after mia_plan_create
  if (rc != MIA_OK) return rc;
  static_assert(2 * kStaticKeyBytes == sizeof(scratch), "both key arrays fit");
  return mel_forward(plan, y, B, T, epilogue, ref, amin, coef, top_db, per_clip, out, keys,
                     2 * kStaticKeyBytes, (hipStream_t)stream);
}

int mia_spectral_stats(const mia_plan* plan, const float* y, int64_t B, int64_t T, const float* freq,
                       float p_exp, int norm, float roll_percent, float flat_power, float flat_amin,
                       float* out, void* stream) {
  FwdParams p;
  int rc = fill_fwd(plan, y, B, T, p);
  if (rc != MIA_OK) return rc;
  MIA_REQUIRE(freq != nullptr && (out != nullptr || B == 0), "null pointer");
  MIA_REQUIRE(p_exp > 0.0f && roll_percent >= 0.0f && roll_percent <= 1.0f,
              "p must be > 0 and roll_percent in [0, 1]");
  p.n_mels = 4;  // rows of the output per clip
  p.power_mode = 1;  // magnitudes in the LDS tile
  p.power = 1.0f;
  p.mout = out;
  p.epi = kEpiStats;
  p.freq = freq;
  p.stat_p = p_exp;
  p.stat_norm = norm;
  p.stat_roll = roll_percent;
  p.flat_power = flat_power;
  p.flat_amin = flat_amin;
  return launch_forward(plan, p, 2, (hipStream_t)stream);
}

size_t mia_mfcc_workspace_bytes(const mia_plan* plan, int64_t B, int64_t T) {
  if (!plan || B <= 0) return 0;
  const int64_t F = mia_plan_num_frames(plan, T);
  if (F <= 0) return 0;
  return align256(mia_mel_workspace_bytes(plan, B, T)) +
         align256(sizeof(float) * (size_t)B * (size_t)plan->n_mels * (size_t)F);
}

int mia_mfcc(const mia_plan* plan, const float* y, int64_t B, int64_t T, float top_db,
             int per_clip, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  MIA_REQUIRE(plan != nullptr && plan->n_mfcc > 0, "plan has no DCT stage");
  MIA_REQUIRE(B >= 0, "batch %lld out of range", (long long)B);
  if (B == 0) return MIA_OK;
  MIA_REQUIRE(workspace != nullptr && workspace_bytes >= mia_mfcc_workspace_bytes(plan, B, T),
              "workspace too small");
  if (B > kMaxGridRows) {
    // Clips are independent when the maximum is per clip: the batch goes through in pieces
    // that fit one launch, each in the same workspace (sized for the whole batch).
    MIA_REQUIRE(per_clip || top_db < 0.0f,
                "a batch-wide top_db maximum over %lld clips needs per_clip != 0 (at most 65535 "
                "clips share one maximum)", (long long)B);
    const int64_t Fq = mia_plan_num_frames(plan, T);
    if (Fq < 0) return (int)Fq;
    for (int64_t b0 = 0; b0 < B; b0 += kMaxGridRows) {
      const int64_t nb = B - b0 < kMaxGridRows ? B - b0 : kMaxGridRows;
      const int rc = mia_mfcc(plan, y + (size_t)b0 * T, nb, T, top_db, per_clip,
                              out + (size_t)b0 * plan->n_mfcc * Fq, workspace, workspace_bytes,
                              stream);
      if (rc != MIA_OK) return rc;
    }
    return MIA_OK;
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t melws = align256(mia_mel_workspace_bytes(plan, B, T));
  float* sdb = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + melws);
  const int64_t F = mia_plan_num_frames(plan, T);
  if (F < 0) return (int)F;
  // dB without the floor; the floor is applied while the DCT stages its slab.
  FwdParams p;
  int rc = fill_fwd(plan, y, B, T, p);
  if (rc != MIA_OK) return rc;
  p.n_mels = plan->n_mels;
  p.row_lo = plan->d_row_lo;
  p.row_len = plan->d_row_len;
  p.row_off = plan->d_row_off;
  p.wts = plan->d_wts;
  p.nnz = plan->nnz;
  p.lds_tables = 0;
  p.power = plan->power;
  p.power_mode = plan->power == 2.0f ? 0 : (plan->power == 1.0f ? 1 : 2);
  p.mout = sdb;
  p.epi = MIA_EPI_DB;
  p.amin = 1e-10f;
  p.inv_ref = 1.0f;
  p.coef = 10.0f;
  p.db_floor = 10.0f * log10f(1e-10f);
  int* keys = reinterpret_cast<int*>(workspace);
  int* min_keys = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) +
                                         align256(sizeof(int) * (size_t)(B + 1)));
  const bool clamp = top_db >= 0.0f;
  if ((rc = launch_init_keys(keys, min_keys, B + 1, s)) != MIA_OK) return rc;
  p.clip_max_key = keys;
  p.clip_min_key = min_keys;
  // The dB values are an intermediate of this call, so their layout is free: when the generic
  // mel kernel and the matrix-core DCT serve the plan, they go frame-major - a tile of a few
  // frames then leaves as one contiguous run of whole lines instead of n_mels short row
  // fragments (which reach memory as read-modify-write pairs), and the DCT reads whole runs.
  const bool fm = mel_frame_major_ok(plan) && dct2_frame_major_ok(plan->n_mels, plan->n_mfcc);
  p.frame_major = fm ? 1 : 0;
  {
    const size_t keys_bytes = 2 * align256(sizeof(int) * (size_t)(B + 1));
    rc = forward_mel_dispatch(plan, p, s, reinterpret_cast<char*>(workspace) + keys_bytes,
                              melws > keys_bytes ? melws - keys_bytes : 0);
  }
  if (rc != MIA_OK) return rc;
  if (clamp && !per_clip) {
    // global maximum: fold keys[0..B) into keys[B]
    rc = launch_reduce_keys(keys, B, s);
    if (rc != MIA_OK) return rc;
  }
  if (fm) {
    return launch_dct2_frame_major(sdb, B, plan->n_mels, F, plan->d_dct, plan->n_mfcc, out,
                                   clamp ? (per_clip ? keys : keys + B) : nullptr,
                                   (long long)(per_clip ? 1 : 0), top_db, s);
  }
  return launch_dct2(sdb, B, plan->n_mels, F, plan->d_dct, plan->n_mfcc, out,
                     clamp ? (per_clip ? keys : keys + B) : nullptr,
                     (long long)(per_clip ? 1 : 0), top_db, s);
}

// --------------------------- stand-alone ops ---------------------------------

int mia_pad_signal(const float* x, int64_t B, int64_t T, int64_t pad, int mode, float* out,
                   void* stream) {
  MIA_REQUIRE(B >= 0 && T >= 0, "negative shape");
  MIA_REQUIRE(pad >= 0, "pad_length must be non-negative, got %lld", (long long)pad);
  MIA_REQUIRE(mode >= 0 && mode <= 2, "Unknown pad_mode code %d", mode);
  if (mode == MIA_PAD_REFLECT && pad > 0) {
    MIA_REQUIRE(pad <= T - 1, "reflect padding of %lld needs a signal longer than %lld samples",
                (long long)pad, (long long)pad);
  }
  if (mode == MIA_PAD_EDGE && pad > 0) MIA_REQUIRE(T >= 1, "empty signal");
  for (int64_t b0 = 0; b0 < B; b0 += kMaxGridRows) {
    const int64_t nb = B - b0 < kMaxGridRows ? B - b0 : kMaxGridRows;
    const int rc = launch_pad(x + (size_t)b0 * T, nb, T, pad, mode, out + (size_t)b0 * (T + 2 * pad),
                              (hipStream_t)stream);
    if (rc != MIA_OK) return rc;
  }
  return MIA_OK;
}

int mia_frame_signal(const float* x, int64_t B, int64_t T, int frame_length, int hop, float* out,
                     void* stream) {
  MIA_REQUIRE(frame_length > 0, "frame_length must be positive, got %d", frame_length);
  MIA_REQUIRE(hop > 0, "hop_length must be positive, got %d", hop);
  MIA_REQUIRE(T >= frame_length,
              "Signal length (%lld) must be >= frame_length (%d). Consider padding the signal.",
              (long long)T, frame_length);
  MIA_REQUIRE(B >= 0, "batch %lld out of range", (long long)B);
  const int64_t F = 1 + (T - frame_length) / hop;
  for (int64_t b0 = 0; b0 < B; b0 += kMaxGridRows) {
    const int64_t nb = B - b0 < kMaxGridRows ? B - b0 : kMaxGridRows;
    const int rc = launch_frame(x + (size_t)b0 * T, nb, T, F, frame_length, hop,
                                out + (size_t)b0 * F * frame_length, (hipStream_t)stream);
    if (rc != MIA_OK) return rc;
  }
  return MIA_OK;
}

int mia_to_db(const float* x, int64_t B, int64_t n, float ref, float amin, float coef,
              float top_db, int per_clip, float* out, void* workspace, void* stream) {
  MIA_REQUIRE(B >= 0 && n >= 0, "negative shape");
  MIA_REQUIRE(top_db < 0.0f || workspace != nullptr, "workspace required for top_db");
  MIA_REQUIRE(B <= kMaxGridRows || per_clip || top_db < 0.0f,
              "one top_db maximum over %lld rows needs per_clip != 0 (at most 65535 rows share a "
              "maximum)", (long long)B);
  for (int64_t b0 = 0; b0 < B; b0 += kMaxGridRows) {  // rows are independent from here on
    const int64_t nb = B - b0 < kMaxGridRows ? B - b0 : kMaxGridRows;
    const int rc = launch_to_db(x + (size_t)b0 * n, nb, n, ref, amin, coef, top_db, per_clip,
                                out + (size_t)b0 * n, reinterpret_cast<int*>(workspace),
                                (hipStream_t)stream);
    if (rc != MIA_OK) return rc;
  }
  return MIA_OK;
}

int mia_from_db(const float* x, int64_t n, float ref, float divisor, float* out, void* stream) {
  MIA_REQUIRE(n >= 0, "negative length");
  MIA_REQUIRE(divisor != 0.0f, "divisor must not be 0");
  if (n == 0) return MIA_OK;
  MIA_REQUIRE(x && out, "null pointer");
  return launch_from_db(x, n, ref, divisor, out, (hipStream_t)stream);
}

int mia_dct2(const float* x, int64_t outer, int n_in, int64_t inner, const float* basis,
             int n_out, float* out, void* stream) {
  MIA_REQUIRE(outer >= 0 && inner >= 0 && n_in > 0 && n_out > 0, "bad shape");
  for (int64_t o0 = 0; o0 < outer; o0 += kMaxGridRows) {
    const int64_t no = outer - o0 < kMaxGridRows ? outer - o0 : kMaxGridRows;
    const int rc = launch_dct2(x + (size_t)o0 * n_in * inner, no, n_in, inner, basis, n_out,
                               out + (size_t)o0 * n_out * inner, nullptr, 0, 0.0f,
                               (hipStream_t)stream);
    if (rc != MIA_OK) return rc;
  }
  return MIA_OK;
}

int mia_magnitude(const void* z, int64_t n, float* out, void* stream) {
  return launch_magnitude(reinterpret_cast<const float2*>(z), n, out, (hipStream_t)stream);
}
int mia_phase(const void* z, int64_t n, float* out, void* stream) {
  return launch_phase(reinterpret_cast<const float2*>(z), n, out, (hipStream_t)stream);
}

}  // extern "C"

// counter.hip -- the k-mer database operations of KPopCountDB (lib/KMerDB.ml) on the GPU:
//   * stats_table_of_core_db      (:171-271)  per-spectrum and per-k-mer statistics under a threshold/power,
//   * add_combined_selected       (:628-736)  rescaled mean / median of a set of spectra (class representatives),
//   * Transformation.compute      (:96-144)   the table transformations of -t / -s.
//
// Storage is the reference's `storage: I32BAVector.t array`: one int32 vector per spectrum ("column"), n_rows
// k-mers each.  On the device: [n_cols][ld] int32, ld = n_rows rounded up to 32 so every column starts on a
// 128-byte line.  Everything here is a stream over that array (4 bytes per count): HBM-bound byte/integer work,
// lanes along k-mers so every load is a full line.
//
// Summation order: the reference adds sequentially (k-mer order for a column, spectrum order for a row).  Row-wise
// quantities are computed by one thread per k-mer in exactly that order.  Column-wise sums are block-tree
// reductions with an ordered final pass: identical to the sequential sum whenever the terms are integers below
// 2^53 (power = 1, the default everywhere on the training path), within rounding otherwise.
#include <math.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "wave_sort.h"

namespace kpop {

namespace {

constexpr int kStatBlock = 256;
constexpr uint64_t kRowsPerStatBlock = 1u << 16;

struct ColPartial {
  double non_zero, max, sum, sum_log;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}

// x ** y for the non-negative operands of this file.  The device pow() is within an ulp but does not return the
// exact value where one exists (pow(3, 1) != 3), and Transformation.compute floors what it computes: the powers a
// user actually passes (0, 1/2, 1, 2, 3) take an exact route, the rest go through pow().
__device__ __forceinline__ double pow_nn(double x, double y) {
  if (y == 1.) return x;
  if (y == 0.) return 1.;
  if (y == 2.) return x * x;
  if (y == 3.) return x * x * x;
  if (y == 0.5) return sqrt(x);
  return pow(x, y);
}
__device__ __forceinline__ double pow_count(double f, double power, bool power_one) { return power_one ? f : pow_nn(f, power); }

// sum of log(count): accumulated as a running product with the exponent split off (one multiply per count
// instead of one log()); the product of n factors carries a relative error of ~sqrt(n) ulp, far below what a sum of
// n rounded logarithms carries.  A zero factor makes the product 0 and the sum -inf, as log 0 does.
struct LogSum {
  double mant = 1.;
  int64_t expo = 0;
  __device__ __forceinline__ void mul(double f) { mant *= f; }
  // call at least once every 8 factors (each < 2^32): 2^700 * 2^256 stays finite
  __device__ __forceinline__ void renorm() {
    if (mant > 0x1p+700) {
      int ex;
      mant = frexp(mant, &ex);
      expo += ex;
    }
  }
  __device__ __forceinline__ double value() const { return log(mant) + (double)expo * 0.693147180559945309417232121458; }
};

constexpr int kStatUnroll = 8;

// pass A (only for relative thresholds): plain sum of v^power per column slab
__global__ __launch_bounds__(kStatBlock) void col_plain_sum_kernel(const int32_t *__restrict__ storage, uint64_t ld,
                                                                  uint64_t n_rows, double power, int power_one,
                                                                  uint32_t n_slabs, double *__restrict__ partial) {
  const uint32_t col = blockIdx.y, slab = blockIdx.x;
  const int32_t *v = storage + (uint64_t)col * ld;
  const uint64_t lo = (uint64_t)slab * kRowsPerStatBlock, hi = min(n_rows, lo + kRowsPerStatBlock);
  double s = 0.;
  uint64_t i = lo + threadIdx.x;
  for (; i + (kStatUnroll - 1) * kStatBlock < hi; i += kStatUnroll * kStatBlock) {
    int32_t c[kStatUnroll];
#pragma unroll
    for (int u = 0; u < kStatUnroll; ++u) c[u] = v[i + u * kStatBlock];
#pragma unroll
    for (int u = 0; u < kStatUnroll; ++u) s += pow_count((double)c[u], power, power_one);
  }
  for (; i < hi; i += kStatBlock) s += pow_count((double)v[i], power, power_one);
  __shared__ double sh[kStatBlock / 64];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.;
    for (int w = 0; w < kStatBlock / 64; ++w) t += sh[w];
    partial[(uint64_t)col * n_slabs + slab] = t;
  }
}

// thr[c] = threshold < 1 ? threshold * sum_c : threshold       (lib/KMerDB.ml:190-195)
__global__ void col_threshold_kernel(const double *__restrict__ partial, uint32_t n_slabs, uint32_t n_cols, double threshold,
                                     double *__restrict__ thr) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  double t = threshold;
  if (threshold < 1.) {
    double s = 0.;
    for (uint32_t k = 0; k < n_slabs; ++k) s += partial[(uint64_t)c * n_slabs + k];
    t = threshold * s;
  }
  thr[c] = t;
}

struct StatAcc {
  double nz = 0., mx = 0., s = 0.;
  LogSum ls;
  __device__ __forceinline__ void add(int32_t c, double threshold, double power, bool power_one) {
    const double f = (double)c;
    if (f >= threshold) {
      nz += 1.;
      mx = fmax(mx, f);
      s += pow_count(f, power, power_one);
      ls.mul(f);
    }
  }
};

// pass B: thresholded statistics per column slab (lib/KMerDB.ml:196-215); sum_log is scaled by `power` at the end
__global__ __launch_bounds__(kStatBlock) void col_stats_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint64_t n_rows,
                                                              double power, int power_one, const double *__restrict__ thr,
                                                              uint32_t n_slabs, ColPartial *__restrict__ partial) {
  const uint32_t col = blockIdx.y, slab = blockIdx.x;
  const int32_t *v = storage + (uint64_t)col * ld;
  const uint64_t lo = (uint64_t)slab * kRowsPerStatBlock, hi = min(n_rows, lo + kRowsPerStatBlock);
  const double threshold = thr[col];
  StatAcc a;
  uint64_t i = lo + threadIdx.x;
  for (; i + (kStatUnroll - 1) * kStatBlock < hi; i += kStatUnroll * kStatBlock) {
    int32_t c[kStatUnroll];
#pragma unroll
    for (int u = 0; u < kStatUnroll; ++u) c[u] = __builtin_nontemporal_load(v + i + u * kStatBlock);
#pragma unroll
    for (int u = 0; u < kStatUnroll; ++u) a.add(c[u], threshold, power, power_one);
    a.ls.renorm();
  }
  for (; i < hi; i += kStatBlock) {
    a.add(v[i], threshold, power, power_one);
    a.ls.renorm();
  }
  __shared__ ColPartial sh[kStatBlock / 64];
  const double nz = wave_sum(a.nz), mx = wave_max(a.mx), s = wave_sum(a.s), sl = wave_sum(a.ls.value());
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ColPartial{nz, mx, s, sl};
  __syncthreads();
  if (threadIdx.x == 0) {
    ColPartial t{0., 0., 0., 0.};
    for (int w = 0; w < kStatBlock / 64; ++w) {
      t.non_zero += sh[w].non_zero;
      t.max = fmax(t.max, sh[w].max);
      t.sum += sh[w].sum;
      t.sum_log += sh[w].sum_log;
    }
    partial[(uint64_t)col * n_slabs + slab] = t;
  }
}

__global__ void col_stats_final_kernel(const ColPartial *__restrict__ partial, uint32_t n_slabs, uint32_t n_cols, double power,
                                       double *__restrict__ col_stats) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  ColPartial t{0., 0., 0., 0.};
  for (uint32_t k = 0; k < n_slabs; ++k) {  // slab order = k-mer order
    const ColPartial p = partial[(uint64_t)c * n_slabs + k];
    t.non_zero += p.non_zero;
    t.max = fmax(t.max, p.max);
    t.sum += p.sum;
    t.sum_log += p.sum_log;
  }
  col_stats[4 * (uint64_t)c + 0] = t.non_zero;
  col_stats[4 * (uint64_t)c + 1] = t.max;
  col_stats[4 * (uint64_t)c + 2] = t.sum;
  col_stats[4 * (uint64_t)c + 3] = t.non_zero > 0. ? t.sum_log * power : 0.;
}

// one thread per k-mer, spectra visited in order: the reference's own summation order (lib/KMerDB.ml:182-215, Row)
__global__ __launch_bounds__(256) void row_stats_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint32_t n_cols,
                                                        uint64_t n_rows, double threshold0, double power, int power_one,
                                                        double *__restrict__ row_stats) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
    const int32_t *v = storage + r;
    double threshold = threshold0;
    if (threshold0 < 1.) {
      double s = 0.;
      for (uint32_t c = 0; c < n_cols; ++c) s += pow_count((double)v[(uint64_t)c * ld], power, power_one);
      threshold = threshold0 * s;
    }
    StatAcc a;
    uint32_t c = 0;
    for (; c + kStatUnroll <= n_cols; c += kStatUnroll) {
      int32_t x[kStatUnroll];
#pragma unroll
      for (int u = 0; u < kStatUnroll; ++u) x[u] = v[(uint64_t)(c + u) * ld];
#pragma unroll
      for (int u = 0; u < kStatUnroll; ++u) a.add(x[u], threshold, power, power_one);
      a.ls.renorm();
    }
    for (; c < n_cols; ++c) a.add(v[(uint64_t)c * ld], threshold, power, power_one);
    double *o = row_stats + 4 * r;
    o[0] = a.nz;
    o[1] = a.mx;
    o[2] = a.s;
    o[3] = a.nz > 0. ? a.ls.value() * power : 0.;
  }
}

// Int32.of_float on x86-64: truncate to the native int, keep the low 32 bits (lib/KMerDB.ml:716)
__device__ __forceinline__ int32_t int32_of_float(double x) {
  if (!(x > -9.2e18 && x < 9.2e18)) return 0;
  return (int32_t)(uint32_t)(uint64_t)(int64_t)x;
}

// a / b correctly rounded, given y = RN(1 / b) computed once per spectrum: two Newton corrections of the
// quotient on FMAs (Markstein: with y the correctly rounded reciprocal and q1 within an ulp, RN(q1 + r1 y) is
// RN(a / b)).  Five FMA-rate operations instead of the ~12-instruction v_div_scale / v_rcp / v_div_fmas / v_div_fixup
// sequence, and bit-identical to it (tests/test_gpu_counter.py::test_division_by_reciprocal_is_exact); y == 0 marks a
// divisor whose reciprocal is not safely rounded (all-ones mantissa) and takes the hardware division.
__device__ __forceinline__ double div_rn(double a, double b, double y) {
  if (y == 0.) return __ddiv_rn(a, b);
  const double q0 = __dmul_rn(a, y);
  const double r0 = __fma_rn(-b, q0, a);
  const double q1 = __fma_rn(r0, y, q0);
  const double r1 = __fma_rn(-b, q1, a);
  return __fma_rn(r1, y, q1);
}

// RN(1 / b), or 0 where div_rn must not be used: all-ones mantissas, and magnitudes far enough out that the
// residuals could leave the normal range
__device__ __forceinline__ double safe_reciprocal(double b) {
  const bool all_ones = (__double_as_longlong(b) & 0xFFFFFFFFFFFFFll) == 0xFFFFFFFFFFFFFll;
  return (all_ones || !(b > 0x1p-500 && b < 0x1p+500)) ? 0. : __ddiv_rn(1., b);
}

__global__ void reciprocal_kernel(const double *__restrict__ norm, uint32_t m, double *__restrict__ rcp) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) rcp[j] = safe_reciprocal(norm[j]);
}

// exposed for the test of div_rn: out[i] = a[i] / b[i] both ways
__global__ void division_probe_kernel(const double *__restrict__ a, const double *__restrict__ b, uint64_t n, double *__restrict__ fast,
                                      double *__restrict__ exact) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fast[i] = div_rn(a[i], b[i], safe_reciprocal(b[i]));
  exact[i] = __ddiv_rn(a[i], b[i]);
}

constexpr int kMeanUnroll = 8;

// RescaledMean: sum over the selected spectra, in the order given, of count * max_norm / norm
// (lib/KMerDB.ml:687-704).  sel/norm/rcp list only the spectra whose norm is positive (:693).  One thread per k-mer,
// eight spectra's loads in flight.
__global__ __launch_bounds__(256) void combine_mean_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint64_t n_rows,
                                                           const uint32_t *__restrict__ sel, const double *__restrict__ norm,
                                                           const double *__restrict__ rcp, uint32_t m, double max_norm,
                                                           int32_t *__restrict__ out, double *__restrict__ norm_partial) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  double acc_norm = 0.;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
    double s = 0.;
    uint32_t j = 0;
    for (; j + kMeanUnroll <= m; j += kMeanUnroll) {
      int32_t c[kMeanUnroll];
#pragma unroll
      for (int u = 0; u < kMeanUnroll; ++u) c[u] = __builtin_nontemporal_load(storage + (uint64_t)sel[j + u] * ld + r);
#pragma unroll
      for (int u = 0; u < kMeanUnroll; ++u) s = __dadd_rn(s, div_rn(__dmul_rn((double)c[u], max_norm), norm[j + u], rcp[j + u]));
    }
    for (; j < m; ++j)
      s = __dadd_rn(s, div_rn(__dmul_rn((double)storage[(uint64_t)sel[j] * ld + r], max_norm), norm[j], rcp[j]));
    acc_norm += s;
    out[r] = int32_of_float(s);
  }
  __shared__ double sh[4];
  acc_norm = wave_sum(acc_norm);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc_norm;
  __syncthreads();
  if (threadIdx.x == 0) norm_partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// RescaledMedian (lib/KMerDB.ml:705-706), up to 32 spectra: one thread per k-mer, the whole row in registers.  Loads
// are coalesced along k-mers exactly as in the mean; the P rescaled values are sorted by a fully unrolled bitonic
// network of v_min_f64 / v_max_f64 pairs (P/2 * log2(P) * (log2(P)+1) / 2 compare-exchanges, no LDS, no shuffles) and
// the upper median sorted[m/2] is picked with a select chain.  Unused slots hold +inf.
template <int P>
__global__ __launch_bounds__(256) void combine_median_thread_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint64_t n_rows,
                                                                    const uint32_t *__restrict__ sel, const double *__restrict__ norm,
                                                                    const double *__restrict__ rcp, uint32_t m, uint32_t n_sel,
                                                                    double max_norm, int32_t *__restrict__ out,
                                                                    double *__restrict__ norm_partial) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t me = m >> 1;
  double acc_norm = 0.;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
    int32_t c[P];
#pragma unroll
    for (int j = 0; j < P; ++j) c[j] = (uint32_t)j < m ? __builtin_nontemporal_load(storage + (uint64_t)sel[j] * ld + r) : 0;
    double v[P];
#pragma unroll
    for (int j = 0; j < P; ++j) v[j] = (uint32_t)j < m ? div_rn(__dmul_rn((double)c[j], max_norm), norm[j], rcp[j]) : INFINITY;
#pragma unroll
    for (int k = 2; k <= P; k <<= 1) {
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
          const int l = i ^ j;
          if (l > i) {
            // v_min_f64 + v_max_f64; a compare with four 32-bit selects measured 1.5x slower
            const double lo = fmin(v[i], v[l]), hi = fmax(v[i], v[l]);
            const bool up = (i & k) == 0;
            v[i] = up ? lo : hi;
            v[l] = up ? hi : lo;
          }
        }
      }
    }
    double med = m ? v[0] : 0.;
#pragma unroll
    for (int j = 1; j < P; ++j)
      if (me == (uint32_t)j) med = v[j];
    const double res = __dmul_rn(med, (double)n_sel);
    acc_norm += res;
    out[r] = int32_of_float(res);
  }
  __shared__ double sh[4];
  acc_norm = wave_sum(acc_norm);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc_norm;
  __syncthreads();
  if (threadIdx.x == 0) norm_partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// the bit pattern of a double as an unsigned key with the same order (negative values only arise from counts that
// wrapped around int32, but they must still sort)
__device__ __forceinline__ uint64_t ordered_key(double x) {
  const uint64_t b = (uint64_t)__double_as_longlong(x);
  return b ^ ((uint64_t)((int64_t)b >> 63) | 0x8000000000000000ull);
}
__device__ __forceinline__ double ordered_value(uint64_t k) {
  const uint64_t b = (k & 0x8000000000000000ull) ? (k ^ 0x8000000000000000ull) : ~k;
  return __longlong_as_double((long long)b);
}

// The element of rank `target` (0-based, ascending) among the wave's 64*R values, without sorting them: quickselect on
// wave ballots.  The values strictly between `lo` and `hi` are still candidates; the first candidate in (register, lane)
// order is the pivot (the registers are looked at until one holds a candidate); two comparisons per register count the
// values below the pivot and those not above it -- over all the values, so the bounds need not be applied -- and one of
// the bounds moves.  Every step is wave-uniform (no divergence, no LDS, no cross-lane data movement but one readlane);
// ties and the zeros that dominate sparse spectra finish in a step.  Expected ~2 ln(m) steps of ~2R comparisons against
// the ~R log^2(64R) compare-exchanges plus cross-lane shuffles of a full sort.
// The values are compared as the doubles they are (f64 comparisons issue at the full rate, 64-bit integer ones on
// order-preserving keys do not): there is no NaN among them -- the norms are positive and finite -- and +inf is the
// padding of the empty slots, never a candidate (hi starts there).
template <int R>
__device__ __forceinline__ double wave_select_rank(const double (&v)[R], uint32_t target) {
  double lo = -INFINITY, hi = INFINITY;
  for (;;) {
    double pivot = 0.;
    bool found = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!found) {
        const uint64_t alive = __ballot(v[r] > lo && v[r] < hi);
        if (alive) {
          const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)alive) - 1);  // (uniform already: v_readlane, not a trip through LDS)
          const uint64_t bits = (uint64_t)__double_as_longlong(v[r]);
          pivot = __longlong_as_double((long long)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(bits >> 32), src) << 32) |
                                                   (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bits, src)));
          found = true;
        }
      }
    }
    if (!found) return lo;  // cannot happen for target < number of values
    uint32_t n_lt = 0, n_le = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      n_lt += (uint32_t)__popcll(__ballot(v[r] < pivot));
      n_le += (uint32_t)__popcll(__ballot(v[r] <= pivot));
    }
    if (target < n_lt) hi = pivot;
    else if (target < n_le) return pivot;
    else lo = pivot;
  }
}

// RescaledMedian, 65 .. 64*R spectra: one wavefront per k-mer.  A tile of TR k-mers x m spectra of raw counts is staged in
// LDS; each wave then takes a k-mer, rescales its m counts into registers (R per lane; which lane holds which spectrum
// does not matter to a sort), finds the value of rank m/2 among the 64*R (empty slots carry +inf) with
// wave_select_rank, and lane 0 stores it times n_sel.
//
// The staging is what the time went into (23.5 of 35.6 ms at 8.39 M k-mers x 500 spectra, measured with the selection
// taken out): a dependent pair of loads per count (sel[j], then the count), four in flight per thread, and nothing in
// flight at all while the waves select.  So: the spectra's offsets are put in LDS once; a thread fetches 16 bytes (four
// k-mers of one spectrum) at a time, all its fetches of a tile at once; and the fetches of tile t + 1 are issued before
// the selection of tile t and land in registers under it.
template <int R>
struct MedianTile {
  // Threads per block and k-mers per tile.  A tile row of 32 k-mers is a whole 128-byte line of its spectrum: narrower rows
  // fetch every line once per tile that touches it (FETCH_SIZE 2.0x the counts at 16 k-mers, 4.0x at 8).  From 257 spectra
  // on, a tile of 32 k-mers no longer fits beside three other blocks' -- and the selection wants the waves -- so the block
  // grows to 8 waves instead of the tile shrinking; beyond 1,024 spectra the tile of an 8-wave block holds 16 k-mers (half
  // lines: 2.0x) where a 4-wave block's held 8.
  static constexpr uint32_t NT = R >= 8 ? 512 : 256;
  static constexpr uint32_t TR = R <= 4 ? 128 / R : R <= 16 ? 32 : 16;
  static constexpr uint32_t NV = (uint32_t)R * 64 * (TR / 4) / NT;  // 16-byte fetches per thread and tile, at most
};

template <int R, bool VEC>
__global__ __launch_bounds__(MedianTile<R>::NT) void combine_median_wave_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint64_t n_rows,
                                                                  const uint32_t *__restrict__ sel, const double *__restrict__ norm,
                                                                  const double *__restrict__ rcp, uint32_t m, uint32_t n_sel,
                                                                  double max_norm, int no_select, int32_t *__restrict__ out,
                                                                  double *__restrict__ norm_partial) {
  constexpr uint32_t NT = MedianTile<R>::NT, TR = MedianTile<R>::TR, TRp = TR + 1, NV = MedianTile<R>::NV, Q = TR / 4;
  extern __shared__ uint64_t median_lds[];  // [m] offsets of the spectra, then the tile [m][TR + 1] of counts
  uint64_t *s_off = median_lds;
  int32_t *tile32 = reinterpret_cast<int32_t *>(median_lds + m);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (uint32_t j = threadIdx.x; j < m; j += NT) s_off[j] = (uint64_t)sel[j] * ld;
  double b[R], y[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t col = (uint32_t)r * 64 + lane;
    b[r] = col < m ? norm[col] : 1.;
    y[r] = col < m ? rcp[col] : 1.;
  }
  __syncthreads();
  const uint32_t me = m >> 1, n_vec = m * Q;
  const uint64_t n_tiles = (n_rows + TR - 1) / TR;
  int4 pre[NV];
  auto fetch = [&](uint64_t t) {
    const uint64_t row0 = t * TR;
#pragma unroll
    for (uint32_t q = 0; q < NV; ++q) {
      const uint32_t e = threadIdx.x + NT * q;
      const uint32_t j = e / Q, r4 = (e % Q) * 4;
      int4 v = make_int4(0, 0, 0, 0);
      if (e < n_vec && row0 + r4 + 4 <= ld) {
        const int32_t *p = storage + s_off[j] + row0 + r4;
        if (VEC) {
          typedef int v4i __attribute__((ext_vector_type(4)));
          const v4i w = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p));
          v = make_int4(w.x, w.y, w.z, w.w);
        } else
          v = make_int4(p[0], p[1], p[2], p[3]);
      }
      pre[q] = v;
    }
  };
  double acc_norm = 0.;
  if (blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint64_t row0 = t * TR;
    __syncthreads();  // the tile before this one has been read
#pragma unroll
    for (uint32_t q = 0; q < NV; ++q) {
      const uint32_t e = threadIdx.x + NT * q;
      if (e < n_vec) {
        int32_t *d = tile32 + (e / Q) * TRp + (e % Q) * 4;
        d[0] = pre[q].x;
        d[1] = pre[q].y;
        d[2] = pre[q].z;
        d[3] = pre[q].w;
      }
    }
    __syncthreads();
    if (t + gridDim.x < n_tiles) fetch(t + gridDim.x);  // in flight under the selection
    for (uint32_t rr = wv; rr < TR && row0 + rr < n_rows; rr += NT / 64) {
      double v[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t col = (uint32_t)r * 64 + lane;
        v[r] = INFINITY;
        if (col < m) v[r] = div_rn(__dmul_rn((double)tile32[col * TRp + rr], max_norm), b[r], y[r]);
      }
      const double med = no_select ? v[0] : m ? wave_select_rank<R>(v, me) : 0.;
      const double res = __dmul_rn(med, (double)n_sel);
      if (lane == 0) {
        acc_norm += res;
        out[row0 + rr] = int32_of_float(res);
      }
    }
  }
  __shared__ double sh[NT / 64];
  acc_norm = wave_sum(acc_norm);
  if (lane == 0) sh[wv] = acc_norm;
  __syncthreads();
  if (threadIdx.x == 0) {
    double total = 0.;
#pragma unroll
    for (uint32_t w = 0; w < NT / 64; ++w) total += sh[w];
    norm_partial[blockIdx.x] = total;
  }
}

// RescaledMedian beyond 1024 spectra (up to 4096): a tile of R k-mers x m spectra is staged in LDS (lanes along
// k-mers, so the loads are full lines), every k-mer's row is sorted by a bitonic network run by the whole block, and
// the upper median sorted[m/2] is multiplied by the number of selected spectra.  Row stride P+1 keeps the staging
// writes off one bank.
__global__ __launch_bounds__(256) void combine_median_block_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint64_t n_rows,
                                                             const uint32_t *__restrict__ sel, const double *__restrict__ norm,
                                                             uint32_t m, uint32_t n_sel, double max_norm, uint32_t P, uint32_t R,
                                                             int32_t *__restrict__ out, double *__restrict__ norm_partial) {
  extern __shared__ double tile[];
  const uint32_t S = P + 1;
  const uint64_t n_tiles = (n_rows + R - 1) / R;
  double acc_norm = 0.;
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint64_t row0 = t * R;
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < R * P; e += blockDim.x) {
      const uint32_t r = e % R, j = e / R;
      double v = INFINITY;
      if (j < m && row0 + r < n_rows)
        v = __ddiv_rn(__dmul_rn((double)storage[(uint64_t)sel[j] * ld + row0 + r], max_norm), norm[j]);
      tile[r * S + j] = v;
    }
    __syncthreads();
    const uint32_t half = P >> 1;
    for (uint32_t k = 2; k <= P; k <<= 1) {
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t q = threadIdx.x; q < R * half; q += blockDim.x) {
          const uint32_t r = q / half, i = q % half;
          const uint32_t a = 2 * j * (i / j) + (i % j), b = a + j;
          double *row = tile + r * S;
          const double x = row[a], y = row[b];
          const bool up = (a & k) == 0;
          if ((x > y) == up) {
            row[a] = y;
            row[b] = x;
          }
        }
        __syncthreads();
      }
    }
    if (threadIdx.x < R && row0 + threadIdx.x < n_rows) {
      const double med = m ? tile[threadIdx.x * S + m / 2] : 0.;
      const double res = __dmul_rn(med, (double)n_sel);
      acc_norm += res;
      out[row0 + threadIdx.x] = int32_of_float(res);
    }
  }
  __shared__ double sh[4];
  acc_norm = wave_sum(acc_norm);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc_norm;
  __syncthreads();
  if (threadIdx.x == 0) norm_partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// RescaledMedian of any number of spectra (the fallback beyond 4096): one thread per k-mer resolves the key of rank
// m/2 bit by bit, most significant first -- 64 passes over the k-mer's m counts, each a coalesced stream (lanes along
// k-mers), rescaling on the fly; nothing is stored per k-mer but the prefix found so far and the remaining rank.
__global__ __launch_bounds__(256) void combine_median_bits_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint64_t n_rows,
                                                                  const uint32_t *__restrict__ sel, const double *__restrict__ norm,
                                                                  const double *__restrict__ rcp, uint32_t m, uint32_t n_sel,
                                                                  double max_norm, int32_t *__restrict__ out,
                                                                  double *__restrict__ norm_partial) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  double acc_norm = 0.;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
    uint64_t prefix = 0ull, known = 0ull;  // bits decided so far and their mask
    uint32_t rank = m >> 1;
    for (int bit = 63; bit >= 0 && m; --bit) {
      const uint64_t b = 1ull << bit;
      uint32_t zeros = 0;
      for (uint32_t j = 0; j < m; ++j) {
        const uint64_t key = ordered_key(div_rn(__dmul_rn((double)storage[(uint64_t)sel[j] * ld + r], max_norm), norm[j], rcp[j]));
        zeros += ((key & known) == prefix && !(key & b)) ? 1u : 0u;
      }
      if (rank >= zeros) {
        rank -= zeros;
        prefix |= b;
      }
      known |= b;
    }
    const double res = __dmul_rn(m ? ordered_value(prefix) : 0., (double)n_sel);
    acc_norm += res;
    out[r] = int32_of_float(res);
  }
  __shared__ double sh[4];
  acc_norm = wave_sum(acc_norm);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc_norm;
  __syncthreads();
  if (threadIdx.x == 0) norm_partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// one block: strided per-thread sums, then a fixed tree (the same result on every run for a given n)
__global__ __launch_bounds__(1024) void sum_partials_kernel(const double *__restrict__ partial, uint32_t n, double *__restrict__ out) {
  double s = 0.;
  for (uint32_t i = threadIdx.x; i < n; i += 1024) s += partial[i];
  __shared__ double sh[16];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.;
    for (int w = 0; w < 16; ++w) t += sh[w];
    *out = t;
  }
}

// Stdlib.max on floats: `if a >= b then a else b` -- a NaN second argument comes back out
__device__ __forceinline__ double ocaml_max(double a, double b) { return a >= b ? a : b; }

// Transformation.compute (lib/KMerDB.ml:96-144); cs = {non_zero, max, sum, sum_log} of the element's spectrum
__device__ __forceinline__ double transform_one(int which, double threshold, double power, const double *__restrict__ cs,
                                                int32_t icounts) {
  const double counts = (double)icounts;
  const double non_zero = cs[0], cmax = cs[1], csum = cs[2], csum_log = cs[3];
  if (threshold < 1.) threshold *= csum;
  switch (which) {
    case KPOP_TRANSF_BINARY: return counts >= threshold ? 1. : 0.;
    case KPOP_TRANSF_POWER:
      if (power == 1.) return counts >= threshold ? counts : 0.;
      return counts >= threshold ? pow_nn(counts, power) : 0.;
    case KPOP_TRANSF_CLR: {
      double v = counts >= threshold ? counts : 0.;
      v = ocaml_max(v, 0.1);
      return log(v) * power - csum_log / non_zero;
    }
    default: {
      double v;
      if (power == 0.) v = cmax * log((counts + 1.) / threshold);
      else {
        const double red = ocaml_max(0., threshold - 1.), c_p = pow_nn(red, power);
        if (power < 1.) v = (pow_nn(counts, power) - c_p) * pow_nn(cmax, 1. - power) / power;
        else v = (pow_nn(counts, power) - c_p) / (pow_nn(threshold, power) - c_p);
      }
      return ocaml_max(0., floor(v) / csum);
    }
  }
}

// spectra-major output: out[c][r], the storage orientation (-t with --table-transpose true, and -s)
__global__ __launch_bounds__(256) void transform_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint32_t n_cols,
                                                        uint64_t n_rows, int which, double threshold, double power,
                                                        const double *__restrict__ col_stats, double *__restrict__ out) {
  const uint32_t c = blockIdx.y;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const int32_t *v = storage + (uint64_t)c * ld;
  double *o = out + (uint64_t)c * n_rows;
  const double cs[4] = {col_stats[4 * (uint64_t)c], col_stats[4 * (uint64_t)c + 1], col_stats[4 * (uint64_t)c + 2],
                        col_stats[4 * (uint64_t)c + 3]};
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; r + 7 * stride < n_rows; r += 8 * stride) {
    int32_t x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = __builtin_nontemporal_load(v + r + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(transform_one(which, threshold, power, cs, x[u]), o + r + u * stride);
  }
  for (; r < n_rows; r += stride) o[r] = transform_one(which, threshold, power, cs, v[r]);
}

// k-mer-major output: out[r][c] (the default table of -t).  64 x 64 tiles through LDS so that both the reads
// (along k-mers) and the writes (along spectra) are contiguous.
__global__ __launch_bounds__(256) void transform_table_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint32_t n_cols,
                                                              uint64_t n_rows, int which, double threshold, double power,
                                                              const double *__restrict__ col_stats, double *__restrict__ out) {
  __shared__ double tile[64][65];
  const uint64_t row_tiles = (n_rows + 63) / 64;
  const uint32_t col0 = blockIdx.y * 64;
  for (uint64_t rt = blockIdx.x; rt < row_tiles; rt += gridDim.x) {
    const uint64_t row0 = rt * 64;
    __syncthreads();
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
      const uint32_t r = threadIdx.x & 63, c = (threadIdx.x >> 6) + 4 * q;
      if (row0 + r < n_rows && col0 + c < n_cols)
        tile[c][r] = transform_one(which, threshold, power, col_stats + 4 * (uint64_t)(col0 + c),
                                   storage[(uint64_t)(col0 + c) * ld + row0 + r]);
    }
    __syncthreads();
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
      const uint32_t c = threadIdx.x & 63, r = (threadIdx.x >> 6) + 4 * q;
      if (row0 + r < n_rows && col0 + c < n_cols) __builtin_nontemporal_store(tile[c][r], out + (row0 + r) * n_cols + col0 + c);
    }
  }
}

uint32_t n_slabs_for(uint64_t n_rows) { return (uint32_t)std::max<uint64_t>(1, (n_rows + kRowsPerStatBlock - 1) / kRowsPerStatBlock); }

int check_transform(int which, double threshold, double power, const char *who) {
  if (which < KPOP_TRANSF_BINARY || which > KPOP_TRANSF_PSEUDO) KPOP_FAIL(KPOP_ERR_INVALID, "%s: unknown transformation %d", who, which);
  if (!(threshold >= 0.) || !(power >= 0.))  // the CLI only accepts non-negative values (bin/KPopCountDB.ml:216-233)
    KPOP_FAIL(KPOP_ERR_INVALID, "%s: Invalid_transformation(%g, %g)", who, threshold, power);
  return 0;
}

// host columns -> device [n_cols][ld]
int upload_columns(const int32_t *const *columns, uint32_t n_cols, uint64_t n_rows, DevBuf &d, uint64_t *ld_out, hipStream_t st) {
  const uint64_t ld = kpop_dev_counter_ld(n_rows);
  KPOP_TRY(d.alloc((uint64_t)n_cols * ld * 4));
  for (uint32_t c = 0; c < n_cols; ++c) {
    if (!columns[c] && n_rows) KPOP_FAIL(KPOP_ERR_INVALID, "k-mer database: column %u is null", c);
    if (n_rows) KPOP_HIP(hipMemcpyAsync(d.as<int32_t>() + (uint64_t)c * ld, columns[c], n_rows * 4, hipMemcpyHostToDevice, st));
  }
  *ld_out = ld;
  return 0;
}

}  // namespace

}  // namespace kpop

using namespace kpop;

extern "C" uint64_t kpop_dev_counter_ld(uint64_t n_rows) { return (n_rows + 31) / 32 * 32; }

extern "C" uint64_t kpop_dev_counter_workspace_bytes(uint32_t n_cols, uint64_t n_rows) {
  const uint64_t slabs = n_slabs_for(n_rows);
  return (uint64_t)n_cols * slabs * (sizeof(ColPartial) + 8) + (uint64_t)n_cols * 8 + (1u << 16) * 8 + 4096;
}

extern "C" int kpop_dev_counter_stats(const int32_t *d_storage, uint64_t ld, uint32_t n_cols, uint64_t n_rows, double threshold,
                                      double power, void *d_workspace, double *d_col_stats, double *d_row_stats, void *stream) {
  KPOP_TRY(require_init());
  KPOP_TRY(check_transform(KPOP_TRANSF_POWER, threshold, power, "kpop_dev_counter_stats"));
  if (n_cols == 0) return KPOP_OK;
  hipStream_t st = as_stream(stream);
  const uint32_t slabs = n_slabs_for(n_rows);
  const int p1 = power == 1.;
  if (d_col_stats) {
    if (!d_workspace) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_counter_stats: workspace is null");
    ColPartial *partial = reinterpret_cast<ColPartial *>(d_workspace);
    double *plain = reinterpret_cast<double *>(partial + (uint64_t)n_cols * slabs);
    double *thr = plain + (uint64_t)n_cols * slabs;
    for (uint32_t c0 = 0; c0 < n_cols; c0 += 65535) {  // spectra ride on grid.y
      const uint32_t nc = std::min<uint32_t>(65535, n_cols - c0);
      const int32_t *src = d_storage + (uint64_t)c0 * ld;
      ColPartial *pp = partial + (uint64_t)c0 * slabs;
      double *pl = plain + (uint64_t)c0 * slabs;
      if (threshold < 1.) {
        col_plain_sum_kernel<<<dim3(slabs, nc), dim3(kStatBlock), 0, st>>>(src, ld, n_rows, power, p1, slabs, pl);
        KPOP_LAUNCH_CHECK();
      }
      col_threshold_kernel<<<dim3(div_up(nc, 256)), dim3(256), 0, st>>>(pl, slabs, nc, threshold, thr + c0);
      KPOP_LAUNCH_CHECK();
      col_stats_kernel<<<dim3(slabs, nc), dim3(kStatBlock), 0, st>>>(src, ld, n_rows, power, p1, thr + c0, slabs, pp);
      KPOP_LAUNCH_CHECK();
      col_stats_final_kernel<<<dim3(div_up(nc, 256)), dim3(256), 0, st>>>(pp, slabs, nc, power, d_col_stats + 4 * (uint64_t)c0);
      KPOP_LAUNCH_CHECK();
    }
  }
  if (d_row_stats && n_rows) {
    row_stats_kernel<<<dim3(capped_grid(div_up(n_rows, 256))), dim3(256), 0, st>>>(d_storage, ld, n_cols, n_rows, threshold, power, p1,
                                                                                  d_row_stats);
    KPOP_LAUNCH_CHECK();
  }
  return KPOP_OK;
}

extern "C" int kpop_dev_counter_combine(const int32_t *d_storage, uint64_t ld, uint64_t n_rows, const uint32_t *d_sel,
                                        const double *d_norm, uint32_t n_valid, uint32_t n_sel, double max_norm, int criterion,
                                        void *d_workspace, int32_t *d_out, double *d_out_norm, void *stream) {
  KPOP_TRY(require_init());
  if (criterion != KPOP_COMBINE_MEAN && criterion != KPOP_COMBINE_MEDIAN)
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_counter_combine: Unknown_combination_criterion(%d)", criterion);
  if (n_rows == 0) return KPOP_OK;
  if (!d_workspace || !d_out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_counter_combine: null argument");
  // the median kernels stage groups of four rows: a leading dimension that is not kpop_dev_counter_ld's (a multiple of 32)
  // would have them treat the last rows of a short group as padding
  if (ld < n_rows || (ld & 3) != 0)
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_counter_combine: ld=%llu for %llu rows (use kpop_dev_counter_ld: at least n_rows, a multiple of 4)",
              (unsigned long long)ld, (unsigned long long)n_rows);
  hipStream_t st = as_stream(stream);
  double *partial = reinterpret_cast<double *>(d_workspace);
  double *rcp = partial + (1u << 16);
  if (n_valid) {
    reciprocal_kernel<<<dim3(div_up(n_valid, 256)), dim3(256), 0, st>>>(d_norm, n_valid, rcp);
    KPOP_LAUNCH_CHECK();
  }
  uint32_t grid;
  if (criterion == KPOP_COMBINE_MEAN) {
    grid = std::min<uint32_t>(div_up(n_rows, 256), 1u << 16);
    combine_mean_kernel<<<dim3(grid), dim3(256), 0, st>>>(d_storage, ld, n_rows, d_sel, d_norm, rcp, n_valid, max_norm, d_out, partial);
  } else if (n_valid <= 32) {
    // (33 .. 64 spectra go to the wave kernel: 2.69 against 2.49 ms at 8.39 M k-mers x 64 when every count is positive, but
    // 1.08 ms when 70 % are zero, and the tables this runs on are mostly zeros)
    grid = std::min<uint32_t>(div_up(n_rows, 256), 1u << 16);
#define KPOP_MEDIAN_THREAD(PP)                                                                                             \
  combine_median_thread_kernel<PP><<<dim3(grid), dim3(256), 0, st>>>(d_storage, ld, n_rows, d_sel, d_norm, rcp, n_valid, n_sel, \
                                                                      max_norm, d_out, partial)
    if (n_valid <= 8) KPOP_MEDIAN_THREAD(8);
    else if (n_valid <= 16) KPOP_MEDIAN_THREAD(16);
    else KPOP_MEDIAN_THREAD(32);
#undef KPOP_MEDIAN_THREAD
  } else if (n_valid <= 2048) {
    const bool vec = (reinterpret_cast<uintptr_t>(d_storage) & 15) == 0 && (ld & 3) == 0;
    const int no_select = (ctx().tune_dbg & 256) ? 1 : 0;  // (a probe: the staging alone)
#define KPOP_MEDIAN_WAVE(RR, VV)                                                                                                    \
  do {                                                                                                                              \
    constexpr uint32_t TR = MedianTile<RR>::TR;                                                                                     \
    const size_t lds = (size_t)n_valid * 8 + (size_t)n_valid * (TR + 1) * 4;                                                        \
    static PerSlotOnce attr_once;                                                                                                   \
    bool &attr_set = attr_once();                                                                                \
    if (!attr_set) {                                                                                                                \
      KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&combine_median_wave_kernel<RR, VV>),                             \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));                                 \
      attr_set = true;                                                                                                              \
    }                                                                                                                               \
    grid = (uint32_t)std::min<uint64_t>((n_rows + TR - 1) / TR, 1u << 16);                                                          \
    combine_median_wave_kernel<RR, VV><<<dim3(grid), dim3(MedianTile<RR>::NT), lds, st>>>(d_storage, ld, n_rows, d_sel, d_norm, rcp, n_valid, \
                                                                            n_sel, max_norm, no_select, d_out, partial);           \
  } while (0)
#define KPOP_MEDIAN_WAVE_R(RR)       \
  do {                               \
    if (vec) KPOP_MEDIAN_WAVE(RR, true); \
    else KPOP_MEDIAN_WAVE(RR, false);    \
  } while (0)
    if (n_valid <= 64) KPOP_MEDIAN_WAVE_R(1);
    else if (n_valid <= 128) KPOP_MEDIAN_WAVE_R(2);
    else if (n_valid <= 256) KPOP_MEDIAN_WAVE_R(4);
    else if (n_valid <= 512) KPOP_MEDIAN_WAVE_R(8);
    else if (n_valid <= 1024) KPOP_MEDIAN_WAVE_R(16);
    else KPOP_MEDIAN_WAVE_R(32);
#undef KPOP_MEDIAN_WAVE_R
#undef KPOP_MEDIAN_WAVE
  } else {
    uint32_t P = 2;
    while (P < n_valid) P <<= 1;
    const uint32_t budget = 8192 - 8;  // doubles of LDS (64 KB less the reduction scratch)
    if (P + 1 > budget) {  // beyond 4096 spectra: bit-serial selection, any m
      grid = std::min<uint32_t>(div_up(n_rows, 256), 1u << 16);
      combine_median_bits_kernel<<<dim3(grid), dim3(256), 0, st>>>(d_storage, ld, n_rows, d_sel, d_norm, rcp, n_valid, n_sel, max_norm,
                                                                  d_out, partial);
      KPOP_LAUNCH_CHECK();
      if (d_out_norm) {
        sum_partials_kernel<<<dim3(1), dim3(1024), 0, st>>>(partial, grid, d_out_norm);
        KPOP_LAUNCH_CHECK();
      }
      return KPOP_OK;
    }
    const uint32_t R = std::max<uint32_t>(1, std::min<uint32_t>(64, budget / (P + 1)));
    grid = (uint32_t)std::min<uint64_t>((n_rows + R - 1) / R, 1u << 16);
    combine_median_block_kernel<<<dim3(grid), dim3(256), (size_t)R * (P + 1) * 8, st>>>(d_storage, ld, n_rows, d_sel, d_norm, n_valid, n_sel,
                                                                                   max_norm, P, R, d_out, partial);
  }
  KPOP_LAUNCH_CHECK();
  if (d_out_norm) {
    sum_partials_kernel<<<dim3(1), dim3(1024), 0, st>>>(partial, grid, d_out_norm);
    KPOP_LAUNCH_CHECK();
  }
  return KPOP_OK;
}

// test hook: a[i] / b[i] by the reciprocal route of the combination kernels and by the hardware division
extern "C" int kpop_dev_division_probe(const double *d_a, const double *d_b, uint64_t n, double *d_fast, double *d_exact, void *stream) {
  KPOP_TRY(require_init());
  if (n == 0) return KPOP_OK;
  division_probe_kernel<<<dim3(capped_grid(div_up(n, 256))), dim3(256), 0, as_stream(stream)>>>(d_a, d_b, n, d_fast, d_exact);
  KPOP_LAUNCH_CHECK();
  return KPOP_OK;
}

extern "C" int kpop_dev_counter_transform(const int32_t *d_storage, uint64_t ld, uint32_t n_cols, uint64_t n_rows, int which,
                                          double threshold, double power, const double *d_col_stats, int kmer_major, double *d_out,
                                          void *stream) {
  KPOP_TRY(require_init());
  KPOP_TRY(check_transform(which, threshold, power, "kpop_dev_counter_transform"));
  if (n_cols == 0 || n_rows == 0) return KPOP_OK;
  hipStream_t st = as_stream(stream);
  if (n_cols > 65535 * 64ull) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_dev_counter_transform: too many spectra");
  if (kmer_major) {
    transform_table_kernel<<<dim3((uint32_t)std::min<uint64_t>((n_rows + 63) / 64, 1u << 20), div_up(n_cols, 64)), dim3(256), 0, st>>>(
        d_storage, ld, n_cols, n_rows, which, threshold, power, d_col_stats, d_out);
  } else {
    for (uint32_t c0 = 0; c0 < n_cols; c0 += 65535) {  // spectra ride on grid.y
      const uint32_t nc = std::min<uint32_t>(65535, n_cols - c0);
      transform_kernel<<<dim3((uint32_t)std::min<uint64_t>(div_up(n_rows, 256 * 8), 1u << 16), nc), dim3(256), 0, st>>>(
          d_storage + (uint64_t)c0 * ld, ld, nc, n_rows, which, threshold, power, d_col_stats + 4 * (uint64_t)c0,
          d_out + (uint64_t)c0 * n_rows);
      KPOP_LAUNCH_CHECK();
    }
  }
  KPOP_LAUNCH_CHECK();
  return KPOP_OK;
}

// ---------------------------------------------------------------------------
// host-buffer entry points
// ---------------------------------------------------------------------------
extern "C" int kpop_counter_stats(const int32_t *const *columns, uint32_t n_cols, uint64_t n_rows, double threshold, double power,
                                  double *col_stats, double *row_stats) {
  KPOP_TRY(require_init());
  KPOP_TRY(check_transform(KPOP_TRANSF_POWER, threshold, power, "kpop_counter_stats"));
  if (n_cols == 0) {  // a k-mer that occurs in no spectrum has all-zero statistics
    if (row_stats) std::fill(row_stats, row_stats + 4 * n_rows, 0.);
    return KPOP_OK;
  }
  if (!columns) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_counter_stats: columns is null");
  hipStream_t st = nullptr;
  DevBuf ds, dw, dc, dr;
  uint64_t ld;
  KPOP_TRY(upload_columns(columns, n_cols, n_rows, ds, &ld, st));
  KPOP_TRY(dw.alloc(kpop_dev_counter_workspace_bytes(n_cols, n_rows)));
  if (col_stats) KPOP_TRY(dc.alloc((uint64_t)n_cols * 32));
  if (row_stats) KPOP_TRY(dr.alloc(n_rows * 32));
  KPOP_TRY(kpop_dev_counter_stats(ds.as<int32_t>(), ld, n_cols, n_rows, threshold, power, dw.p, col_stats ? dc.as<double>() : nullptr,
                                  row_stats ? dr.as<double>() : nullptr, st));
  if (col_stats) KPOP_HIP(hipMemcpyAsync(col_stats, dc.p, (uint64_t)n_cols * 32, hipMemcpyDeviceToHost, st));
  if (row_stats && n_rows) KPOP_HIP(hipMemcpyAsync(row_stats, dr.p, n_rows * 32, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  return KPOP_OK;
}

extern "C" int kpop_counter_combine(const int32_t *const *columns, uint64_t n_rows, const uint32_t *sel, uint32_t n_sel,
                                    const double *col_sum, int criterion, int32_t *out, double *out_norm) {
  KPOP_TRY(require_init());
  if (criterion != KPOP_COMBINE_MEAN && criterion != KPOP_COMBINE_MEDIAN)
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_counter_combine: Unknown_combination_criterion(%d)", criterion);
  if (out_norm) *out_norm = 0.;
  if (n_rows == 0) return KPOP_OK;
  if (!out || (n_sel && (!columns || !sel || !col_sum))) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_counter_combine: null argument");
  // only the spectra with a positive norm enter the histogram (lib/KMerDB.ml:693); they are packed in visiting order
  std::vector<const int32_t *> cols;
  std::vector<uint32_t> idx;
  std::vector<double> norm;
  double max_norm = 0.;
  for (uint32_t s = 0; s < n_sel; ++s) max_norm = std::max(max_norm, col_sum[sel[s]]);  // :646-660
  for (uint32_t s = 0; s < n_sel; ++s)
    if (col_sum[sel[s]] > 0.) {
      idx.push_back((uint32_t)cols.size());
      cols.push_back(columns[sel[s]]);
      norm.push_back(col_sum[sel[s]]);
    }
  const uint32_t m = (uint32_t)cols.size();
  hipStream_t st = nullptr;
  DevBuf ds, dsel, dnorm, dw, dout, dn;
  uint64_t ld;
  KPOP_TRY(upload_columns(cols.data(), m, n_rows, ds, &ld, st));
  KPOP_TRY(dsel.alloc((uint64_t)m * 4));
  KPOP_TRY(dnorm.alloc((uint64_t)m * 8));
  KPOP_TRY(dw.alloc(kpop_dev_counter_workspace_bytes(m, n_rows)));
  KPOP_TRY(dout.alloc(n_rows * 4));
  KPOP_TRY(dn.alloc(8));
  if (m) {
    KPOP_HIP(hipMemcpyAsync(dsel.p, idx.data(), (uint64_t)m * 4, hipMemcpyHostToDevice, st));
    KPOP_HIP(hipMemcpyAsync(dnorm.p, norm.data(), (uint64_t)m * 8, hipMemcpyHostToDevice, st));
  }
  KPOP_TRY(kpop_dev_counter_combine(ds.as<int32_t>(), ld, n_rows, dsel.as<uint32_t>(), dnorm.as<double>(), m, n_sel, max_norm,
                                    criterion, dw.p, dout.as<int32_t>(), dn.as<double>(), st));
  KPOP_HIP(hipMemcpyAsync(out, dout.p, n_rows * 4, hipMemcpyDeviceToHost, st));
  if (out_norm) KPOP_HIP(hipMemcpyAsync(out_norm, dn.p, 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  return KPOP_OK;
}

extern "C" int kpop_counter_transform(const int32_t *const *columns, uint32_t n_cols, uint64_t n_rows, int which, double threshold,
                                      double power, const double *col_stats, int kmer_major, double *out) {
  KPOP_TRY(require_init());
  KPOP_TRY(check_transform(which, threshold, power, "kpop_counter_transform"));
  if (n_cols == 0 || n_rows == 0) return KPOP_OK;
  if (!columns || !col_stats || !out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_counter_transform: null argument");
  hipStream_t st = nullptr;
  // spectra are independent: batches bound the device footprint to ~12 bytes per count of a batch
  const uint64_t budget = 1ull << 32;
  const uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_cols, budget / std::max<uint64_t>(1, n_rows * 12)));
  if (kmer_major && batch < n_cols) {
    // k-mer-major output interleaves the spectra: one batch only
    if ((uint64_t)n_cols * n_rows * 12 > (200ull << 30)) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_counter_transform: table larger than device memory");
  }
  const uint32_t step = kmer_major ? n_cols : batch;
  for (uint32_t c0 = 0; c0 < n_cols; c0 += step) {
    const uint32_t nc = std::min(step, n_cols - c0);
    DevBuf ds, dc, dout;
    uint64_t ld;
    KPOP_TRY(upload_columns(columns + c0, nc, n_rows, ds, &ld, st));
    KPOP_TRY(dc.alloc((uint64_t)nc * 32));
    KPOP_TRY(dout.alloc((uint64_t)nc * n_rows * 8));
    KPOP_HIP(hipMemcpyAsync(dc.p, col_stats + 4 * (uint64_t)c0, (uint64_t)nc * 32, hipMemcpyHostToDevice, st));
    KPOP_TRY(kpop_dev_counter_transform(ds.as<int32_t>(), ld, nc, n_rows, which, threshold, power, dc.as<double>(), kmer_major,
                                        dout.as<double>(), st));
    KPOP_HIP(hipMemcpyAsync(out + (kmer_major ? 0 : (uint64_t)c0 * n_rows), dout.p, (uint64_t)nc * n_rows * 8, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipStreamSynchronize(st));
  }
  return KPOP_OK;
}

// summary_large.hip -- per-row distance summaries when a row (r1 distances) does not fit LDS:
// the relatedness-engine case, a query against a database of 10^4..10^6 twisted vectors
// (README.md:1101; lib/Matrix.ml:691-766 with keep_at_most ~ 300).
//
// The r2 x r1 matrix is still never formed: query rows are processed in chunks whose distance
// rows live in a library workspace, and each row is reduced by one 1024-thread block with
//   - two ordered block reductions            mean, sample sd           (lib/Matrix.ml:651-655,657-683)
//   - SELECT by narrowing the key range       upper median, MAD, and the keep_at_most-th smallest distance: one
//                                             2048-bin pass + one pass that ranks the chosen bin's keys in LDS is the
//                                             common case (no sort of the row; more passes only for crowded values)
//   - ordered compaction + a small LDS sort   the closest rows, whole tie groups counted (:648-649)
// Differences from the r1 <= 4096 kernel: mean and sd are tree sums (not the reference's
// ascending chain), equal up to rounding (tests hold 1e-10); at most kLargeMaxNb neighbours
// are returned per row (out_n still reports the reference's eff_len).
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "summary_types.h"
#include "space_ops.h"

namespace kpop {

constexpr int kLT = 1024;             // threads per row
constexpr uint32_t kLargeMaxNb = 2048;  // neighbours returned per row at most (= kLargeNeighbours, distance.hip)

__device__ __forceinline__ uint64_t f64_key(double x) {  // order-preserving map to u64
  const uint64_t b = (uint64_t)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(uint64_t k) {
  const uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

// sum over the block, same value in every thread; fixed order => bitwise reproducible
__device__ double block_sum(double v, double *s_w) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = __dadd_rn(v, __shfl_down(v, o, 64));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_w[wv] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < kLT / 64; ++w) t = __dadd_rn(t, s_w[w]);
  return t;
}

// exclusive prefix of a 0/1 flag over the block (thread order) + block total
__device__ uint32_t block_scan_flag(bool flag, uint32_t *s_w, uint32_t *total) {
  const uint64_t m = __ballot(flag);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t below = (uint32_t)__popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
  __syncthreads();
  if (lane == 0) s_w[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t pre = 0, tot = 0;
  for (int w = 0; w < kLT / 64; ++w) {
    if (w < wv) pre += s_w[w];
    tot += s_w[w];
  }
  *total = tot;
  return pre + below;
}

// how a selection reaches its elements: a plain array, or a SAMPLE of a long row -- kSampleRun consecutive elements out of
// every `stride` (runs of whole 128-byte lines, so the sample costs what it reads)
constexpr uint32_t kSampleRun = 1024;
struct PlainRow {
  const double *p;
  __device__ __forceinline__ double operator[](uint32_t i) const { return p[i]; }
};
struct SampledRow {
  const double *p;
  uint32_t stride;  // elements between the starts of two runs
  __device__ __forceinline__ double operator[](uint32_t i) const { return p[(uint64_t)(i / kSampleRun) * stride + (i % kSampleRun)]; }
};

// TRANSFORM 0: key of row[i]; 1: key of |row[i] - centre|
template <int TRANSFORM, class Row>
__device__ __forceinline__ uint64_t elem_key(const Row &row, uint32_t i, double centre) {
  const double x = TRANSFORM ? fabs(__dsub_rn(row[i], centre)) : row[i];
  return f64_key(x);
}

// ---------------------------------------------------------------------------
// Selection by narrowing the key range.  A pass bins the keys of [lo, hi] into kBins equal (power-of-two wide) bins and
// finds the bin that holds the wanted rank; a row of smoothly spread distances leaves ~n / kBins keys there, so after ONE
// such pass the bin's keys fit an LDS list (kCand) and are ranked exactly.  A bin that is still too full (many equal or
// crowded values) becomes the new range: every pass narrows it at least 1024-fold, and a range of one key is a tie
// group, known without looking further.  Round 1 took eight 8-bit passes per selection whatever the data.
// Up to kSel ranks are served by the same passes (they share the row, not the bin).
// ---------------------------------------------------------------------------
constexpr uint32_t kBins = 2048, kCand = 2048;
constexpr int kSel = 2;     // (what most callers' LDS tables hold)
constexpr int kSelMax = 4;  // ... and the most a call may ask for, with tables of its own to match (the sample kernel: four ranks of one scan)
constexpr int kSelLoads = 8;  // loads in flight a thread in the passes of a selection
static_assert(kBins == 2 * kLT, "block_select_ranks: a thread owns two bins");

struct Sel {          // one selection in progress / done
  uint32_t rank;      // wanted 0-based rank
  uint64_t lo, hi;    // current key range (inclusive)
  uint32_t below;     // keys strictly below lo
  uint64_t value;     // result
  uint32_t n_less, n_equal;
  int done;
};

__device__ __forceinline__ int range_shift(uint64_t lo, uint64_t hi) {  // smallest s with (hi - lo) >> s < kBins
  const uint64_t span = hi - lo;
  const int bits = span ? 64 - __clzll((long long)span) : 0;
  return bits > 11 ? bits - 11 : 0;
}

// All threads call with the same arguments.  s_hist: kSel * kBins u32; s_cand: kSel * kCand u64; s_misc: 64 u32.
// SUMSQ: the first pass over the row also adds up (row[i] - centre)^2 into *sumsq_part (this thread's share; the caller
// reduces it) -- the sample variance rides on the first selection pass instead of a pass of its own.
template <int TRANSFORM, bool SUMSQ = false, class Row = PlainRow>
__device__ void block_select_ranks(const Row row, uint32_t n, double centre, Sel *sel, int n_sel, uint32_t *s_hist,
                                   uint64_t *s_cand, uint32_t *s_misc, double *sumsq_part = nullptr) {
  for (int round = 0; round < 8; ++round) {
    bool any = false;
    for (int t = 0; t < n_sel; ++t) any = any || !sel[t].done;
    if (!any) return;
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < (uint32_t)n_sel * kBins; q += kLT) s_hist[q] = 0;
    __syncthreads();
    int shift[kSelMax];
    for (int t = 0; t < n_sel; ++t) shift[t] = range_shift(sel[t].lo, sel[t].hi);
    double sq = 0.0;
    for (uint32_t i0 = threadIdx.x; i0 < n; i0 += kLT * kSelLoads) {  // eight loads in flight a thread (a block a CU, a chain of round trips: four were 45k of a 65k-cycle pass over 120,000 candidates)
      double v[kSelLoads];
#pragma unroll
      for (int u = 0; u < kSelLoads; ++u) v[u] = i0 + u * kLT < n ? row[i0 + u * kLT] : 0.0;
#pragma unroll
      for (int u = 0; u < kSelLoads; ++u) {
        if (i0 + u * kLT >= n) continue;
        const uint64_t k = f64_key(TRANSFORM ? fabs(__dsub_rn(v[u], centre)) : v[u]);
        if (SUMSQ && round == 0) {
          const double dv = __dsub_rn(v[u], centre);
          sq = __dadd_rn(sq, __dmul_rn(dv, dv));
        }
        for (int t = 0; t < n_sel; ++t)
          if (!sel[t].done && k >= sel[t].lo && k <= sel[t].hi) atomicAdd(&s_hist[t * kBins + (uint32_t)((k - sel[t].lo) >> shift[t])], 1u);
      }
    }
    if (SUMSQ && round == 0) *sumsq_part = sq;
    __syncthreads();
    // the bin of each rank: every thread owns two neighbouring bins, a scan over the block gives the keys below them (one
    // thread walking the 2,048 counters -- a chain of dependent LDS reads, ~0.1 ms -- was most of a selection's time)
    for (int t = 0; t < n_sel; ++t) {
      if (sel[t].done) continue;  // (uniform)
      const uint32_t c0 = s_hist[t * kBins + 2 * threadIdx.x], c1 = s_hist[t * kBins + 2 * threadIdx.x + 1];
      uint32_t incl = c0 + c1;
      const int lane_ = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane_ >= o) incl += up;
      }
      if (lane_ == 63) s_misc[16 + wv_] = incl;
      if (threadIdx.x == 0) s_misc[t * 4 + 0] = 0xFFFFFFFFu;
      __syncthreads();
      uint32_t before = sel[t].below, total = sel[t].below;
      for (int w = 0; w < kLT / 64; ++w) {
        if (w < wv_) before += s_misc[16 + w];
        total += s_misc[16 + w];
      }
      before += incl - (c0 + c1);  // keys below this thread's first bin
      const uint32_t rank = sel[t].rank;
      if (before <= rank && rank < before + c0) {
        s_misc[t * 4 + 0] = 2 * threadIdx.x;
        s_misc[t * 4 + 1] = before;
        s_misc[t * 4 + 2] = c0;
      } else if (before + c0 <= rank && rank < before + c0 + c1) {
        s_misc[t * 4 + 0] = 2 * threadIdx.x + 1;
        s_misc[t * 4 + 1] = before + c0;
        s_misc[t * 4 + 2] = c1;
      }
      __syncthreads();
      if (threadIdx.x == 0 && s_misc[t * 4 + 0] == 0xFFFFFFFFu) {  // the rank lies beyond the keys in range: only reachable with NaNs in the row
        s_misc[t * 4 + 0] = kBins - 1;
        s_misc[t * 4 + 1] = total;
        s_misc[t * 4 + 2] = s_hist[t * kBins + kBins - 1];
      }
    }
    __syncthreads();
    bool collect[kSelMax];
    for (int t = 0; t < n_sel; ++t) {
      collect[t] = false;
      if (sel[t].done) continue;
      const uint32_t bin = s_misc[t * 4 + 0], cum = s_misc[t * 4 + 1], cnt = s_misc[t * 4 + 2];
      const uint64_t blo = sel[t].lo + ((uint64_t)bin << shift[t]);
      // (u64 arithmetic spelt out: HIP's min() of an `unsigned long` and an `unsigned long long` goes through double, and
      // rounded these keys to multiples of 1,024 -- round 2's selections went wrong on tie groups of more than kCand)
      const uint64_t bend = blo + (((uint64_t)1 << shift[t]) - (uint64_t)1);
      const uint64_t bhi = shift[t] ? (bend < sel[t].hi ? bend : sel[t].hi) : blo;
#ifdef KPOP_SELECT_DEBUG
      if (threadIdx.x == 0) printf("round %d sel %d shift %d bin %u cum %u cnt %u range %.17g .. %.17g rank %u\n", round, t, shift[t], bin, cum, cnt, key_f64(blo), key_f64(bhi), sel[t].rank);
#endif
      sel[t].lo = blo;
      sel[t].hi = bhi;
      sel[t].below = cum;
      if (blo == bhi) {  // a single key: a tie group
        sel[t].value = blo;
        sel[t].n_less = cum;
        sel[t].n_equal = cnt;
        sel[t].done = 1;
      } else if (cnt <= kCand) {
        collect[t] = true;
      }
    }
    bool need = false;
    for (int t = 0; t < n_sel; ++t) need = need || collect[t];
    if (need) {
      // gather the keys of the chosen bins and rank them exactly
      __syncthreads();
      if (threadIdx.x < (uint32_t)n_sel) s_misc[32 + threadIdx.x] = 0;
      __syncthreads();
      for (uint32_t i0 = threadIdx.x; i0 < n; i0 += kLT * kSelLoads) {
        double v[kSelLoads];
#pragma unroll
        for (int u = 0; u < kSelLoads; ++u) v[u] = i0 + u * kLT < n ? row[i0 + u * kLT] : 0.0;
#pragma unroll
        for (int u = 0; u < kSelLoads; ++u) {
          if (i0 + u * kLT >= n) continue;
          const uint64_t k = f64_key(TRANSFORM ? fabs(__dsub_rn(v[u], centre)) : v[u]);
          for (int t = 0; t < n_sel; ++t)
            if (collect[t] && k >= sel[t].lo && k <= sel[t].hi) {
              const uint32_t at = atomicAdd(&s_misc[32 + t], 1u);
              if (at < kCand) s_cand[t * kCand + at] = k;
            }
        }
      }
      __syncthreads();
      for (int t = 0; t < n_sel; ++t) {
        if (!collect[t]) continue;
        const uint32_t m = min(s_misc[32 + t], kCand), want = sel[t].rank - sel[t].below;
        // the key with exactly `want` keys below it among the m candidates (ties: any member of the group whose span holds `want`)
        __syncthreads();
        if (threadIdx.x == 0) s_misc[40] = 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < m; c += kLT) {
          const uint64_t kc = s_cand[t * kCand + c];
          uint32_t less = 0, eq = 0;
          for (uint32_t o = 0; o < m; ++o) {
            const uint64_t ko = s_cand[t * kCand + o];
            less += ko < kc;
            eq += ko == kc;
          }
          if (less <= want && want < less + eq) {  // every member of the group qualifies and writes the same values
            s_misc[40] = c;
            s_misc[41] = less;
            s_misc[42] = eq;
          }
        }
        __syncthreads();
        const uint32_t c = s_misc[40];
        sel[t].value = s_cand[t * kCand + (c == 0xFFFFFFFFu ? 0 : c)];
        sel[t].n_less = sel[t].below + s_misc[41];
        sel[t].n_equal = s_misc[42];
        sel[t].done = 1;
      }
    }
  }
}

__device__ bool row_failed(const RowCounts *cnt, uint32_t row);

__global__ __launch_bounds__(kLT) void summary_large_kernel(const double *__restrict__ rows, uint32_t r1, uint32_t row0,
                                                            uint32_t req_len, uint32_t max_neighbours,
                                                            double *__restrict__ out_stats, uint32_t *__restrict__ out_n,
                                                            uint32_t *__restrict__ out_idx, double *__restrict__ out_dist,
                                                            double *__restrict__ out_z, const RowCounts *only_failed = nullptr) {
  if (only_failed && !row_failed(only_failed, blockIdx.x)) return;  // (round 3: the fallback of the two-pass path)
  __shared__ double s_w[kLT / 64];
  __shared__ uint32_t s_wu[kLT / 64];
  __shared__ uint32_t s_hist[kSel * kBins];
  __shared__ uint64_t s_cand[kSel * kCand];
  __shared__ uint32_t s_misc[64];
  __shared__ uint64_t s_mm[2 * (kLT / 64)];
  __shared__ double s_cd[kLargeMaxNb];
  __shared__ uint32_t s_ci[kLargeMaxNb];
  __shared__ uint32_t s_ti[kLargeMaxNb];  // columns of the tie group at the cut-off distance
  const double *row = rows + (uint64_t)blockIdx.x * r1;
  const uint32_t j = row0 + blockIdx.x;
  const uint32_t n = r1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // pass 1: sum (mean, lib/Matrix.ml:651-655) and the key range
  double part = 0.0;
  uint64_t kmin = ~0ull, kmax = 0;
  for (uint32_t i = threadIdx.x; i < n; i += kLT) {
    const double x = row[i];
    part = __dadd_rn(part, x);
    const uint64_t k = f64_key(x);
    kmin = min(kmin, k);
    kmax = max(kmax, k);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (uint64_t)__shfl_xor((unsigned long long)kmin, o, 64));
    kmax = max(kmax, (uint64_t)__shfl_xor((unsigned long long)kmax, o, 64));
  }
  if (lane == 0) {
    s_mm[wv] = kmin;
    s_mm[kLT / 64 + wv] = kmax;
  }
  const double mean = n ? block_sum(part, s_w) / (double)n : 0.0;  // block_sum synchronises
  for (int w = 0; w < kLT / 64; ++w) {
    kmin = min(kmin, s_mm[w]);
    kmax = max(kmax, s_mm[kLT / 64 + w]);
  }
  // upper median = element n/2 of the sorted row (:645-647), and the value that closes the neighbour list: groups are
  // added while eff_len < req_len (:648-649), so eff_len = (#less + #equal) of the element of rank req_len - 1.  The
  // squared deviations of the sample sd (:657-662,679-683) are added up by the first of these passes.
  double median = 0.0, mad = 0.0, sd = 0.0;
  uint32_t eff = n;
  Sel sel[kSel];
  if (n) {
    sel[0] = Sel{n / 2, kmin, kmax, 0, 0, 0, 0, 0};
    const bool cut = req_len < n;
    sel[1] = Sel{cut ? req_len - 1 : 0, kmin, kmax, 0, 0, 0, 0, cut ? 0 : 1};
    double sq = 0.0;
    block_select_ranks<0, true>(PlainRow{row}, n, mean, sel, 2, s_hist, s_cand, s_misc, &sq);
    const double ss = block_sum(sq, s_w);
    sd = (n > 1) ? sqrt(ss / ((double)n - 1.0)) : 0.0;
    median = key_f64(sel[0].value);
    if (cut) eff = sel[1].n_less + sel[1].n_equal;
  }
  // the M closest, by (distance, column)
  const uint32_t M = min(min(eff, max_neighbours), kLargeMaxNb);
  if (M) {
    uint64_t vkey;
    uint32_t n_lt;
    if (req_len < n && M == eff) {  // the list ends with the tie group just found
      vkey = sel[1].value;
      n_lt = sel[1].n_less;
    } else {  // the list is cut short (or holds everything): its last value is the element of rank M - 1
      Sel s3[1] = {Sel{M - 1, kmin, kmax, 0, 0, 0, 0, 0}};
      block_select_ranks<0>(PlainRow{row}, n, 0.0, s3, 1, s_hist, s_cand, s_misc);
      vkey = s3[0].value;
      n_lt = s3[0].n_less;
    }
    const double vM = key_f64(vkey);
    const uint32_t want_eq = M - n_lt;
    uint32_t got_lt = 0, got_eq = 0;
    for (uint32_t base = 0; base < n; base += kLT) {  // column order => ties resolved by column, as the multimap does
      const uint32_t i = base + threadIdx.x;
      const double dv = (i < n) ? row[i] : 0.0;
      const bool is_lt = (i < n) && dv < vM, is_eq = (i < n) && dv == vM;
      if (!__syncthreads_or(is_lt || (is_eq && got_eq < want_eq))) continue;  // nothing of interest in this stretch
      uint32_t tot_lt, tot_eq;
      const uint32_t p_lt = block_scan_flag(is_lt, s_wu, &tot_lt);
      const uint32_t p_eq = block_scan_flag(is_eq, s_wu, &tot_eq);
      if (is_lt) {
        s_cd[got_lt + p_lt] = dv;
        s_ci[got_lt + p_lt] = i;
      }
      if (is_eq && got_eq + p_eq < want_eq) s_ti[got_eq + p_eq] = i;
      got_lt += tot_lt;
      got_eq += tot_eq;
    }
    // sort the n_lt closer candidates by (distance, column), padded to a power of two with +inf
    uint32_t NP = 1;
    while (NP < n_lt) NP <<= 1;
    __syncthreads();
    for (uint32_t q = n_lt + threadIdx.x; q < NP; q += kLT) {
      s_cd[q] = __longlong_as_double(0x7FF0000000000000ll);
      s_ci[q] = 0xFFFFFFFFu;
    }
    for (uint32_t s = 2; s <= NP; s <<= 1)
      for (uint32_t t = s >> 1; t > 0; t >>= 1) {
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < NP / 2; q += kLT) {
          const uint32_t a = 2 * q - (q & (t - 1)), b = a + t;
          const bool asc = (a & s) == 0;
          const double da = s_cd[a], db = s_cd[b];
          const uint32_t ia = s_ci[a], ib = s_ci[b];
          const bool gt = (db < da) || (db == da && ib < ia);
          if (gt == asc) {
            s_cd[a] = db; s_cd[b] = da;
            s_ci[a] = ib; s_ci[b] = ia;
          }
        }
      }
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < M; q += kLT) {
      const double dq = (q < n_lt) ? s_cd[q] : vM;  // the tie group follows, already in column order
      out_idx[(uint64_t)j * max_neighbours + q] = (q < n_lt) ? s_ci[q] : s_ti[q - n_lt];
      out_dist[(uint64_t)j * max_neighbours + q] = dq;
      double zz = __dsub_rn(dq, mean) / sd;
      if (zz != zz) zz = __longlong_as_double((long long)0xFFF8000000000000ull);  // x86 invalid-operation NaN, see distance.hip
      out_z[(uint64_t)j * max_neighbours + q] = zz;
    }
  }
  // MAD = element n/2 of |d - median| (:671-678).  Its keys lie between 0 and the larger of the two distances from the
  // median to the ends of the row's range: known without a pass
  if (n) {
    const double far = fmax(fabs(__dsub_rn(key_f64(kmax), median)), fabs(__dsub_rn(key_f64(kmin), median)));
    Sel sm[1] = {Sel{n / 2, f64_key(0.0), f64_key(far), 0, 0, 0, 0, 0}};
    block_select_ranks<1>(PlainRow{row}, n, median, sm, 1, s_hist, s_cand, s_misc);
    mad = key_f64(sm[0].value);
  }
  if (threadIdx.x == 0) {
    out_stats[(uint64_t)j * 4 + 0] = mean;
    out_stats[(uint64_t)j * 4 + 1] = sd;
    out_stats[(uint64_t)j * 4 + 2] = median;
    out_stats[(uint64_t)j * 4 + 3] = mad;
    out_n[j] = eff;
  }
}

// ===========================================================================
// Round 3: the same summaries in TWO passes over the distance rows instead of eight to ten.
//
// A row of 10^6 distances does not fit anything on chip, so every selection above walks it from HBM several times, and
// one block per row leaves most of the chip idle.  Here the order statistics are BRACKETED first, from a sample:
//
//   sample   (one block a row; reads 6 % of it)  65,536 elements in runs of 1,024: their mean m^, and by exact selection
//            on the sample the keys (lo, hi) that bracket the row's median with six sigmas to spare (+-1.2 % of the ranks)
//            and the key `cut` below which, generously, lie the keep_at_most closest;
//   pass 1   (all CUs: the row cut into slices of 32,768)  every element once: counted below lo / at lo / at hi, appended
//            to the row's candidate list when strictly between, appended to its neighbour list when <= cut; sum d and
//            sum (d - m^)^2 per slice;
//   finish 1 (one block a row)  mean and sd from the slices' sums in slice order (sum (d - mean)^2 = sum (d - m^)^2 -
//            n (mean - m^)^2: m^ is within sd / 256 of the mean, nothing cancels); the EXACT median by selection among
//            the ~2 % candidates at the rank the counts leave open; the exact cut value, eff_len and the sorted neighbours
//            from the neighbour list; then the sample again, for the bracket of the MAD (|d - median|);
//   pass 2   (all CUs)  |d - median| counted / collected the same way;
//   finish 2 the exact MAD.
//
// Every result is the exact order statistic -- the sample only decides what is collected.  A row whose brackets miss
// (the rank falls outside them: a database in an adversarial order), whose lists overflow (a tie group of tens of thousands
// at the median or the cut) or whose keep_at_most exceeds what the lists hold is flagged and redone by the kernel above.
// HBM traffic: the rows once written, twice read.
// ===========================================================================
#ifdef KPOP_SUMMARY_STAMPS  // development only (tools/probes): phase clocks of block 0 of the one-block-a-row kernels
__device__ unsigned long long g_sum_stamps[32];
#define KPOP_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sum_stamps[i] = __builtin_readcyclecounter(); } while (0)
#define KPOP_WHY(c) do { if (threadIdx.x == 0) { atomicAdd(&g_sum_stamps[24 + (c)], 1ull); g_sum_stamps[31] = blockIdx.x; } } while (0)  // why a row was left to the fall-back
#else
#define KPOP_STAMP(i) do { } while (0)
#define KPOP_WHY(c) do { } while (0)
#endif

constexpr uint32_t kSample = 65536, kSlice = 32768, kNbSort = 4096;  // (kNbCap: summary_types.h)
// the brackets hold ~2.3 % of a row (six sigmas of a 65,536-element sample's quantile, both sides): room for 6 %, at least 65,536
static inline uint32_t cand_cap_for(uint32_t r1) { return std::max<uint32_t>(65536u, ((r1 / 16 + 4095u) & ~4095u)); }

// (RowInfo, RowCounts: summary_types.h)

__device__ __forceinline__ uint32_t sample_count(uint32_t n, uint32_t *stride) {
  if (n <= kSample) {
    *stride = kSampleRun;
    return n;
  }
  const uint32_t runs = kSample / kSampleRun;
  *stride = n / runs;  // >= kSampleRun
  return runs * kSampleRun;
}

// brackets of rank r of n from a sample of s: sample ranks r s / n -+ 6 sqrt(s) / 2 (six standard deviations of a sample
// quantile's rank), clamped
__device__ __forceinline__ void bracket_ranks(uint32_t r, uint32_t n, uint32_t s, uint32_t *a, uint32_t *b) {
  const double c = (double)r * (double)s / (double)n, w = 3.0 * sqrt((double)s) + 2.0;
  const double lo = c - w, hi = c + w;
  *a = lo <= 0.0 ? 0u : (uint32_t)lo;
  *b = hi >= (double)(s - 1) ? s - 1 : (uint32_t)hi;
}

__global__ __launch_bounds__(kLT) void summary2_sample_kernel(const double *__restrict__ rows, uint32_t r1, uint32_t req_len,
                                                              RowInfo *__restrict__ info, RowCounts *__restrict__ cnt) {
  __shared__ double s_w[kLT / 64];
  __shared__ uint32_t s_hist[kSel * kBins];
  __shared__ uint64_t s_cand[kSel * kCand];
  __shared__ uint32_t s_misc[64];
  __shared__ uint64_t s_mm[2 * (kLT / 64)];
  const double *row = rows + (uint64_t)blockIdx.x * r1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t stride;
  const uint32_t s = sample_count(r1, &stride);
  const SampledRow sr{row, stride};
  double part = 0.0;
  uint64_t kmin = ~0ull, kmax = 0;
  for (uint32_t i = threadIdx.x; i < s; i += kLT) {
    const double x = sr[i];
    part = __dadd_rn(part, x);
    const uint64_t k = f64_key(x);
    kmin = min(kmin, k);
    kmax = max(kmax, k);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (uint64_t)__shfl_xor((unsigned long long)kmin, o, 64));
    kmax = max(kmax, (uint64_t)__shfl_xor((unsigned long long)kmax, o, 64));
  }
  if (lane == 0) {
    s_mm[wv] = kmin;
    s_mm[kLT / 64 + wv] = kmax;
  }
  const double m_hat = s ? block_sum(part, s_w) / (double)s : 0.0;
  for (int w = 0; w < kLT / 64; ++w) {
    kmin = min(kmin, s_mm[w]);
    kmax = max(kmax, s_mm[kLT / 64 + w]);
  }
  uint32_t a, b;
  bracket_ranks(r1 / 2, r1, s, &a, &b);
  Sel sel[kSel] = {Sel{a, kmin, kmax, 0, 0, 0, 0, 0}, Sel{b, kmin, kmax, 0, 0, 0, 0, 0}};
  block_select_ranks<0, false, SampledRow>(sr, s, 0.0, sel, 2, s_hist, s_cand, s_misc);
  // the neighbour threshold: three times the sample rank the keep_at_most-th closest would have, and a margin; a bracket
  // that reaches the sample's end takes everything (the whole row is the list: only small rows can afford that)
  uint64_t kcut = ~0ull;
  {
    const double want = 3.0 * (double)req_len * (double)s / (double)r1 + 16.0;
    if (want < (double)(s - 1)) {
      Sel sc[1] = {Sel{(uint32_t)want, kmin, kmax, 0, 0, 0, 0, 0}};
      block_select_ranks<0, false, SampledRow>(sr, s, 0.0, sc, 1, s_hist, s_cand, s_misc);
      kcut = sc[0].value;
    }
  }
  if (threadIdx.x == 0) {
    RowInfo &I = info[blockIdx.x];
    I.m_hat = m_hat;
    I.klo = a == 0 ? 0ull : sel[0].value;           // rank 0 of the sample bounds nothing from below
    I.khi = b == s - 1 ? ~0ull : sel[1].value;
    I.kcut = kcut;
    I.sample_n = s;
    I.sample_stride = stride;
    RowCounts z = {};
    cnt[blockIdx.x] = z;
  }
}

// append to a per-row list with one atomic per wave
__device__ __forceinline__ uint32_t wave_append(bool want, uint32_t *counter, int lane) {
  const uint64_t m = __ballot(want);
  uint32_t base = 0;
  if (m) {
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
  }
  return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// pass 1 (PASS = 1) and pass 2 (PASS = 2) over slice blockIdx.x of row blockIdx.y
template <int PASS>
__global__ __launch_bounds__(256) void summary2_pass_kernel(const double *__restrict__ rows, uint32_t r1, const RowInfo *__restrict__ info,
                                                            RowCounts *__restrict__ cnt, double *__restrict__ cand, uint32_t *__restrict__ nb_idx,
                                                            double *__restrict__ nb_d, double *__restrict__ part, uint32_t n_slices, uint32_t kCandCap) {
  __shared__ uint32_t s_c[4];
  __shared__ double s_p[2][4];
  // the block's candidates wait in LDS and go to the row's list in one piece: a global atomic per wave and iteration on the
  // row's ONE counter (124 waves a row) was what the pass waited for -- 1.17 ms against the 0.45 its 2 GB take to read
  constexpr uint32_t kStage = 4096;
  __shared__ double s_stage[kStage];
  __shared__ uint32_t s_n, s_base;
  const uint32_t j = blockIdx.y, sl = blockIdx.x;
  const double *row = rows + (uint64_t)j * r1;
  const RowInfo I = info[j];
  RowCounts *C = cnt + j;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t i0 = sl * kSlice, i1 = min(r1, i0 + kSlice);
  const uint64_t lo = PASS == 1 ? I.klo : I.mlo, hi = PASS == 1 ? I.khi : I.mhi;
  double *my_cand = cand + (uint64_t)j * kCandCap;
  uint32_t c_lt = 0, c_eqlo = 0, c_eqhi = 0;
  double sum = 0.0, sq = 0.0;
  constexpr int U = 8;  // loads in flight per thread
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  auto flush = [&]() {  // all threads; s_n is stable (between barriers)
    const uint32_t cnt_ = s_n;
    if (threadIdx.x == 0) s_base = atomicAdd(PASS == 1 ? &C->n_cand : &C->m_cand, cnt_);
    __syncthreads();
    const uint32_t b0 = s_base;
    for (uint32_t q = threadIdx.x; q < cnt_; q += 256)
      if (b0 + q < kCandCap) my_cand[b0 + q] = s_stage[q];
    __syncthreads();
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
  };
  for (uint32_t base = i0; base < i1; base += 256 * U) {
    if (s_n > kStage - 256 * U) flush();  // (block-uniform: read between barriers) room for a whole iteration of candidates
    double dv8[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = base + u * 256 + threadIdx.x;
      dv8[u] = i < i1 ? row[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = base + u * 256 + threadIdx.x;
      const bool ok = i < i1;
      const double d = dv8[u];
      const double x = PASS == 1 ? d : fabs(__dsub_rn(d, I.median));
      const uint64_t k = f64_key(x);
      if (ok) {
        c_lt += k < lo;
        c_eqlo += k == lo;
        c_eqhi += (k == hi) && hi != lo;
        if (PASS == 1) {
          sum = __dadd_rn(sum, d);
          const double dv = __dsub_rn(d, I.m_hat);
          sq = __dadd_rn(sq, __dmul_rn(dv, dv));
        }
      }
      const bool is_cand = ok && k > lo && k < hi;
      if (__ballot(is_cand)) {
        const uint32_t at = wave_append(is_cand, &s_n, lane);
        if (is_cand) s_stage[at] = x;
      }
      if (PASS == 1) {
        const bool is_nb = ok && k <= I.kcut;
        if (__ballot(is_nb)) {
          const uint32_t at = wave_append(is_nb, &C->n_nb, lane);
          if (is_nb && at < kNbCap) {
            nb_idx[(uint64_t)j * kNbCap + at] = i;
            nb_d[(uint64_t)j * kNbCap + at] = d;
          }
        }
      }
    }
    __syncthreads();  // s_n is read at the top of the next iteration
  }
  if (s_n) flush();
  // the block's counts: one atomic each; its sums: one slot each (added up in slice order by the finish kernel)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    c_lt += __shfl_xor((int)c_lt, o, 64);
    c_eqlo += __shfl_xor((int)c_eqlo, o, 64);
    c_eqhi += __shfl_xor((int)c_eqhi, o, 64);
    if (PASS == 1) {
      sum = __dadd_rn(sum, __shfl_xor(sum, o, 64));
      sq = __dadd_rn(sq, __shfl_xor(sq, o, 64));
    }
  }
  if (threadIdx.x < 4) s_c[threadIdx.x] = 0;
  __syncthreads();
  if (lane == 0) {
    if (c_lt) atomicAdd(&s_c[0], c_lt);
    if (c_eqlo) atomicAdd(&s_c[1], c_eqlo);
    if (c_eqhi) atomicAdd(&s_c[2], c_eqhi);
    s_p[0][wv] = sum;
    s_p[1][wv] = sq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_c[0]) atomicAdd(PASS == 1 ? &C->lt_lo : &C->m_lt, s_c[0]);
    if (s_c[1]) atomicAdd(PASS == 1 ? &C->eq_lo : &C->m_eqlo, s_c[1]);
    if (s_c[2]) atomicAdd(PASS == 1 ? &C->eq_hi : &C->m_eqhi, s_c[2]);
    if (PASS == 1) {
      part[((uint64_t)j * n_slices + sl) * 2 + 0] = __dadd_rn(__dadd_rn(__dadd_rn(s_p[0][0], s_p[0][1]), s_p[0][2]), s_p[0][3]);
      part[((uint64_t)j * n_slices + sl) * 2 + 1] = __dadd_rn(__dadd_rn(__dadd_rn(s_p[1][0], s_p[1][1]), s_p[1][2]), s_p[1][3]);
    }
  }
}

// the exact order statistic of rank r from what a pass left: counts below / at the bracket's ends and the values
// strictly inside it.  false: the bracket missed or the list overflowed.
__device__ bool pick_from_bracket(uint32_t r, uint32_t lt, uint32_t eqlo, uint32_t n_cand, uint32_t eqhi, uint64_t klo, uint64_t khi,
                                  const double *cand, uint32_t kCandCap, uint32_t *s_hist, uint64_t *s_cand, uint32_t *s_misc, double *out) {
  if (r < lt) return false;
  if (r < lt + eqlo) {
    *out = key_f64(klo);
    return true;
  }
  if (r < lt + eqlo + n_cand) {
    if (n_cand > kCandCap) return false;
    // the list's own range: its keys lie strictly between the bracket's
    Sel sel[1] = {Sel{r - lt - eqlo, klo, khi, 0, 0, 0, 0, 0}};
    block_select_ranks<0>(PlainRow{cand}, n_cand, 0.0, sel, 1, s_hist, s_cand, s_misc);
    *out = key_f64(sel[0].value);
    return true;
  }
  if (r < lt + eqlo + n_cand + eqhi) {
    *out = key_f64(khi);
    return true;
  }
  return false;
}

// the neighbours of a row from its list of every element with key <= kcut (the n_nb smallest): eff_len (lib/Matrix.ml:648-649),
// the first min(eff_len, max_neighbours, 2,048) by (distance, column) and their z-scores written out.  false: the list
// overflowed, is shorter than what is asked for, or ends in a tie group too large to sort here.
__device__ bool neighbours_from_list(uint32_t n, uint32_t req_len, uint32_t max_neighbours, uint32_t n_nb, const uint32_t *my_idx,
                                     const double *my_d, uint64_t kcut, double mean, double sd, uint32_t j, uint32_t *__restrict__ out_idx,
                                     double *__restrict__ out_dist, double *__restrict__ out_z, uint32_t *s_hist, uint64_t *s_cand,
                                     uint32_t *s_misc, double *s_cd, uint32_t *s_ci, uint32_t *s_take, uint32_t *eff_out) {
  // ---- the neighbours: the list holds every element with key <= kcut, i.e. the n_nb smallest
  uint32_t eff = n;
  bool ok = true;
  if (n_nb > kNbCap || (n_nb < n && req_len > n_nb) || (n_nb == n && n > kNbCap)) ok = false;
  // The usual case first: the whole list fits the sort buffer (it holds ~3 req_len + 16 x the row's share of the sample) -- sorted
  // whole by (distance, column), the effective length and the first M read off the sorted list.  The same answers as the selections
  // below (which remain for longer lists), without their passes: 276k of this kernel's 790k cycles went into a list of 1,029.
  if (ok && req_len < n && n_nb >= 1 && n_nb <= kNbSort) {
    __syncthreads();
    if (threadIdx.x == 0) (*s_take) = 0;
    __syncthreads();
    uint32_t NP = 1;
    while (NP < n_nb) NP <<= 1;
    for (uint32_t q = threadIdx.x; q < NP; q += kLT) {
      const double d = q < n_nb ? my_d[q] : __longlong_as_double(0x7FF0000000000000ll);
      s_cd[q] = d;
      s_ci[q] = q < n_nb ? my_idx[q] : 0xFFFFFFFFu;
      // (a NaN, or a -0.0 -- below +0.0 as a key, equal to it as a number --: the selections' way, which speaks in keys)
      if (q < n_nb && (d != d || __double_as_longlong(d) == (long long)0x8000000000000000ull)) (*s_take) = 1;
    }
    __syncthreads();
    const bool plain = (*s_take) == 0;
    if (plain) {
      for (uint32_t sz = 2; sz <= NP; sz <<= 1)
        for (uint32_t t = sz >> 1; t > 0; t >>= 1) {
          __syncthreads();
          for (uint32_t q = threadIdx.x; q < NP / 2; q += kLT) {
            const uint32_t a = 2 * q - (q & (t - 1)), b = a + t;
            const bool asc = (a & sz) == 0;
            const double da = s_cd[a], db = s_cd[b];
            const uint32_t ia = s_ci[a], ib = s_ci[b];
            const bool gt = (db < da) || (db == da && ib < ia);
            if (gt == asc) {
              s_cd[a] = db; s_cd[b] = da;
              s_ci[a] = ib; s_ci[b] = ia;
            }
          }
        }
      __syncthreads();
      const double v = s_cd[req_len - 1];  // (req_len <= n_nb: checked above)
      for (uint32_t q = threadIdx.x; q < n_nb; q += kLT)
        if (s_cd[q] <= v && (q + 1 == n_nb || s_cd[q + 1] > v)) (*s_take) = q + 1;  // the one place where the values at or below v end
      __syncthreads();
      eff = (*s_take);
      const uint32_t M = min(min(eff, max_neighbours), kLargeMaxNb);
      for (uint32_t q = threadIdx.x; q < M; q += kLT) {
        out_idx[(uint64_t)j * max_neighbours + q] = s_ci[q];
        out_dist[(uint64_t)j * max_neighbours + q] = s_cd[q];
        double zz = __dsub_rn(s_cd[q], mean) / sd;
        if (zz != zz) zz = __longlong_as_double((long long)0xFFF8000000000000ull);  // x86 invalid-operation NaN, see distance.hip
        out_z[(uint64_t)j * max_neighbours + q] = zz;
      }
      __syncthreads();
      *eff_out = eff;
      return true;
    }
  }
  if (ok) {
    uint32_t M;
    uint64_t vkey = ~0ull;
    const uint64_t k_top = kcut == ~0ull ? f64_key(__longlong_as_double(0x7FEFFFFFFFFFFFFFll)) : kcut;
    if (req_len < n) {  // (always, here: the two-pass path is taken for keep_at_most <= 2,048 and rows of 65,536 and more)
      Sel sc[1] = {Sel{req_len - 1, 0ull, k_top, 0, 0, 0, 0, 0}};  // (from key 0: a caller's matrix may hold negative entries)
      block_select_ranks<0>(PlainRow{my_d}, n_nb, 0.0, sc, 1, s_hist, s_cand, s_misc);
      eff = sc[0].n_less + sc[0].n_equal;
      vkey = sc[0].value;
    }
    M = min(min(eff, max_neighbours), kLargeMaxNb);
    if (M && M < eff) {  // the list is cut short by the caller's stride: its last value is the element of rank M - 1
      Sel s3[1] = {Sel{M - 1, 0ull, k_top, 0, 0, 0, 0, 0}};
      block_select_ranks<0>(PlainRow{my_d}, n_nb, 0.0, s3, 1, s_hist, s_cand, s_misc);
      vkey = s3[0].value;
    }
    if (M) {
      // gather the elements closer than the last value, and those AT it; sort by (distance, column); the first M
      __syncthreads();
      if (threadIdx.x == 0) (*s_take) = 0;
      __syncthreads();
      for (uint32_t q = threadIdx.x; q < n_nb; q += kLT) {
        const uint64_t k = f64_key(my_d[q]);
        if (k <= vkey) {
          const uint32_t at = atomicAdd(&(*s_take), 1u);
          if (at < kNbSort) {
            s_cd[at] = my_d[q];
            s_ci[at] = my_idx[q];
          }
        }
      }
      __syncthreads();
      const uint32_t got = (*s_take);
      if (got > kNbSort || got < M) ok = false;  // (a tie group too large to sort here)
      else {
        uint32_t NP = 1;
        while (NP < got) NP <<= 1;
        for (uint32_t q = got + threadIdx.x; q < NP; q += kLT) {
          s_cd[q] = __longlong_as_double(0x7FF0000000000000ll);
          s_ci[q] = 0xFFFFFFFFu;
        }
        for (uint32_t sz = 2; sz <= NP; sz <<= 1)
          for (uint32_t t = sz >> 1; t > 0; t >>= 1) {
            __syncthreads();
            for (uint32_t q = threadIdx.x; q < NP / 2; q += kLT) {
              const uint32_t a = 2 * q - (q & (t - 1)), b = a + t;
              const bool asc = (a & sz) == 0;
              const double da = s_cd[a], db = s_cd[b];
              const uint32_t ia = s_ci[a], ib = s_ci[b];
              const bool gt = (db < da) || (db == da && ib < ia);
              if (gt == asc) {
                s_cd[a] = db; s_cd[b] = da;
                s_ci[a] = ib; s_ci[b] = ia;
              }
            }
          }
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < M; q += kLT) {
          out_idx[(uint64_t)j * max_neighbours + q] = s_ci[q];
          out_dist[(uint64_t)j * max_neighbours + q] = s_cd[q];
          double zz = __dsub_rn(s_cd[q], mean) / sd;
          if (zz != zz) zz = __longlong_as_double((long long)0xFFF8000000000000ull);  // x86 invalid-operation NaN, see distance.hip
          out_z[(uint64_t)j * max_neighbours + q] = zz;
        }
      }
    }
  }
  *eff_out = eff;
  return ok;
}

// finish 1 (STAGE = 1): mean, sd, median, neighbours, the MAD's bracket; finish 2 (STAGE = 2): the MAD and the row of statistics
template <int STAGE>
__global__ __launch_bounds__(kLT) void summary2_finish_kernel(const double *__restrict__ rows, uint32_t r1, uint32_t row0, uint32_t req_len,
                                                              uint32_t max_neighbours, RowInfo *__restrict__ info, RowCounts *__restrict__ cnt,
                                                              const double *__restrict__ cand, const uint32_t *__restrict__ nb_idx,
                                                              const double *__restrict__ nb_d, const double *__restrict__ part, uint32_t n_slices,
                                                              uint32_t kCandCap, double *__restrict__ out_stats, uint32_t *__restrict__ out_n,
                                                              uint32_t *__restrict__ out_idx, double *__restrict__ out_dist, double *__restrict__ out_z) {
  __shared__ uint32_t s_hist[kSel * kBins];
  __shared__ uint64_t s_cand[kSel * kCand];
  __shared__ uint32_t s_misc[64];
  __shared__ double s_cd[kNbSort];
  __shared__ uint32_t s_ci[kNbSort];
  __shared__ uint32_t s_take;
  const uint32_t jl = blockIdx.x, j = row0 + jl, n = r1;
  RowInfo &I = info[jl];
  RowCounts &C = cnt[jl];
  if (C.fail) return;
  const double *my_cand = cand + (uint64_t)jl * kCandCap;
  if (STAGE == 2) {
    double mad = 0.0;
    const bool ok = pick_from_bracket(n / 2, C.m_lt, C.m_eqlo, C.m_cand, C.m_eqhi, I.mlo, I.mhi, my_cand, kCandCap, s_hist, s_cand, s_misc, &mad);
    if (threadIdx.x == 0) {
      if (!ok) C.fail = 1;
      else {
        out_stats[(uint64_t)j * 4 + 0] = I.mean;
        out_stats[(uint64_t)j * 4 + 1] = I.sd;
        out_stats[(uint64_t)j * 4 + 2] = I.median;
        out_stats[(uint64_t)j * 4 + 3] = mad;
      }
    }
    return;
  }
  // ---- mean and sd: the slices' sums in slice order (every thread the same chain: a few dozen additions)
  double sum = 0.0, sqh = 0.0;
  for (uint32_t s = 0; s < n_slices; ++s) {
    sum = __dadd_rn(sum, part[((uint64_t)jl * n_slices + s) * 2 + 0]);
    sqh = __dadd_rn(sqh, part[((uint64_t)jl * n_slices + s) * 2 + 1]);
  }
  const double mean = sum / (double)n;
  const double dm = __dsub_rn(mean, I.m_hat);
  const double ss = fmax(0.0, __dsub_rn(sqh, __dmul_rn((double)n, __dmul_rn(dm, dm))));
  const double sd = n > 1 ? sqrt(ss / ((double)n - 1.0)) : 0.0;
  // ---- the median
  double median = 0.0;
  bool ok = pick_from_bracket(n / 2, C.lt_lo, C.eq_lo, C.n_cand, C.eq_hi, I.klo, I.khi, my_cand, kCandCap, s_hist, s_cand, s_misc, &median);
  // ---- the neighbours: the list holds every element with key <= kcut, i.e. the n_nb smallest
  uint32_t eff = n;
  if (ok)
    ok = neighbours_from_list(n, req_len, max_neighbours, C.n_nb, nb_idx + (uint64_t)jl * kNbCap, nb_d + (uint64_t)jl * kNbCap, I.kcut, mean, sd, j,
                              out_idx, out_dist, out_z, s_hist, s_cand, s_misc, s_cd, s_ci, &s_take, &eff);
  // ---- the bracket of the MAD: the sample's |d - median| at the ranks around the middle
  uint64_t mlo = 0, mhi = ~0ull;
  if (ok) {
    const SampledRow sr{rows + (uint64_t)jl * r1, I.sample_stride};
    const uint32_t s = I.sample_n;
    uint32_t a, b;
    bracket_ranks(n / 2, n, s, &a, &b);
    // keys of |d - median| over the sample lie in [key(0), key(max |..|)]: the selection narrows from the full non-negative range
    Sel sel[kSel] = {Sel{a, f64_key(0.0), f64_key(__longlong_as_double(0x7FEFFFFFFFFFFFFFll)), 0, 0, 0, 0, 0},
                     Sel{b, f64_key(0.0), f64_key(__longlong_as_double(0x7FEFFFFFFFFFFFFFll)), 0, 0, 0, 0, 0}};
    block_select_ranks<1, false, SampledRow>(sr, s, median, sel, 2, s_hist, s_cand, s_misc);
    mlo = a == 0 ? 0ull : sel[0].value;
    mhi = b == s - 1 ? ~0ull : sel[1].value;
  }
  if (threadIdx.x == 0) {
    if (!ok) C.fail = 1;
    I.mean = mean;
    I.sd = sd;
    I.median = median;
    I.mlo = mlo;
    I.mhi = mhi;
    if (ok) out_n[j] = eff;
  }
}

__device__ bool row_failed(const RowCounts *cnt, uint32_t row) { return cnt[row].fail != 0; }

// ===========================================================================
// Round 3, second step: the summary of q query rows against r1 >= 131,072 reference rows WITHOUT the q x r1 distances in HBM.
//
// The two-pass path above still has the tiled distance kernel write every row (2 GB for 256 x 10^6) and reads it twice.
// Here the brackets come first, from the distances to a SAMPLE of the reference rows (65,536 rows at even spacing, gathered
// and run through the tiled kernel: 6 % of the work), and then ONE kernel computes every distance -- the same sequential chain
// per pair as distance_rowwise_kernel, so the same bits -- and decides on the spot what to keep of it:
//   - counted: below / at the ends of the median's bracket, strictly inside it; strictly inside the INNER region of the MAD;
//   - added up: d and (d - m^)^2 (per thread and row across the tiles, then in a fixed order);
//   - kept (8 bytes, in the row's segment of the stripe): the candidates of the median (strictly inside its bracket) and of
//     the MAD -- the two BANDS either side of the median where |d - median| can reach the MAD's bracket whatever the median
//     turns out to be inside its own bracket; about 13 % of a row;
//   - kept with its column: d <= cut, the neighbours' list.
// A block owns 256 query rows x a STRIPE of 2,048 reference rows (64 tiles of 32), so a (row, stripe) segment has one
// writer and no global counter: its count is written once, at the end.  The finish kernel (one block a row) compacts the
// segments, finds the exact median among its candidates at the rank the counts leave open, then the exact MAD: with the
// median known, every element of the inner region has |d - median| <= X_in = max(U_in - median, median - L_in) and every
// element beyond the bands >= X_out = min(median - L_lo, U_hi - median); the candidate a' of rank n/2 - n_inner among the
// bands' |d - median| IS the MAD if X_in <= a' <= X_out (checked: a certificate, not an assumption).  Anything that does not
// hold -- a bracket that missed, a list that overflowed, the certificate -- flags the row, and flagged rows are redone from
// distance rows computed then (the tiled kernel and summary_large_kernel, both launched always and gated on the flags).
// ===========================================================================
constexpr uint32_t kFW = 32, kFQ = 256, kFDC = 16;  // (kStripe, kMaxStripes, StripeRec, FusedThr: summary_types.h)

static inline uint32_t fused_cand_cap(uint32_t r1) { return (r1 / 4 + 4095u) & ~4095u; }  // compacted candidates: room for a quarter of a row

// out[i] = a[floor(i r1 / s)]: the sample of the reference rows, at even spacing (a database sorted by class is sampled
// class by class in proportion)
__global__ __launch_bounds__(256) void sample_gather_kernel(const double *__restrict__ a, uint32_t r1, uint32_t n_dims, uint32_t s,
                                                            double *__restrict__ out) {
  const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= s) return;
  const uint64_t src = ((uint64_t)i * r1) / s;
  for (uint32_t c = lane; c < n_dims; c += 64) out[(uint64_t)i * n_dims + c] = a[src * n_dims + c];
}

// one block per query row over s sample distances: m^, the brackets, the bands.  SAMPLED = false: `src` holds [rows][s]
// distances to a sample of the reference rows (the one-kernel path); true: `src` holds the distance rows themselves
// ([rows][r1]) and the sample is kSampleRun consecutive elements out of every `stride` (the one-pass path over rows)
template <bool SAMPLED>
__global__ __launch_bounds__(kLT) void fused_sample_kernel(const double *__restrict__ src, uint32_t s_plain, uint32_t r1, uint32_t req_len,
                                                           RowInfo *__restrict__ info, RowCounts *__restrict__ cnt, FusedThr *__restrict__ thr) {
  __shared__ double s_w[kLT / 64];
  __shared__ uint32_t s_hist[kSelMax * kBins];  // (four selections share the first scans: 96 KB of tables)
  __shared__ uint64_t s_cand[kSelMax * kCand];
  __shared__ uint32_t s_misc[64];
  __shared__ uint64_t s_mm[2 * (kLT / 64)];
  uint32_t stride = 0;
  const uint32_t s = SAMPLED ? sample_count(r1, &stride) : s_plain;
  using Row = typename std::conditional<SAMPLED, SampledRow, PlainRow>::type;
  Row sr;
  if constexpr (SAMPLED) sr = SampledRow{src + (uint64_t)blockIdx.x * r1, stride};
  else sr = PlainRow{src + (uint64_t)blockIdx.x * s};
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  KPOP_STAMP(16);
  double part = 0.0;
  uint64_t kmin = ~0ull, kmax = 0;
  for (uint32_t i0 = threadIdx.x; i0 < s; i0 += kLT * 8) {  // (eight loads in flight; the sum in the order one load a turn gave)
    double x8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x8[u] = i0 + u * kLT < s ? sr[i0 + u * kLT] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (i0 + u * kLT >= s) continue;
      const double x = x8[u];
      part = __dadd_rn(part, x);
      const uint64_t k = f64_key(x);
      kmin = kmin < k ? kmin : k;
      kmax = kmax > k ? kmax : k;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint64_t omin = (uint64_t)__shfl_xor((unsigned long long)kmin, o, 64), omax = (uint64_t)__shfl_xor((unsigned long long)kmax, o, 64);
    kmin = kmin < omin ? kmin : omin;
    kmax = kmax > omax ? kmax : omax;
  }
  if (lane == 0) {
    s_mm[wv] = kmin;
    s_mm[kLT / 64 + wv] = kmax;
  }
  const double m_hat = s ? block_sum(part, s_w) / (double)s : 0.0;
  for (int w = 0; w < kLT / 64; ++w) {
    kmin = kmin < s_mm[w] ? kmin : s_mm[w];
    kmax = kmax > s_mm[kLT / 64 + w] ? kmax : s_mm[kLT / 64 + w];
  }
  KPOP_STAMP(17);
  uint32_t a, b;
  bracket_ranks(r1 / 2, r1, s, &a, &b);
  // A selection narrows [lo, hi] by bins that are linear in KEY space -- logarithmic in the value.  One sample far below the rest
  // (the query's distance to ITSELF when its own column is among the sampled ones, a handful of near-duplicates: 0 or 1e-8 beside
  // values around 1.4) stretches the first range over hundreds of binades, the whole sample lands in one or two bins and every
  // selection goes a round or more longer: 0.9 -> 2.7 ms per 1,024 rows of BASELINE config 4's all-vs-all step.  So: where the
  // smallest sample is below a sixteenth of the mean, the samples below that mark are counted (`below` of the selections that
  // start above them) and the range starts at the smallest sample above it.
  uint64_t klow = kmin;
  uint32_t n_out = 0;
  if (s > 64 && m_hat > 0.0 && key_f64(kmin) < m_hat * 0.0625) {  // (uniform)
    const uint64_t kmark = f64_key(m_hat * 0.0625);
    uint64_t k2 = ~0ull;
    uint32_t c = 0;
    for (uint32_t i0 = threadIdx.x; i0 < s; i0 += kLT * 8) {
      double x8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x8[u] = i0 + u * kLT < s ? sr[i0 + u * kLT] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (i0 + u * kLT >= s) continue;
        const uint64_t k = f64_key(x8[u]);
        if (k < kmark) ++c;
        else k2 = k2 < k ? k2 : k;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t o2 = (uint64_t)__shfl_xor((unsigned long long)k2, o, 64);
      k2 = k2 < o2 ? k2 : o2;
      c += (uint32_t)__shfl_xor((int)c, o, 64);
    }
    __syncthreads();
    if (lane == 0) {
      s_mm[wv] = k2;
      s_mm[kLT / 64 + wv] = c;
    }
    __syncthreads();
    klow = ~0ull;
    for (int w = 0; w < kLT / 64; ++w) {
      klow = klow < s_mm[w] ? klow : s_mm[w];
      n_out += (uint32_t)s_mm[kLT / 64 + w];
    }
    __syncthreads();
    if (klow == ~0ull || klow > kmax) {  // (nothing above the mark: as before)
      klow = kmin;
      n_out = 0;
    }
  }
  KPOP_STAMP(18);
  auto from_low = [&](uint32_t rank, int done) { return rank >= n_out ? Sel{rank, klow, kmax, n_out, 0, 0, 0, done} : Sel{rank, kmin, kmax, 0, 0, 0, 0, done}; };
  // the bracket's two ranks, the sample's own median and the neighbour threshold: four ranks of the same keys, ONE set of passes over the
  // sample (they were two sets of two: every pass streams the sample from HBM again, 134 MB for 256 rows)
  uint64_t kcut = ~0ull;
  const double want = 3.0 * (double)req_len * (double)s / (double)r1 + 16.0;
  const bool cut = want < (double)(s - 1);
  Sel s4[kSelMax] = {from_low(a, 0), from_low(b, 0), from_low(s / 2, 0), from_low(cut ? (uint32_t)want : 0u, cut ? 0 : 1)};
  block_select_ranks<0, false, Row>(sr, s, 0.0, s4, 4, s_hist, s_cand, s_misc);
  const Sel sel[2] = {s4[0], s4[1]}, sm[2] = {s4[2], s4[3]};
  KPOP_STAMP(19);
  KPOP_STAMP(20);
  if (cut) kcut = sm[1].value;
  const double ms = key_f64(sm[0].value);
  const double far = fmax(fabs(__dsub_rn(key_f64(kmax), ms)), fabs(__dsub_rn(key_f64(kmin), ms)));
  Sel sa[kSel] = {Sel{a, f64_key(0.0), f64_key(far), 0, 0, 0, 0, 0}, Sel{b, f64_key(0.0), f64_key(far), 0, 0, 0, 0, 0}};
  block_select_ranks<1, false, Row>(sr, s, ms, sa, 2, s_hist, s_cand, s_misc);
  KPOP_STAMP(21);
  if (threadIdx.x == 0) {
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    const bool no_lo = a == 0, no_hi = b == s - 1;  // rank 0 / the last rank of the sample bound nothing
    RowInfo &I = info[blockIdx.x];
    I.m_hat = m_hat;
    I.klo = no_lo ? 0ull : sel[0].value;
    I.khi = no_hi ? ~0ull : sel[1].value;
    I.kcut = kcut;
    I.sample_n = s;
    I.sample_stride = stride;
    RowCounts z = {};
    cnt[blockIdx.x] = z;
    FusedThr T;
    T.lo = no_lo ? -inf : key_f64(sel[0].value);
    T.hi = no_hi ? inf : key_f64(sel[1].value);
    T.cut = kcut == ~0ull ? inf : key_f64(kcut);
    T.mhat = m_hat;
    if (no_lo || no_hi) {  // no bracket to speak of: everything is a candidate
      T.Llo = -inf; T.Uhi = inf; T.Lin = inf; T.Uin = -inf;
    } else {
      const double alo = key_f64(sa[0].value), ahi = key_f64(sa[1].value);
      // the median lies within w of the sample's: |d - median| and |d - ms| differ by at most w, and so do their order statistics
      const double w = fmax(T.hi - ms, ms - T.lo);
      const double h = alo - w;  // |d - median| < h for every median in its bracket: inside (hi - h, lo + h)
      if (h > 0.0) { T.Lin = T.hi - h; T.Uin = T.lo + h; } else { T.Lin = inf; T.Uin = -inf; }
      T.Llo = T.lo - (ahi + w);
      T.Uhi = T.hi + (ahi + w);
    }
    thr[blockIdx.x] = T;
  }
}

// distances of 256 query rows to a stripe of 2,048 reference rows, tile by tile (32 columns x 256 rows; a thread owns 4 x 8,
// the dimensions 16 at a time through LDS: distance_rowwise_kernel's loop), and what the summary keeps of them
template <int KIND>
__global__ __launch_bounds__(256, 2) void summary_fused_pass_kernel(
    const double *__restrict__ a, uint32_t r1, const double *__restrict__ b, uint32_t q, uint32_t n_dims,
    const double *__restrict__ metric, double p, const FusedThr *__restrict__ thr, double *__restrict__ seg,
    StripeRec *__restrict__ rec, double *__restrict__ part, RowCounts *__restrict__ cnt, uint32_t *__restrict__ nb_idx,
    double *__restrict__ nb_d, uint32_t n_stripes) {
  __shared__ __attribute__((aligned(16))) double As[kFDC][kFW + 2];
  __shared__ __attribute__((aligned(16))) double Bs[kFDC][kFQ + 2];
  __shared__ double s_metric[kFDC];
  __shared__ FusedThr s_thr[kFQ];
  __shared__ uint32_t s_ccnt[kFQ];
  __shared__ double s_sum[kFQ], s_sq[kFQ];   // a row's running sums and counts: owned by the lane of its first column group
  __shared__ uint32_t s_cnt[kFQ][3];         // (the registers are the distances' -- as in distance_rowwise_kernel, 250 of 256)
  const uint32_t stripe = blockIdx.x, j0 = blockIdx.y * kFQ;
  const uint32_t cg = threadIdx.x & 7u, rg = threadIdx.x >> 3;  // 8 column groups x 32 row groups
  const uint32_t ti = cg * 4, tj = rg * 8;
  const uint32_t sc = threadIdx.x & 15u, rbase = threadIdx.x >> 4;  // staging: 16 dimensions x 16 rows a sweep
  for (uint32_t e = threadIdx.x; e < kFQ; e += 256) {
    s_thr[e] = thr[min(j0 + e, q - 1)];
    s_ccnt[e] = 0;
    s_sum[e] = 0.0;
    s_sq[e] = 0.0;
    s_cnt[e][0] = s_cnt[e][1] = s_cnt[e][2] = 0;
  }
  const uint32_t ref0 = stripe * kStripe, ref1 = min(r1, ref0 + kStripe);
  const uint32_t n_tiles = (ref1 - ref0 + kFW - 1) / kFW, n_chunks = (n_dims + kFDC - 1) / kFDC;
  double ra[2], rb[16];
  auto prefetch_a = [&](uint32_t tile, uint32_t c0) {  // the reference rows: from HBM
    const uint32_t i0 = ref0 + tile * kFW;
    const bool cok = c0 + sc < n_dims;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t row = rbase + u * 16;
      ra[u] = (cok && i0 + row < ref1) ? a[(uint64_t)(i0 + row) * n_dims + c0 + sc] : 0.0;
    }
  };
  auto prefetch_b = [&](uint32_t c0) {  // the query rows: the same 256 for every tile, from L2
    const bool cok = c0 + sc < n_dims;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const uint32_t row = rbase + u * 16;
      rb[u] = (cok && j0 + row < q) ? b[(uint64_t)(j0 + row) * n_dims + c0 + sc] : 0.0;
    }
  };
  prefetch_a(0, 0);
  for (uint32_t tile = 0; tile < n_tiles; ++tile) {
    double acc[8][4];
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
      for (int x = 0; x < 4; ++x) acc[y][x] = 0.0;
    prefetch_b(0);  // (after the epilogue: its sixteen registers a lane are the epilogue's until then)
    for (uint32_t ch = 0; ch < n_chunks; ++ch) {
      const uint32_t c0 = ch * kFDC;
      __syncthreads();  // the previous chunk's readers are done
#pragma unroll
      for (int u = 0; u < 2; ++u) As[sc][rbase + u * 16] = ra[u];
#pragma unroll
      for (int u = 0; u < 16; ++u) Bs[sc][rbase + u * 16] = rb[u];
      if (threadIdx.x < kFDC) s_metric[threadIdx.x] = (c0 + threadIdx.x < n_dims) ? metric[c0 + threadIdx.x] : 0.0;
      __syncthreads();
      if (ch + 1 < n_chunks) {
        prefetch_a(tile, c0 + kFDC);
        prefetch_b(c0 + kFDC);
      } else if (tile + 1 < n_tiles) prefetch_a(tile + 1, 0);  // (lands under this chunk's arithmetic and the epilogue)
      const uint32_t lim = min((uint32_t)kFDC, n_dims - c0);
      for (uint32_t cc = 0; cc < lim; ++cc) {
        double av[4], bv[8];
#pragma unroll
        for (int x = 0; x < 4; ++x) av[x] = As[cc][ti + x];
#pragma unroll
        for (int y = 0; y < 8; ++y) bv[y] = Bs[cc][tj + y];
        const double mc = s_metric[cc];
#pragma unroll
        for (int y = 0; y < 8; ++y)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const double diff = __dsub_rn(av[x], bv[y]);  // lib/Space.ml:192-200, as distance_rowwise_kernel
            acc[y][x] = __dadd_rn(acc[y][x], component<KIND>(diff, mc, p));
          }
      }
    }
    // ---- what the summary keeps of the tile
    const uint32_t i0 = ref0 + tile * kFW;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const uint32_t rl = tj + y, j = j0 + rl;
      const FusedThr T = s_thr[rl];
      double sum = 0.0, sq = 0.0;
      uint32_t pk = 0;  // lt | eqlo << 6 | eqhi << 12 | nmed << 18 | inner << 24: at most 4 each here, 32 over the row's eight lanes
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const uint32_t i = i0 + ti + x;
        if (i >= ref1 || j >= q) continue;
        const double d = scale_distance<KIND>(acc[y][x], p);
        const bool medc = d > T.lo && d < T.hi;
        const bool in = d > T.Lin && d < T.Uin;
        pk += (d < T.lo ? 1u : 0u) + (d == T.lo ? 1u << 6 : 0u) + ((d == T.hi && T.hi != T.lo) ? 1u << 12 : 0u) + (medc ? 1u << 18 : 0u) +
              (in ? 1u << 24 : 0u);
        sum = __dadd_rn(sum, d);
        const double dv = __dsub_rn(d, T.mhat);
        sq = __dadd_rn(sq, __dmul_rn(dv, dv));
        if (medc || (!in && d >= T.Llo && d <= T.Uhi)) {
          const uint32_t slot = atomicAdd(&s_ccnt[rl], 1u);
          seg[(uint64_t)j * r1 + ref0 + slot] = d;  // (at most as many as the stripe has columns)
        }
        if (d <= T.cut) {
          const uint32_t at = atomicAdd(&cnt[j].n_nb, 1u);
          if (at < kNbCap) {
            nb_idx[(uint64_t)j * kNbCap + at] = i;
            nb_d[(uint64_t)j * kNbCap + at] = d;
          }
        }
      }
      // the row's eight column groups sit in eight neighbouring lanes: added up in a fixed order, kept by the first
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        sum = __dadd_rn(sum, __shfl_xor(sum, o, 64));
        sq = __dadd_rn(sq, __shfl_xor(sq, o, 64));
        pk += (uint32_t)__shfl_xor((int)pk, o, 64);
      }
      if (cg == 0) {
        s_sum[rl] = __dadd_rn(s_sum[rl], sum);
        s_sq[rl] = __dadd_rn(s_sq[rl], sq);
        s_cnt[rl][0] += (pk & 63u) | (((pk >> 6) & 63u) << 16);
        s_cnt[rl][1] += ((pk >> 12) & 63u) | (((pk >> 18) & 63u) << 16);
        s_cnt[rl][2] += pk >> 24;
      }
    }
  }
  __syncthreads();
  {
    const uint32_t rl = threadIdx.x, j = j0 + rl;
    if (j < q) {
      const uint64_t at = (uint64_t)j * n_stripes + stripe;
      rec[at] = StripeRec{s_cnt[rl][0], s_cnt[rl][1], s_cnt[rl][2], s_ccnt[rl]};
      part[at * 2 + 0] = s_sum[rl];
      part[at * 2 + 1] = s_sq[rl];
    }
  }
}

// ONE pass over the distance rows (slice blockIdx.x of row blockIdx.y) -- the one-kernel path's bookkeeping on rows that
// exist (kpop_summarize_distances on a caller's matrix; the default route for large first operands): everything counted,
// the candidates of the median AND of the MAD's bands collected, so that no second pass for |d - median| is needed.
// (The median's bracket is compared as keys -- a caller's matrix may hold -0.0, which differs from +0.0 as a key --, the
// bands as doubles, here and in the finish kernel alike.)
__global__ __launch_bounds__(256) void summary1_pass_kernel(const double *__restrict__ rows, uint32_t r1, const RowInfo *__restrict__ info,
                                                            const FusedThr *__restrict__ thr, RowCounts *__restrict__ cnt,
                                                            double *__restrict__ cand, uint32_t *__restrict__ cand_i, uint32_t *__restrict__ nb_idx,
                                                            double *__restrict__ nb_d, double *__restrict__ part, uint32_t n_slices, uint32_t kCandCap) {
  __shared__ uint32_t s_c[8];
  __shared__ double s_p[2][4];
  constexpr uint32_t kStage = 4096;
  // (staged: the candidates' columns, 16 KB; their values are read again on the way out -- out of L2, the block has just had
  // them -- rather than staged beside them: 48 KB of LDS a block left three blocks a CU and cost this kernel a third)
  __shared__ uint32_t s_stage_i[kStage];
  __shared__ uint32_t s_n, s_base;
  const uint32_t j = blockIdx.y, sl = blockIdx.x;
  const double *row = rows + (uint64_t)j * r1;
  const RowInfo I = info[j];
  const FusedThr T = thr[j];
  RowCounts *C = cnt + j;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t i0 = sl * kSlice, i1 = min(r1, i0 + kSlice);
  double *my_cand = cand + (uint64_t)j * kCandCap;
  uint32_t *my_cand_i = cand_i ? cand_i + (uint64_t)j * kCandCap : nullptr;
  uint32_t c_lt = 0, c_eqlo = 0, c_eqhi = 0, c_med = 0, c_in = 0;
  double sum = 0.0, sq = 0.0;
  constexpr int U = 8;  // loads in flight per thread
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  auto flush = [&]() {  // all threads; s_n is stable (between barriers)
    const uint32_t cnt_ = s_n;
    if (threadIdx.x == 0) s_base = atomicAdd(&C->n_cand, cnt_);
    __syncthreads();
    const uint32_t b0 = s_base;
    for (uint32_t q0 = threadIdx.x; q0 < cnt_; q0 += 256 * 8) {  // (eight of the gather's loads in flight a thread)
      uint32_t ii[8];
      double vv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        ii[u] = q0 + 256u * u < cnt_ ? s_stage_i[q0 + 256u * u] : i0;  // (past the end: a column of the slice, not stored)
        vv[u] = row[ii[u]];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t q = q0 + 256u * u;
        if (q < cnt_ && b0 + q < kCandCap) {
          my_cand[b0 + q] = vv[u];
          if (my_cand_i) my_cand_i[b0 + q] = ii[u];  // (kept for the matrix-core path's refinement, distance_mfma.hip)
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
  };
  for (uint32_t base = i0; base < i1; base += 256 * U) {
    if (s_n > kStage - 256 * U) flush();  // (block-uniform: read between barriers) room for a whole iteration of candidates
    double dv8[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = base + u * 256 + threadIdx.x;
      dv8[u] = i < i1 ? row[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = base + u * 256 + threadIdx.x;
      const bool ok = i < i1;
      const double d = dv8[u];
      const uint64_t k = f64_key(d);
      const bool medc = k > I.klo && k < I.khi;
      const bool in = d > T.Lin && d < T.Uin;
      if (ok) {
        c_lt += k < I.klo;
        c_eqlo += k == I.klo;
        c_eqhi += (k == I.khi) && I.khi != I.klo;
        c_med += medc;
        c_in += in;
        sum = __dadd_rn(sum, d);
        const double dv = __dsub_rn(d, I.m_hat);
        sq = __dadd_rn(sq, __dmul_rn(dv, dv));
      }
      const bool is_cand = ok && (medc || (!in && d >= T.Llo && d <= T.Uhi));
      if (__ballot(is_cand)) {
        const uint32_t at = wave_append(is_cand, &s_n, lane);
        if (is_cand) s_stage_i[at] = i;
      }
      const bool is_nb = ok && k <= I.kcut;
      if (__ballot(is_nb)) {
        const uint32_t at = wave_append(is_nb, &C->n_nb, lane);
        if (is_nb && at < kNbCap) {
          nb_idx[(uint64_t)j * kNbCap + at] = i;
          nb_d[(uint64_t)j * kNbCap + at] = d;
        }
      }
    }
    __syncthreads();  // s_n is read at the top of the next iteration
  }
  if (s_n) flush();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    c_lt += __shfl_xor((int)c_lt, o, 64);
    c_eqlo += __shfl_xor((int)c_eqlo, o, 64);
    c_eqhi += __shfl_xor((int)c_eqhi, o, 64);
    c_med += __shfl_xor((int)c_med, o, 64);
    c_in += __shfl_xor((int)c_in, o, 64);
    sum = __dadd_rn(sum, __shfl_xor(sum, o, 64));
    sq = __dadd_rn(sq, __shfl_xor(sq, o, 64));
  }
  if (threadIdx.x < 8) s_c[threadIdx.x] = 0;
  __syncthreads();
  if (lane == 0) {
    if (c_lt) atomicAdd(&s_c[0], c_lt);
    if (c_eqlo) atomicAdd(&s_c[1], c_eqlo);
    if (c_eqhi) atomicAdd(&s_c[2], c_eqhi);
    if (c_med) atomicAdd(&s_c[3], c_med);
    if (c_in) atomicAdd(&s_c[4], c_in);
    s_p[0][wv] = sum;
    s_p[1][wv] = sq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_c[0]) atomicAdd(&C->lt_lo, s_c[0]);
    if (s_c[1]) atomicAdd(&C->eq_lo, s_c[1]);
    if (s_c[2]) atomicAdd(&C->eq_hi, s_c[2]);
    if (s_c[3]) atomicAdd(&C->m_lt, s_c[3]);    // (the one-pass path keeps n_med and n_inner where the two-pass path keeps pass 2's counts)
    if (s_c[4]) atomicAdd(&C->m_eqlo, s_c[4]);
    part[((uint64_t)j * n_slices + sl) * 2 + 0] = __dadd_rn(__dadd_rn(__dadd_rn(s_p[0][0], s_p[0][1]), s_p[0][2]), s_p[0][3]);
    part[((uint64_t)j * n_slices + sl) * 2 + 1] = __dadd_rn(__dadd_rn(__dadd_rn(s_p[1][0], s_p[1][1]), s_p[1][2]), s_p[1][3]);
  }
}

// The same pass for rows the library made itself (the matrix-core path's approximate rows: no negative zero, no key needed) -- every
// distance through seven comparisons against the row's thresholds as DOUBLES (the counts of d < lo, d <= lo, d < hi, d <= hi and of the
// inner region: what lies below, at and inside the median's bracket are differences of these), and the candidates of a whole turn
// (eight elements a lane) appended to the block's stage with ONE LDS atomic a wavefront: twelve elements in a hundred are candidates,
// so every wavefront appends at every element, and an atomic round trip each was most of what the pass above spends (2.7 ms per
// 1,024 x 1M at 3.1 TB/s; this one is bound by the read).  Same counts, same candidate and neighbour sets as the pass above.
__global__ __launch_bounds__(256) void summary1_pass_plain_kernel(const double *__restrict__ rows, uint32_t r1, const FusedThr *__restrict__ thr,
                                                                  RowCounts *__restrict__ cnt, double *__restrict__ cand, uint32_t *__restrict__ cand_i,
                                                                  uint32_t *__restrict__ nb_idx, double *__restrict__ nb_d, double *__restrict__ part,
                                                                  uint32_t n_slices, uint32_t kCandCap) {
  __shared__ uint32_t s_c[8];
  __shared__ double s_p[2][4];
  constexpr uint32_t kStage = 4096;
  __shared__ uint32_t s_stage_i[kStage];
  __shared__ uint32_t s_n, s_base;
  const uint32_t j = blockIdx.y, sl = blockIdx.x;
  const double *row = rows + (uint64_t)j * r1;
  const FusedThr T = thr[j];
  RowCounts *C = cnt + j;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t i0 = sl * kSlice, i1 = min(r1, i0 + kSlice);
  double *my_cand = cand + (uint64_t)j * kCandCap;
  uint32_t *my_cand_i = cand_i ? cand_i + (uint64_t)j * kCandCap : nullptr;
  uint32_t c_lt = 0, c_le = 0, c_lth = 0, c_leh = 0, c_in = 0;
  double sum = 0.0, sq = 0.0;
  constexpr int U = 8;  // loads in flight per thread
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  auto flush = [&]() {  // all threads; s_n is stable (between barriers)
    const uint32_t cnt_ = s_n;
    if (threadIdx.x == 0) s_base = atomicAdd(&C->n_cand, cnt_);
    __syncthreads();
    const uint32_t b0 = s_base;
    for (uint32_t q0 = threadIdx.x; q0 < cnt_; q0 += 256 * 8) {  // (eight of the gather's loads in flight a thread)
      uint32_t ii[8];
      double vv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        ii[u] = q0 + 256u * u < cnt_ ? s_stage_i[q0 + 256u * u] : i0;  // (past the end: a column of the slice, not stored)
        vv[u] = row[ii[u]];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t q = q0 + 256u * u;
        if (q < cnt_ && b0 + q < kCandCap) {
          my_cand[b0 + q] = vv[u];
          if (my_cand_i) my_cand_i[b0 + q] = ii[u];
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
  };
  const double nan = __longlong_as_double(0x7FF8000000000000ll);
  for (uint32_t base = i0; base < i1; base += 256 * U) {
    if (s_n > kStage - 256 * U) flush();  // (block-uniform: read between barriers) room for a whole turn of candidates
    const bool whole = base + 256 * U <= i1;  // (uniform)
    double dv8[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = base + u * 256 + threadIdx.x;
      dv8[u] = (whole || i < i1) ? row[i] : nan;  // (a NaN is below, at and inside nothing)
    }
    uint64_t cmask[U];
    uint32_t n_app = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double d = dv8[u];
      const bool lt = d < T.lo, le = d <= T.lo, lth = d < T.hi, leh = d <= T.hi;
      const bool in = d > T.Lin && d < T.Uin;
      c_lt += lt ? 1u : 0u;
      c_le += le ? 1u : 0u;
      c_lth += lth ? 1u : 0u;
      c_leh += leh ? 1u : 0u;
      c_in += in ? 1u : 0u;
      const bool valid = whole || d == d;
      sum = __dadd_rn(sum, valid ? d : 0.0);
      const double dvv = valid ? __dsub_rn(d, T.mhat) : 0.0;
      sq = __dadd_rn(sq, __dmul_rn(dvv, dvv));
      cmask[u] = __ballot((!le && lth) || (!in && d >= T.Llo && d <= T.Uhi));
      n_app += (uint32_t)__popcll(cmask[u]);
      if (__ballot(d <= T.cut)) {  // (one element in a thousand)
        const bool is_nb = d <= T.cut;
        const uint32_t at = wave_append(is_nb, &C->n_nb, lane);
        if (is_nb && at < kNbCap) {
          nb_idx[(uint64_t)j * kNbCap + at] = base + u * 256 + threadIdx.x;
          nb_d[(uint64_t)j * kNbCap + at] = d;
        }
      }
    }
    if (n_app) {  // (wavefront-uniform) the turn's candidates: one atomic, then every lane's columns at its places
      uint32_t at0 = 0;
      if (lane == 0) at0 = atomicAdd(&s_n, n_app);
      at0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)at0);
      const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if ((cmask[u] >> lane) & 1ull) s_stage_i[at0 + (uint32_t)__popcll(cmask[u] & below)] = base + u * 256 + threadIdx.x;
        at0 += (uint32_t)__popcll(cmask[u]);
      }
    }
    __syncthreads();  // s_n is read at the top of the next turn
  }
  if (s_n) flush();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    c_lt += __shfl_xor((int)c_lt, o, 64);
    c_le += __shfl_xor((int)c_le, o, 64);
    c_lth += __shfl_xor((int)c_lth, o, 64);
    c_leh += __shfl_xor((int)c_leh, o, 64);
    c_in += __shfl_xor((int)c_in, o, 64);
    sum = __dadd_rn(sum, __shfl_xor(sum, o, 64));
    sq = __dadd_rn(sq, __shfl_xor(sq, o, 64));
  }
  if (threadIdx.x < 8) s_c[threadIdx.x] = 0;
  __syncthreads();
  if (lane == 0) {
    const uint32_t eqlo = c_le - c_lt, nmed = T.lo < T.hi ? c_lth - c_le : 0u, eqhi = T.hi != T.lo ? c_leh - c_lth : 0u;
    if (c_lt) atomicAdd(&s_c[0], c_lt);
    if (eqlo) atomicAdd(&s_c[1], eqlo);
    if (eqhi) atomicAdd(&s_c[2], eqhi);
    if (nmed) atomicAdd(&s_c[3], nmed);
    if (c_in) atomicAdd(&s_c[4], c_in);
    s_p[0][wv] = sum;
    s_p[1][wv] = sq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_c[0]) atomicAdd(&C->lt_lo, s_c[0]);
    if (s_c[1]) atomicAdd(&C->eq_lo, s_c[1]);
    if (s_c[2]) atomicAdd(&C->eq_hi, s_c[2]);
    if (s_c[3]) atomicAdd(&C->m_lt, s_c[3]);
    if (s_c[4]) atomicAdd(&C->m_eqlo, s_c[4]);
    part[((uint64_t)j * n_slices + sl) * 2 + 0] = __dadd_rn(__dadd_rn(__dadd_rn(s_p[0][0], s_p[0][1]), s_p[0][2]), s_p[0][3]);
    part[((uint64_t)j * n_slices + sl) * 2 + 1] = __dadd_rn(__dadd_rn(__dadd_rn(s_p[1][0], s_p[1][1]), s_p[1][2]), s_p[1][3]);
  }
}

struct BandRow {  // the candidates of the MAD among a row's compacted list: NaN for the others (its key is beyond every range)
  const double *p;
  double Llo, Lin, Uin, Uhi;
  __device__ __forceinline__ double operator[](uint32_t i) const {
    const double d = p[i];
    const bool in = d > Lin && d < Uin;
    return (!in && d >= Llo && d <= Uhi) ? d : __longlong_as_double(0x7FF8000000000000ll);
  }
};

// SEG: the candidates lie in per-stripe segments of `seg` with their counts in `rec` (the one-kernel path); otherwise in the
// row's contiguous list `ccand` with the counts in `cnt` (the one-pass path over distance rows; n_stripes = its slices)
template <bool SEG>
__global__ __launch_bounds__(kLT) void fused_finish_kernel(const double *__restrict__ seg, uint32_t r1, uint32_t row0, uint32_t req_len,
                                                           uint32_t max_neighbours, const RowInfo *__restrict__ info,
                                                           const FusedThr *__restrict__ thr, RowCounts *__restrict__ cnt,
                                                           const StripeRec *__restrict__ rec, const double *__restrict__ part,
                                                           uint32_t n_stripes, uint32_t *__restrict__ pre, double *__restrict__ ccand, uint32_t cap,
                                                           const uint32_t *__restrict__ nb_idx, const double *__restrict__ nb_d,
                                                           uint32_t *__restrict__ n_failed, double *__restrict__ out_stats,
                                                           uint32_t *__restrict__ out_n, uint32_t *__restrict__ out_idx,
                                                           double *__restrict__ out_dist, double *__restrict__ out_z,
                                                           const uint32_t *__restrict__ seg_i = nullptr, uint32_t *__restrict__ ccand_i = nullptr,
                                                           uint32_t seg_stride = kStripe, uint64_t seg_ld = 0) {
  // (seg_i / ccand_i: the candidates' columns beside their values, compacted the same way -- the matrix-core path's refinement reads them;
  //  seg_stride: the elements between the starts of two records' segments -- kStripe, or the matrix-core kernel's sub-stripes of a quarter;
  //  seg_ld: the elements between two rows' segments -- r1 (0), or the matrix-core kernel's whole stripes)
  __shared__ uint32_t s_hist[kSel * kBins];
  __shared__ uint64_t s_cand[kSel * kCand];
  __shared__ uint32_t s_misc[64];
  __shared__ double s_cd[kNbSort];
  __shared__ uint32_t s_ci[kNbSort];
  __shared__ uint32_t s_take;
  __shared__ uint32_t s_wu[kLT / 64];
  __shared__ uint32_t s_tot[8];
  __shared__ uint64_t s_mm[2 * (kLT / 64)];
  const uint32_t jl = blockIdx.x, j = row0 + jl, n = r1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const RowInfo I = info[jl];
  const FusedThr T = thr[jl];
  const StripeRec *my_rec = rec + (uint64_t)jl * n_stripes;
  uint32_t *my_pre = pre + (uint64_t)jl * n_stripes;
  double *my_c = ccand + (uint64_t)jl * cap;
  if (threadIdx.x < 8) s_tot[threadIdx.x] = 0;
  __syncthreads();
  KPOP_STAMP(0);
  uint32_t lt, eqlo, eqhi, nmed, n_inner, n_c;
  bool ok;
  if constexpr (SEG) {
    // ---- the stripes' counts: totals, and where each stripe's candidates go in the compacted list
    uint32_t running = 0;
    for (uint32_t base = 0; base < n_stripes; base += kLT) {
      const uint32_t st = base + threadIdx.x;
      StripeRec R = {0, 0, 0, 0};
      if (st < n_stripes) R = my_rec[st];
      // exclusive scan of c_cnt over the block
      uint32_t incl = R.c_cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
      }
      __syncthreads();
      if (lane == 63) s_wu[wv] = incl;
      __syncthreads();
      uint32_t before = 0, total = 0;
      for (int w = 0; w < kLT / 64; ++w) {
        if (w < wv) before += s_wu[w];
        total += s_wu[w];
      }
      if (st < n_stripes) {
        my_pre[st] = running + before + incl - R.c_cnt;
        atomicAdd(&s_tot[0], R.lt_eqlo & 0xFFFFu);
        atomicAdd(&s_tot[1], R.lt_eqlo >> 16);
        atomicAdd(&s_tot[2], R.eqhi_nmed & 0xFFFFu);
        atomicAdd(&s_tot[3], R.eqhi_nmed >> 16);
        atomicAdd(&s_tot[4], R.inner);
      }
      running += total;
    }
    __syncthreads();
    lt = s_tot[0]; eqlo = s_tot[1]; eqhi = s_tot[2]; nmed = s_tot[3]; n_inner = s_tot[4]; n_c = running;
    ok = n_c <= cap;
    // ---- the candidates, compacted: a wavefront per record, or -- records of up to 512 elements, ~60 of them filled -- sixteen lanes
    if (ok) {
      const uint32_t gl = seg_stride <= 512u ? 16u : 64u, gpw = 64u / gl;  // lanes a record, records a wavefront at a time
      const uint32_t lg = (uint32_t)lane % gl, g = (uint32_t)lane / gl;
      for (uint32_t st = (uint32_t)wv * gpw + g; st < n_stripes; st += (kLT / 64) * gpw) {
        const uint32_t c = my_rec[st].c_cnt, at = my_pre[st];
        const double *src = seg + (uint64_t)jl * (seg_ld ? seg_ld : (uint64_t)r1) + (uint64_t)st * seg_stride;
        for (uint32_t e0 = 0; e0 < c; e0 += gl * 8) {  // eight loads in flight a lane (a loop of one was latency-bound: 0.24 ms)
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = e0 + u * gl + lg < c ? src[e0 + u * gl + lg] : 0.0;
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (e0 + u * gl + lg < c) my_c[at + e0 + u * gl + lg] = v[u];
          if (seg_i) {
            const uint32_t *srci = seg_i + (uint64_t)jl * (seg_ld ? seg_ld : (uint64_t)r1) + (uint64_t)st * seg_stride;
            uint32_t *my_ci = ccand_i + (uint64_t)jl * cap;
            uint32_t vi[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vi[u] = e0 + u * gl + lg < c ? srci[e0 + u * gl + lg] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (e0 + u * gl + lg < c) my_ci[at + e0 + u * gl + lg] = vi[u];
          }
        }
      }
    }
    if (threadIdx.x == 0) {  // (the totals where the one-pass path keeps them: the matrix-core path's refinement reads them from there)
      RowCounts *C = cnt + jl;
      C->lt_lo = lt; C->eq_lo = eqlo; C->eq_hi = eqhi; C->m_lt = nmed; C->m_eqlo = n_inner; C->n_cand = n_c;
    }
    __syncthreads();
  } else {
    const RowCounts C = cnt[jl];
    lt = C.lt_lo; eqlo = C.eq_lo; eqhi = C.eq_hi; nmed = C.m_lt; n_inner = C.m_eqlo; n_c = C.n_cand;
    ok = n_c <= cap;
  }
  // ---- mean and sd: the stripes' sums in a fixed order -- runs of eight stripes by a thread each, then the runs' sums in
  // order by every thread out of LDS (one chain over 512 global loads was a quarter of this kernel)
  KPOP_STAMP(1);
  double sum = 0.0, sqh = 0.0;
  {
    double *s_ps = s_cd;  // (the neighbours' sort buffer, not yet in use: 2 x 1,024 doubles)
    const uint32_t n_runs = (n_stripes + 7) / 8;  // <= 1,024
    if (threadIdx.x < n_runs) {
      double a0 = 0.0, a1 = 0.0;
      for (uint32_t st = threadIdx.x * 8; st < min(n_stripes, threadIdx.x * 8 + 8); ++st) {
        a0 = __dadd_rn(a0, part[((uint64_t)jl * n_stripes + st) * 2 + 0]);
        a1 = __dadd_rn(a1, part[((uint64_t)jl * n_stripes + st) * 2 + 1]);
      }
      s_ps[2 * threadIdx.x] = a0;
      s_ps[2 * threadIdx.x + 1] = a1;
    }
    __syncthreads();
    for (uint32_t u = 0; u < n_runs; ++u) {
      sum = __dadd_rn(sum, s_ps[2 * u]);
      sqh = __dadd_rn(sqh, s_ps[2 * u + 1]);
    }
    __syncthreads();
  }
  const double mean = sum / (double)n;
  const double dm = __dsub_rn(mean, I.m_hat);
  const double ss = fmax(0.0, __dsub_rn(sqh, __dmul_rn((double)n, __dmul_rn(dm, dm))));
  const double sd = n > 1 ? sqrt(ss / ((double)n - 1.0)) : 0.0;
  // ---- the median: rank n / 2 among [lt below the bracket | eqlo at its lower end | nmed inside | eqhi at its upper end]
  KPOP_STAMP(2);
  double median = 0.0;
  const uint32_t r = n / 2;
  if (ok) {
    if (r < lt) { ok = false; KPOP_WHY(1); }
    else if (r < lt + eqlo) median = key_f64(I.klo);
    else if (r < lt + eqlo + nmed) {
      Sel sm[1] = {Sel{r - lt - eqlo, I.klo + 1, I.khi - 1, 0, 0, 0, 0, 0}};  // keys strictly inside the bracket
      block_select_ranks<0>(PlainRow{my_c}, n_c, 0.0, sm, 1, s_hist, s_cand, s_misc);
      median = key_f64(sm[0].value);
    } else if (r < lt + eqlo + nmed + eqhi) median = key_f64(I.khi);
    else { ok = false; KPOP_WHY(2); }
  }
  KPOP_STAMP(3);
  // ---- the neighbours
  uint32_t eff = n;
  if (ok)
    ok = neighbours_from_list(n, req_len, max_neighbours, cnt[jl].n_nb, nb_idx + (uint64_t)jl * kNbCap, nb_d + (uint64_t)jl * kNbCap, I.kcut, mean, sd,
                              j, out_idx, out_dist, out_z, s_hist, s_cand, s_misc, s_cd, s_ci, &s_take, &eff);
  KPOP_STAMP(4);
  if (!ok) KPOP_WHY(3);
  // ---- the MAD
  double mad = 0.0;
  if (ok) {
    const BandRow br{my_c, T.Llo, T.Lin, T.Uin, T.Uhi};
    uint64_t kmin = ~0ull, kmax = 0;
    uint32_t n_band = 0;
    for (uint32_t i0 = threadIdx.x; i0 < n_c; i0 += kLT * 8) {  // (eight loads in flight a thread)
      double d8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d8[u] = i0 + u * kLT < n_c ? br[i0 + u * kLT] : __longlong_as_double(0x7FF8000000000000ll);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double d = d8[u];
        if (d == d) {
          const uint64_t k = f64_key(fabs(__dsub_rn(d, median)));
          kmin = kmin < k ? kmin : k;
          kmax = kmax > k ? kmax : k;
          ++n_band;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t omin = (uint64_t)__shfl_xor((unsigned long long)kmin, o, 64), omax = (uint64_t)__shfl_xor((unsigned long long)kmax, o, 64);
      kmin = kmin < omin ? kmin : omin;
      kmax = kmax > omax ? kmax : omax;
      n_band += (uint32_t)__shfl_xor((int)n_band, o, 64);
    }
    __syncthreads();
    if (lane == 0) {
      s_mm[wv] = kmin;
      s_mm[kLT / 64 + wv] = kmax;
      s_wu[wv] = n_band;
    }
    __syncthreads();
    n_band = 0;
    for (int w = 0; w < kLT / 64; ++w) {
      kmin = kmin < s_mm[w] ? kmin : s_mm[w];
      kmax = kmax > s_mm[kLT / 64 + w] ? kmax : s_mm[kLT / 64 + w];
      n_band += s_wu[w];
    }
    KPOP_STAMP(5);
    if (r < n_inner || r - n_inner >= n_band) { ok = false; KPOP_WHY(4); }
    else {
      Sel sa[1] = {Sel{r - n_inner, kmin, kmax, 0, 0, 0, 0, 0}};
      block_select_ranks<1, false, BandRow>(br, n_c, median, sa, 1, s_hist, s_cand, s_misc);
      mad = key_f64(sa[0].value);
      // the certificate: everything inside the inner region is at most X_in from the median, everything beyond the bands at
      // least X_out (the same subtractions, monotone): with X_in <= mad <= X_out the ranks are what was assumed
      const uint32_t n_outer = n - n_inner - n_band;
      if (n_inner) {
        const double x_in = fmax(__dsub_rn(T.Uin, median), __dsub_rn(median, T.Lin));
        if (!(mad >= x_in)) { ok = false; KPOP_WHY(5); }
      }
      if (n_outer) {
        const double x_out = fmin(__dsub_rn(median, T.Llo), __dsub_rn(T.Uhi, median));
        if (!(mad <= x_out)) { ok = false; KPOP_WHY(6); }
      }
    }
  }
  KPOP_STAMP(6);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
#ifdef KPOP_SUMMARY_STAMPS
    g_sum_stamps[8] = n_c;
    g_sum_stamps[9] = cnt[jl].n_nb;
    g_sum_stamps[10] = nmed;
#endif
  }
  if (threadIdx.x == 0) {
    if (!ok) {
      cnt[jl].fail = 1;
      atomicAdd(n_failed, 1u);
    } else {
      out_stats[(uint64_t)j * 4 + 0] = mean;
      out_stats[(uint64_t)j * 4 + 1] = sd;
      out_stats[(uint64_t)j * 4 + 2] = median;
      out_stats[(uint64_t)j * 4 + 3] = mad;
      out_n[j] = eff;
    }
  }
}

// ---- host side ------------------------------------------------------------
bool summary_fused_applies(uint32_t r1, uint32_t keep_at_most) {
  const uint32_t req_len = keep_at_most ? keep_at_most : r1;
  return ctx().tune_summary2 == 2 && req_len <= kLargeMaxNb && r1 >= 4 * kSlice && (r1 + kStripe - 1) / kStripe <= kMaxStripes;
}
uint32_t summary_fused_sample_rows(uint32_t r1) { return std::min<uint32_t>(kSample, r1); }

struct FusedScratch {
  RowInfo *info;
  RowCounts *cnt;
  FusedThr *thr;
  StripeRec *rec;
  double *part;
  uint32_t *pre;
  double *nb_d;
  uint32_t *nb_idx;
  double *ccand;
  uint32_t *ccand_i;
  uint32_t *n_failed;
};
static uint64_t carve_fused(void *scratch, uint32_t n_rows, uint32_t r1, FusedScratch *F, uint32_t stripe = kStripe) {
  // records a row: one a stripe, or -- the matrix-core kernel, stripe = kStripe / 4 -- FOUR to every stripe of kStripe, the last one too
  const uint64_t n_stripes = stripe == kStripe ? (r1 + kStripe - 1) / kStripe : (uint64_t)(kStripe / stripe) * ((r1 + kStripe - 1) / kStripe);
  char *p0 = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(scratch) + 255) & ~(uintptr_t)255), *p = p0;
  auto take = [&](uint64_t bytes) {
    char *at = p;
    p += (bytes + 255) & ~255ull;
    return at;
  };
  F->n_failed = reinterpret_cast<uint32_t *>(take(256));
  F->info = reinterpret_cast<RowInfo *>(take((uint64_t)n_rows * sizeof(RowInfo)));
  F->cnt = reinterpret_cast<RowCounts *>(take((uint64_t)n_rows * sizeof(RowCounts)));
  F->thr = reinterpret_cast<FusedThr *>(take((uint64_t)n_rows * sizeof(FusedThr)));
  F->rec = reinterpret_cast<StripeRec *>(take((uint64_t)n_rows * n_stripes * sizeof(StripeRec)));
  F->part = reinterpret_cast<double *>(take((uint64_t)n_rows * n_stripes * 16));
  F->pre = reinterpret_cast<uint32_t *>(take((uint64_t)n_rows * n_stripes * 4));
  F->nb_d = reinterpret_cast<double *>(take((uint64_t)n_rows * kNbCap * 8));
  F->nb_idx = reinterpret_cast<uint32_t *>(take((uint64_t)n_rows * kNbCap * 4));
  F->ccand = reinterpret_cast<double *>(take((uint64_t)n_rows * fused_cand_cap(r1) * 8));
  F->ccand_i = reinterpret_cast<uint32_t *>(take((uint64_t)n_rows * fused_cand_cap(r1) * 4));
  return (uint64_t)(p - p0) + 256;
}
uint64_t summary_fused_scratch_bytes(uint32_t n_rows, uint32_t r1) {
  FusedScratch F;
  return carve_fused(nullptr, n_rows, r1, &F);
}
uint64_t summary_select_scratch_bytes(uint32_t n_rows, uint32_t r1) {
  FusedScratch F;
  return carve_fused(nullptr, n_rows, r1, &F, kStripe / 4);
}

int launch_sample_gather(const double *a, uint32_t r1, uint32_t n_dims, uint32_t s, double *out, hipStream_t st) {
  sample_gather_kernel<<<dim3((s + 3) / 4), dim3(256), 0, st>>>(a, r1, n_dims, s, out);
  KPOP_LAUNCH_CHECK();
  return 0;
}

// a, b: the prepared operands (b: the chunk's n_rows query rows); srow: [n_rows][s] distances to the sample; seg: [n_rows][r1]
// (segments now, distance rows of the fallback later).  *gate: the device word that counts the rows left to the fallback.
int launch_summary_fused(int kind, const double *a, uint32_t r1, const double *b, uint32_t n_rows, uint32_t n_dims, const double *metric, double p,
                         const double *srow, uint32_t s, uint32_t row0, uint32_t keep_at_most, uint32_t max_neighbours, double *out_stats,
                         uint32_t *out_n, uint32_t *out_idx, double *out_dist, double *out_z, double *seg, void *scratch, hipStream_t st,
                         const uint32_t **gate) {
  const uint32_t req_len = keep_at_most ? keep_at_most : r1;
  const uint32_t n_stripes = (r1 + kStripe - 1) / kStripe;
  FusedScratch F;
  carve_fused(scratch, n_rows, r1, &F);
  KPOP_HIP(hipMemsetAsync(F.n_failed, 0, 256, st));
  fused_sample_kernel<false><<<dim3(n_rows), dim3(kLT), 0, st>>>(srow, s, r1, req_len, F.info, F.cnt, F.thr);
  KPOP_LAUNCH_CHECK();
  const dim3 grid(n_stripes, (n_rows + kFQ - 1) / kFQ);
#define KPOP_FUSED(K) \
  summary_fused_pass_kernel<K><<<grid, dim3(256), 0, st>>>(a, r1, b, n_rows, n_dims, metric, p, F.thr, seg, F.rec, F.part, F.cnt, F.nb_idx, F.nb_d, n_stripes)
  if (kind == KPOP_EUCLIDEAN) KPOP_FUSED(KPOP_EUCLIDEAN);
  else if (kind == KPOP_COSINE) KPOP_FUSED(KPOP_COSINE);
  else KPOP_FUSED(KPOP_MINKOWSKI);
#undef KPOP_FUSED
  KPOP_LAUNCH_CHECK();
  fused_finish_kernel<true><<<dim3(n_rows), dim3(kLT), 0, st>>>(seg, r1, row0, req_len, max_neighbours, F.info, F.thr, F.cnt, F.rec, F.part, n_stripes, F.pre,
                                                          F.ccand, fused_cand_cap(r1), F.nb_idx, F.nb_d, F.n_failed, out_stats, out_n, out_idx, out_dist,
                                                          out_z);
  KPOP_LAUNCH_CHECK();
  *gate = F.n_failed;
  return 0;
}

// The same with the pass on the f64 MATRIX cores (distance_mfma.hip: summary_select_mfma_kernel -- the approximate distances classified in the
// accumulators' registers, no distance row written): thresholds from the distances to the sample (`srow`, [n_rows][s], approximate as
// well), the pass, the finish over the stripes' segments (values in `seg`, columns in `seg_i`, both [n_rows][r1]).  What it reports is
// approximate: `lists` is what the exact refinement (summary_refine_kernel) reads.
int launch_select_mfma(int kind, const double *a, uint32_t r1, uint32_t q, uint32_t n_dims, const void *mscratch, uint32_t q_room, const FusedThr *thr, double *seg,
                       uint32_t *seg_i, StripeRec *rec, double *part, RowCounts *cnt, uint32_t *nb_idx, double *nb_d, uint32_t n_stripes, hipStream_t st);
int launch_summary_fused_mfma(int kind, const double *a, uint32_t r1, uint32_t n_rows, uint32_t n_dims, const double *srow, uint32_t s, uint32_t row0,
                              uint32_t keep_at_most, uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx, double *out_dist,
                              double *out_z, double *seg, uint32_t *seg_i, void *scratch, const void *mscratch, uint32_t q_room, hipStream_t st,
                              SummaryLists *lists) {
  const uint32_t req_len = keep_at_most ? keep_at_most : r1;
  const uint32_t n_stripes = (r1 + kStripe - 1) / kStripe;  // blocks of the matrix-core kernel along the reference rows; it keeps FOUR records a stripe
  FusedScratch F;
  carve_fused(scratch, n_rows, r1, &F, kStripe / 4);
  KPOP_HIP(hipMemsetAsync(F.n_failed, 0, 256, st));
  // (a last stripe that is not whole leaves some of its four records unwritten -- a lane whose 512 slots lie past the end still writes
  // its record, of zeros: every record of a row is written, see the kernel)
  fused_sample_kernel<false><<<dim3(n_rows), dim3(kLT), 0, st>>>(srow, s, r1, req_len, F.info, F.cnt, F.thr);
  KPOP_LAUNCH_CHECK();
  KPOP_TRY(launch_select_mfma(kind, a, r1, n_rows, n_dims, mscratch, q_room, F.thr, seg, seg_i, F.rec, F.part, F.cnt, F.nb_idx, F.nb_d, n_stripes, st));
  fused_finish_kernel<true><<<dim3(n_rows), dim3(kLT), 0, st>>>(seg, r1, row0, req_len, max_neighbours, F.info, F.thr, F.cnt, F.rec, F.part, 4 * n_stripes, F.pre,
                                                          F.ccand, fused_cand_cap(r1), F.nb_idx, F.nb_d, F.n_failed, out_stats, out_n, out_idx, out_dist,
                                                          out_z, seg_i, F.ccand_i, kStripe / 4, (uint64_t)n_stripes * kStripe);
  KPOP_LAUNCH_CHECK();
  *lists = SummaryLists{F.info, F.thr, F.cnt, F.ccand, F.ccand_i, fused_cand_cap(r1), F.nb_idx, F.nb_d};
  return 0;
}
bool summary_select_mfma_applies(uint32_t r1, uint32_t keep_at_most) {
  const uint32_t req_len = keep_at_most ? keep_at_most : r1;
  return req_len <= kLargeMaxNb && r1 >= 4 * kSlice && 4 * ((r1 + kStripe - 1) / kStripe) <= kMaxStripes;  // (four records a stripe)
}

// the rows the fused path flagged, from distance rows computed meanwhile (gated the same way: nothing runs when none failed)
int launch_summary_failed_rows(const double *rows, uint32_t n_rows, uint32_t r1, uint32_t row0, uint32_t keep_at_most, uint32_t max_neighbours,
                               double *out_stats, uint32_t *out_n, uint32_t *out_idx, double *out_dist, double *out_z, void *scratch,
                               hipStream_t st) {
  FusedScratch F;
  carve_fused(scratch, n_rows, r1, &F);
  summary_large_kernel<<<dim3(n_rows), dim3(kLT), 0, st>>>(rows, r1, row0, keep_at_most ? keep_at_most : r1, max_neighbours, out_stats, out_n, out_idx,
                                                           out_dist, out_z, F.cnt);
  KPOP_LAUNCH_CHECK();
  return 0;
}

// the same over rows another path flagged (distance_mfma.hip: `flags` = its RowCounts, one a row of the chunk)
int launch_summary_flagged_rows(const double *rows, uint32_t n_rows, uint32_t r1, uint32_t row0, uint32_t keep_at_most, uint32_t max_neighbours,
                                double *out_stats, uint32_t *out_n, uint32_t *out_idx, double *out_dist, double *out_z, const void *flags, hipStream_t st) {
  summary_large_kernel<<<dim3(n_rows), dim3(kLT), 0, st>>>(rows, r1, row0, keep_at_most ? keep_at_most : r1, max_neighbours, out_stats, out_n, out_idx,
                                                           out_dist, out_z, reinterpret_cast<const RowCounts *>(flags));
  KPOP_LAUNCH_CHECK();
  return 0;
}
static_assert(sizeof(RowCounts) == 48 && offsetof(RowCounts, fail) == 36, "distance_mfma.hip writes RowCounts::fail by its word index");

static inline uint32_t rows_cand_cap(uint32_t r1) { return std::max(cand_cap_for(r1), fused_cand_cap(r1)); }

uint64_t summary_large_scratch_bytes(uint32_t n_rows, uint32_t r1) {
  const uint64_t n_slices = (r1 + kSlice - 1) / kSlice;
  return (uint64_t)n_rows * (sizeof(RowInfo) + sizeof(RowCounts) + sizeof(FusedThr) + (uint64_t)rows_cand_cap(r1) * 12 + (uint64_t)kNbCap * 12 + n_slices * 16) +
         8192;
}

// scratch = nullptr (or tune "summary2" 0): the one-block-per-row kernel alone.  "summary2" 1 (default): ONE pass over the
// rows (brackets and bands from a sample, certificate for the MAD); 3: two passes (the first version of round 3).
int launch_summary_large(const double *rows, uint32_t n_rows, uint32_t r1, uint32_t row0, uint32_t keep_at_most,
                         uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx,
                         double *out_dist, double *out_z, hipStream_t st, void *scratch, SummaryLists *lists, bool plain_rows, const double *srow, uint32_t srow_n) {
  // (srow: [n_rows][srow_n] distances of the query rows to a SAMPLE OF THE REFERENCE ROWS at even spacing -- the brackets and bands come from
  // them; nullptr: from runs of 1,024 consecutive elements of the distance rows themselves.  A database laid out lineage by lineage makes
  // neighbouring elements of a distance row near-copies of each other: 64 runs then speak for 640 of 10,000 clusters, the brackets
  // miss (126 + 176 of 512 rows on clusters of 100) and the rows go through the ten-pass kernel -- 8.1 ms where random rows take 2.3)
  if (lists) *lists = SummaryLists{};
  const uint32_t req_len = keep_at_most ? keep_at_most : r1;
  const bool by_brackets = scratch && ctx().tune_summary2 && req_len <= kLargeMaxNb && r1 >= 2 * kSlice;
  if (!by_brackets) {
    summary_large_kernel<<<dim3(n_rows), dim3(kLT), 0, st>>>(rows, r1, row0, req_len, max_neighbours, out_stats, out_n, out_idx, out_dist,
                                                             out_z, nullptr);
    KPOP_LAUNCH_CHECK();
    return 0;
  }
  const uint32_t n_slices = (r1 + kSlice - 1) / kSlice;
  char *p = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(scratch) + 255) & ~(uintptr_t)255);
  uint32_t *n_failed = reinterpret_cast<uint32_t *>(p);
  p += 256;
  RowInfo *info = reinterpret_cast<RowInfo *>(p);
  p += ((uint64_t)n_rows * sizeof(RowInfo) + 255) & ~255ull;
  RowCounts *cnt = reinterpret_cast<RowCounts *>(p);
  p += ((uint64_t)n_rows * sizeof(RowCounts) + 255) & ~255ull;
  FusedThr *thr = reinterpret_cast<FusedThr *>(p);
  p += ((uint64_t)n_rows * sizeof(FusedThr) + 255) & ~255ull;
  const uint32_t cap = rows_cand_cap(r1);
  double *cand = reinterpret_cast<double *>(p);
  p += (uint64_t)n_rows * cap * 8;
  double *nb_d = reinterpret_cast<double *>(p);
  p += (uint64_t)n_rows * kNbCap * 8;
  uint32_t *nb_idx = reinterpret_cast<uint32_t *>(p);
  p += (uint64_t)n_rows * kNbCap * 4;
  double *part = reinterpret_cast<double *>(p);
  p += ((uint64_t)n_rows * n_slices * 16 + 255) & ~255ull;
  uint32_t *cand_i = lists ? reinterpret_cast<uint32_t *>(p) : nullptr;
  if (ctx().tune_summary2 != 3) {
    if (lists) *lists = SummaryLists{info, thr, cnt, cand, cand_i, cap, nb_idx, nb_d, n_failed};
    KPOP_HIP(hipMemsetAsync(n_failed, 0, 256, st));
    if (srow)
      fused_sample_kernel<false><<<dim3(n_rows), dim3(kLT), 0, st>>>(srow, srow_n, r1, req_len, info, cnt, thr);
    else
      fused_sample_kernel<true><<<dim3(n_rows), dim3(kLT), 0, st>>>(rows, 0, r1, req_len, info, cnt, thr);
    KPOP_LAUNCH_CHECK();
    // (plain_rows: the caller made the rows itself -- distances >= +0.0 -- and they need no keys; kpop_tune("summary_pass", 0): the general pass anyway)
    if (plain_rows && ctx().tune_summary_pass)
      summary1_pass_plain_kernel<<<dim3(n_slices, n_rows), dim3(256), 0, st>>>(rows, r1, thr, cnt, cand, cand_i, nb_idx, nb_d, part, n_slices, cap);
    else
      summary1_pass_kernel<<<dim3(n_slices, n_rows), dim3(256), 0, st>>>(rows, r1, info, thr, cnt, cand, cand_i, nb_idx, nb_d, part, n_slices, cap);
    KPOP_LAUNCH_CHECK();
    fused_finish_kernel<false><<<dim3(n_rows), dim3(kLT), 0, st>>>(rows, r1, row0, req_len, max_neighbours, info, thr, cnt, nullptr, part, n_slices, nullptr, cand,
                                                                   cap, nb_idx, nb_d, n_failed, out_stats, out_n, out_idx, out_dist, out_z);
    KPOP_LAUNCH_CHECK();
  } else {
    summary2_sample_kernel<<<dim3(n_rows), dim3(kLT), 0, st>>>(rows, r1, req_len, info, cnt);
    KPOP_LAUNCH_CHECK();
    summary2_pass_kernel<1><<<dim3(n_slices, n_rows), dim3(256), 0, st>>>(rows, r1, info, cnt, cand, nb_idx, nb_d, part, n_slices, cap);
    KPOP_LAUNCH_CHECK();
    summary2_finish_kernel<1><<<dim3(n_rows), dim3(kLT), 0, st>>>(rows, r1, row0, req_len, max_neighbours, info, cnt, cand, nb_idx, nb_d, part, n_slices,
                                                                  cap, out_stats, out_n, out_idx, out_dist, out_z);
    KPOP_LAUNCH_CHECK();
    summary2_pass_kernel<2><<<dim3(n_slices, n_rows), dim3(256), 0, st>>>(rows, r1, info, cnt, cand, nb_idx, nb_d, part, n_slices, cap);
    KPOP_LAUNCH_CHECK();
    summary2_finish_kernel<2><<<dim3(n_rows), dim3(kLT), 0, st>>>(rows, r1, row0, req_len, max_neighbours, info, cnt, cand, nb_idx, nb_d, part, n_slices,
                                                                  cap, out_stats, out_n, out_idx, out_dist, out_z);
    KPOP_LAUNCH_CHECK();
  }
  summary_large_kernel<<<dim3(n_rows), dim3(kLT), 0, st>>>(rows, r1, row0, req_len, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z, cnt);
  KPOP_LAUNCH_CHECK();
  return 0;
}

}  // namespace kpop

#ifdef KPOP_SUMMARY_STAMPS
extern "C" int kpop_debug_summary_stamps(unsigned long long *out) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kpop::g_sum_stamps), 32 * 8) == hipSuccess ? 0 : -1;
}
#endif

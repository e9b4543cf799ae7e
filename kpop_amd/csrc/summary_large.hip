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

#include "common.h"

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

// TRANSFORM 0: key of row[i]; 1: key of |row[i] - centre|
template <int TRANSFORM>
__device__ __forceinline__ uint64_t elem_key(const double *row, uint32_t i, double centre) {
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
constexpr int kSel = 2;

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
template <int TRANSFORM, bool SUMSQ = false>
__device__ void block_select_ranks(const double *row, uint32_t n, double centre, Sel *sel, int n_sel, uint32_t *s_hist,
                                   uint64_t *s_cand, uint32_t *s_misc, double *sumsq_part = nullptr) {
  for (int round = 0; round < 8; ++round) {
    bool any = false;
    for (int t = 0; t < n_sel; ++t) any = any || !sel[t].done;
    if (!any) return;
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < (uint32_t)n_sel * kBins; q += kLT) s_hist[q] = 0;
    __syncthreads();
    int shift[kSel];
    for (int t = 0; t < n_sel; ++t) shift[t] = range_shift(sel[t].lo, sel[t].hi);
    double sq = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += kLT) {
      const uint64_t k = elem_key<TRANSFORM>(row, i, centre);
      if (SUMSQ && round == 0) {
        const double dv = __dsub_rn(row[i], centre);
        sq = __dadd_rn(sq, __dmul_rn(dv, dv));
      }
      for (int t = 0; t < n_sel; ++t)
        if (!sel[t].done && k >= sel[t].lo && k <= sel[t].hi) atomicAdd(&s_hist[t * kBins + (uint32_t)((k - sel[t].lo) >> shift[t])], 1u);
    }
    if (SUMSQ && round == 0) *sumsq_part = sq;
    __syncthreads();
    // the bin of each rank (one thread per selection walks 2048 counters)
    if (threadIdx.x < (uint32_t)n_sel && !sel[threadIdx.x].done) {
      const int t = threadIdx.x;
      uint32_t cum = sel[t].below, bin = 0;
      for (; bin < kBins; ++bin) {
        const uint32_t c = s_hist[t * kBins + bin];
        if (cum + c > sel[t].rank) break;
        cum += c;
      }
      if (bin == kBins) bin = kBins - 1;  // only reachable with NaNs in the row
      s_misc[t * 4 + 0] = bin;
      s_misc[t * 4 + 1] = cum;
      s_misc[t * 4 + 2] = s_hist[t * kBins + bin];
    }
    __syncthreads();
    bool collect[kSel];
    for (int t = 0; t < n_sel; ++t) {
      collect[t] = false;
      if (sel[t].done) continue;
      const uint32_t bin = s_misc[t * 4 + 0], cum = s_misc[t * 4 + 1], cnt = s_misc[t * 4 + 2];
      const uint64_t blo = sel[t].lo + ((uint64_t)bin << shift[t]);
      const uint64_t bhi = shift[t] ? min(sel[t].hi, blo + ((1ull << shift[t]) - 1)) : blo;
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
      for (uint32_t i = threadIdx.x; i < n; i += kLT) {
        const uint64_t k = elem_key<TRANSFORM>(row, i, centre);
        for (int t = 0; t < n_sel; ++t)
          if (collect[t] && k >= sel[t].lo && k <= sel[t].hi) {
            const uint32_t at = atomicAdd(&s_misc[32 + t], 1u);
            if (at < kCand) s_cand[t * kCand + at] = k;
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

__global__ __launch_bounds__(kLT) void summary_large_kernel(const double *__restrict__ rows, uint32_t r1, uint32_t row0,
                                                            uint32_t req_len, uint32_t max_neighbours,
                                                            double *__restrict__ out_stats, uint32_t *__restrict__ out_n,
                                                            uint32_t *__restrict__ out_idx, double *__restrict__ out_dist,
                                                            double *__restrict__ out_z) {
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
    block_select_ranks<0, true>(row, n, mean, sel, 2, s_hist, s_cand, s_misc, &sq);
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
      block_select_ranks<0>(row, n, 0.0, s3, 1, s_hist, s_cand, s_misc);
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
    block_select_ranks<1>(row, n, median, sm, 1, s_hist, s_cand, s_misc);
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

int launch_summary_large(const double *rows, uint32_t n_rows, uint32_t r1, uint32_t row0, uint32_t keep_at_most,
                         uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx,
                         double *out_dist, double *out_z, hipStream_t st) {
  const uint32_t req_len = keep_at_most ? keep_at_most : r1;
  summary_large_kernel<<<dim3(n_rows), dim3(kLT), 0, st>>>(rows, r1, row0, req_len, max_neighbours, out_stats, out_n,
                                                           out_idx, out_dist, out_z);
  KPOP_LAUNCH_CHECK();
  return 0;
}

}  // namespace kpop

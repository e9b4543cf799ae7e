// summary_large.hip -- per-row distance summaries when a row (r1 distances) does not fit LDS:
// the relatedness-engine case, a query against a database of 10^4..10^6 twisted vectors
// (README.md:1101; lib/Matrix.ml:691-766 with keep_at_most ~ 300).
//
// The r2 x r1 matrix is still never formed: query rows are processed in chunks whose distance
// rows live in a library workspace, and each row is reduced by one 1024-thread block with
//   - two ordered block reductions            mean, sample sd           (lib/Matrix.ml:651-655,657-683)
//   - MSB-first radix SELECT on the f64 bits  upper median, MAD, and the keep_at_most-th smallest
//                                             distance (8 histogram passes each; no sort of the row)
//   - ordered compaction + a small LDS sort   the closest rows, whole tie groups counted (:648-649)
// Differences from the r1 <= 4096 kernel: mean and sd are tree sums (not the reference's
// ascending chain), equal up to rounding (tests hold 1e-10); at most kLargeMaxNb neighbours
// are returned per row (out_n still reports the reference's eff_len).
#include <algorithm>

#include "common.h"

namespace kpop {

constexpr int kLT = 1024;             // threads per row
constexpr uint32_t kLargeMaxNb = 2048;  // neighbours returned per row at most

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

// rank-th smallest (0-based) of the n transformed elements; also how many are strictly smaller
// and how many are equal.  All threads return the same values.
template <int TRANSFORM>
__device__ uint64_t block_select(const double *row, uint32_t n, double centre, uint32_t rank, uint32_t *s_hist,
                                 uint32_t *s_pick, uint32_t *n_less, uint32_t *n_equal) {
  uint64_t prefix = 0, mask = 0;
  uint32_t r = rank;
  for (int shift = 56; shift >= 0; shift -= 8) {
    __syncthreads();
    if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += kLT) {
      const uint64_t k = elem_key<TRANSFORM>(row, i, centre);
      if ((k & mask) == prefix) atomicAdd(&s_hist[(uint32_t)(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t cum = 0, b = 0;
      for (; b < 256; ++b) {
        if (cum + s_hist[b] > r) break;
        cum += s_hist[b];
      }
      if (b == 256) b = 255;  // only reachable with NaNs in the row
      s_pick[0] = b;
      s_pick[1] = r - cum;
      s_pick[2] = s_hist[b];
    }
    __syncthreads();
    prefix |= (uint64_t)s_pick[0] << shift;
    mask |= 255ull << shift;
    r = s_pick[1];
  }
  *n_less = rank - r;
  *n_equal = s_pick[2];
  return prefix;
}

__global__ __launch_bounds__(kLT) void summary_large_kernel(const double *__restrict__ rows, uint32_t r1, uint32_t row0,
                                                            uint32_t req_len, uint32_t max_neighbours,
                                                            double *__restrict__ out_stats, uint32_t *__restrict__ out_n,
                                                            uint32_t *__restrict__ out_idx, double *__restrict__ out_dist,
                                                            double *__restrict__ out_z) {
  __shared__ double s_w[kLT / 64];
  __shared__ uint32_t s_wu[kLT / 64];
  __shared__ uint32_t s_hist[256];
  __shared__ uint32_t s_pick[3];
  __shared__ double s_cd[kLargeMaxNb];
  __shared__ uint32_t s_ci[kLargeMaxNb];
  __shared__ uint32_t s_ti[kLargeMaxNb];  // columns of the tie group at the cut-off distance
  const double *row = rows + (uint64_t)blockIdx.x * r1;
  const uint32_t j = row0 + blockIdx.x;
  const uint32_t n = r1;
  // mean (lib/Matrix.ml:651-655) and sample sd (:657-662,679-683)
  double part = 0.0;
  for (uint32_t i = threadIdx.x; i < n; i += kLT) part = __dadd_rn(part, row[i]);
  const double mean = n ? block_sum(part, s_w) / (double)n : 0.0;
  part = 0.0;
  for (uint32_t i = threadIdx.x; i < n; i += kLT) {
    const double dv = __dsub_rn(row[i], mean);
    part = __dadd_rn(part, __dmul_rn(dv, dv));
  }
  const double ss = block_sum(part, s_w);
  const double sd = (n > 1) ? sqrt(ss / ((double)n - 1.0)) : 0.0;
  // upper median = element n/2 of the sorted row (:645-647); MAD = element n/2 of |d - median| (:671-678)
  uint32_t lt, eq;
  double median = 0.0, mad = 0.0;
  if (n) {
    median = key_f64(block_select<0>(row, n, 0.0, n / 2, s_hist, s_pick, &lt, &eq));
    mad = key_f64(block_select<1>(row, n, median, n / 2, s_hist, s_pick, &lt, &eq));
  }
  // eff_len: groups are added while eff_len < req_len (:648-649)
  uint32_t eff = n;
  if (n && req_len < n) {
    block_select<0>(row, n, 0.0, req_len - 1, s_hist, s_pick, &lt, &eq);
    eff = lt + eq;
  }
  // the M closest, by (distance, column)
  const uint32_t M = min(min(eff, max_neighbours), kLargeMaxNb);
  if (M) {
    const double vM = key_f64(block_select<0>(row, n, 0.0, M - 1, s_hist, s_pick, &lt, &eq));
    const uint32_t n_lt = lt, want_eq = M - lt;
    uint32_t got_lt = 0, got_eq = 0;
    for (uint32_t base = 0; base < n; base += kLT) {  // column order => ties resolved by column, as the multimap does
      const uint32_t i = base + threadIdx.x;
      const double dv = (i < n) ? row[i] : 0.0;
      const bool is_lt = (i < n) && dv < vM, is_eq = (i < n) && dv == vM;
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

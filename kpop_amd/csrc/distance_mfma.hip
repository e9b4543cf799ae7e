// distance_mfma.hip -- the distance step of the large-reference summary on the f64 MATRIX cores, with an exact refinement.
//
// kpop_dev_distance_summary of a few hundred query rows against 10^5 .. 10^6 reference rows (the relatedness engine,
// /root/reference README.md:1101; lib/Matrix.ml:691-766) spent three quarters of its time computing the q x r1 distances in
// the vector pipe, four unfusable f64 operations per pair and dimension (lib/Space.ml:182-205), while the matrix pipe idled.
// Here:
//   1. distance_rows_mfma_kernel: d2 = |a|^2_m + |b|^2_m - 2 a . (b m) with the dot products as f64 MFMAs (v_mfma_f64_16x16x4:
//      a wavefront keeps 64 query rows as A fragments in registers and streams the reference rows through LDS 16 at a time).
//      The values differ from the reference's chain in the last bits -- cancellation: |d2~ - d2| <= G2 = gamma (|a|^2 + |b|^2) --
//      so they are used to LOCATE, never to report: the summary's selection machinery (summary_large.hip) runs on them as it
//      would on exact rows,
//   2. summary_refine_kernel (one block a query row) then makes everything that is reported exact again.  With u = d^2
//      (u = the value itself for the cosine form) every approximate value is within G2 of the exact one, hence
//        - every row NOT within 2 G2 of the approximate req_len-th smallest is certainly not a neighbour: the rows that are
//          (a handful) are recomputed with the reference's sequential chain, sorted by (distance, column) and cut with their ties;
//        - every value more than 2 G2 below (above) the approximate median is certainly below (above) the exact one: the exact
//          median is the exact (n/2 - #below)-th smallest of the few in between; the MAD likewise around both of its edges;
//      mean and standard deviation stay sums of the approximate values (1e-13 relative: the tests allow 1e-10 for rows this long).
//      A row whose bands overflow or whose ranks do not fall inside them is FLAGGED and redone from exact distance rows by the
//      kernels that were there before (the same fall-back the bracket path uses).
// Euclidean and cosine forms (Minkowski's |x|^p is no contraction), up to 128 dimensions, req_len <= max_neighbours <= 2,048.
#include <algorithm>

#include "common.h"
#include "summary_types.h"
#include "space_ops.h"

namespace kpop {

using f64x4m = __attribute__((ext_vector_type(4))) double;

// out[row] = sum_c m_c x_c^2; scaled (optional) = x m; *smax (optional) = the largest sum (one atomic a block)
__global__ __launch_bounds__(256) void row_sumsq_kernel(const double *__restrict__ x, uint32_t rows, uint32_t n_dims, const double *__restrict__ metric,
                                                        double *__restrict__ out, double *__restrict__ scaled, unsigned long long *__restrict__ smax) {
  __shared__ unsigned long long s_max;
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
  const uint32_t l = threadIdx.x & 15u;
  // (a few thousand blocks at most: every block ends with an atomic on the ONE word of the maximum, and 62,500 of those took 0.7 ms)
  for (uint32_t row = blockIdx.x * 16 + (threadIdx.x >> 4); row < ((rows + 15u) & ~15u); row += gridDim.x * 16) {
  double acc = 0.0;
  if (row < rows) {
    for (uint32_t c0 = 0; c0 < n_dims; c0 += 128) {  // (eight loads in flight a lane: 128 dimensions a sweep)
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t c = c0 + l + 16u * k;
        v[k] = c < n_dims ? x[(uint64_t)row * n_dims + c] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t c = c0 + l + 16u * k;
        if (c < n_dims) {
          const double m = metric[c];
          acc += v[k] * v[k] * m;
          if (scaled) scaled[(uint64_t)row * n_dims + c] = v[k] * m;
        }
      }
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 16);
  if (row < rows && l == 0) {
    out[row] = acc;
    if (smax) atomicMax(&s_max, (unsigned long long)__double_as_longlong(acc));  // (sums of squares: not negative, their bit patterns order as they do)
  }
  }
  __syncthreads();
  if (smax && threadIdx.x == 0) atomicMax(smax, s_max);
}

// sqrt(u), u >= 0, to an ulp or two in ten operations (the f64 vector operations of the epilogue and the f64 MFMAs share the
// same units: what the epilogue spends the matrix pipe waits for).  rsq's 23 bits, one Newton step, one correction with the
// residual; 0 -> 0; values below 1e-290 (their rsq overflows) -> 0: they ARE zero at the accuracy of these rows
__device__ __forceinline__ double sqrt_fast(double u) {
  const double y = __builtin_amdgcn_rsq(fmax(u, 1e-290));
  double g = u * y;
  const double h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  const double r2 = __builtin_fma(-g, g, u);
  return __builtin_fma(r2, h, g);
}

// rows~[j][i] for 64 MI query rows a block (16 MI a wavefront, their A fragments in registers for the whole kernel) and a run of
// 16-row tiles of the reference set (the B fragments, through LDS, double-buffered; the next tile's loads fly under the MFMAs).
// MI = 2 for up to 64 dimensions, 132 registers: three wavefronts a SIMD, one's square roots and stores under another's MFMAs.
template <int KIND, int KS, int MI>  // KS steps of 4 dimensions: n_dims <= 4 KS
__global__ __launch_bounds__(256) void distance_rows_mfma_kernel(const double *__restrict__ a, uint32_t r1, const double *__restrict__ bm, uint32_t q, uint32_t n_dims,
                                                                 const double *__restrict__ sa, const double *__restrict__ sb, double *__restrict__ out,
                                                                 uint32_t tiles_per_block, const double *__restrict__ ia = nullptr) {
  // (ia: the reference rows come as they are, still to be divided by their norms -- sa is the sum of squares of the row AS IT WILL BE, ia
  // the norm's reciprocal, a dot product is scaled where it comes out; nullptr: the rows are what they are)
  constexpr int TS = 4 * KS + 2;  // a staged row's doubles: + 2 spreads a half-wavefront's 16 rows over the banks
  __shared__ double s_tile[2][16][TS];
  __shared__ double s_sa[2][16], s_ia[2][16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t l15 = (uint32_t)lane & 15u, l4 = (uint32_t)lane >> 4;
  const uint32_t j0 = blockIdx.y * (64u * MI) + (uint32_t)wv * (16u * MI);
  double af[MI][KS];  // A fragments: lane = (query row l15 of the M tile, dimension 4 ks + l4)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint32_t row = j0 + 16u * mi + l15, c = 4u * ks + l4;
      af[mi][ks] = (row < q && c < n_dims) ? bm[(uint64_t)row * n_dims + c] : 0.0;
    }
  double sbv[MI][4];  // the norms of the rows a lane's accumulators belong to: row 16 mi + l4 + 4 rr
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const uint32_t row = j0 + 16u * mi + l4 + 4u * rr;
      sbv[mi][rr] = row < q ? sb[row] : 0.0;
    }
  const uint32_t n_tiles = (r1 + 15u) / 16u;
  const uint32_t t0 = blockIdx.x * tiles_per_block, t1 = min(n_tiles, t0 + tiles_per_block);
  if (t0 >= t1) return;
  const bool rows_full = j0 + 16u * MI <= q;  // (wavefront-uniform)
  constexpr uint32_t PER = (16u * 4u * KS + 255u) / 256u;  // doubles of a tile a thread moves
  double pre[PER], pre_sa = 0.0, pre_ia = 1.0;
  auto fetch = [&](uint32_t t) {
#pragma unroll
    for (uint32_t e = 0; e < PER; ++e) {
      const uint32_t idx = threadIdx.x + 256u * e, r = idx / (4u * KS), c = idx % (4u * KS);
      const uint32_t i = min(16u * t + r, r1 - 1u);  // (rows past the end: the last row again, never stored)
      pre[e] = c < n_dims ? a[(uint64_t)i * n_dims + c] : 0.0;
    }
    if (threadIdx.x < 16) {
      pre_sa = sa[min(16u * t + threadIdx.x, r1 - 1u)];
      if (ia) pre_ia = ia[min(16u * t + threadIdx.x, r1 - 1u)];
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (uint32_t e = 0; e < PER; ++e) {
      const uint32_t idx = threadIdx.x + 256u * e, r = idx / (4u * KS), c = idx % (4u * KS);
      s_tile[buf][r][c] = pre[e];
    }
    if (threadIdx.x < 16) {
      s_sa[buf][threadIdx.x] = pre_sa;
      s_ia[buf][threadIdx.x] = pre_ia;
    }
  };
  static_assert((16u * 4u * KS) % 256u == 0, "a tile is a whole number of sweeps of the block");
  fetch(t0);
  put(0);
  __syncthreads();
  for (uint32_t t = t0; t < t1; ++t) {
    const int buf = (int)((t - t0) & 1u);
    fetch(min(t + 1, t1 - 1));  // (in flight under this tile's MFMAs; the last tile is fetched twice rather than branched around)
    f64x4m acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[mi] = f64x4m{0.0, 0.0, 0.0, 0.0};
    if constexpr (KS <= 16) {  // the tile's B fragments first, then the MFMAs back to back
      double bf[KS];  // B fragment: lane = (dimension 4 ks + l4, reference row l15)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bf[ks] = s_tile[buf][l15][4 * ks + (int)l4];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi][ks], bf[ks], acc[mi], 0, 0, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const double bf = s_tile[buf][l15][4 * ks + (int)l4];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi][ks], bf, acc[mi], 0, 0, 0);
      }
    }
    const double sai = s_sa[buf][l15], m2i = -2.0 * s_ia[buf][l15];  // (-2 times the scale of the tile's reference row: 1 unless ia)
    __builtin_amdgcn_s_setprio(2);  // (the square roots and stores of this tile ahead of the other wavefronts' MFMAs: they end the tile, the MFMAs only fill the pipe)
    put(buf ^ 1);  // (its readers passed the barrier that ended the previous tile)
    const uint32_t i = 16u * t + l15;
    if (rows_full && 16u * t + 16u <= r1) {  // (uniform) straight-line stores
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {  // lane holds query rows l4 + 4 rr of the M tile, reference row l15
          const uint32_t row = j0 + 16u * mi + l4 + 4u * rr;
          double u = sai + sbv[mi][rr] + m2i * acc[mi][rr];
          u = u > 0.0 ? u : 0.0;
          out[(uint64_t)row * r1 + i] = KIND == KPOP_EUCLIDEAN ? sqrt_fast(u) : u * 0.5;
        }
    } else {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const uint32_t row = j0 + 16u * mi + l4 + 4u * rr;
          double u = sai + sbv[mi][rr] + m2i * acc[mi][rr];
          u = u > 0.0 ? u : 0.0;
          if (row < q && i < r1) out[(uint64_t)row * r1 + i] = KIND == KPOP_EUCLIDEAN ? sqrt_fast(u) : u * 0.5;
        }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }
}

// The summary's ONE pass over the pairs, inside the contraction: the kernel above with the operands of the MFMA exchanged -- a lane then
// holds ONE query row (l15 of its M tile) x four reference rows of the tile, so the query's thresholds (FusedThr: the median's bracket,
// the MAD's bands, the neighbours' cut, from the distances to a sample) live in 16 registers a query -- and every distance classified
// where it is made.  No distance row is written (round 5's path wrote 2 GB of them per 256 queries x 1M and read them back).
// What a lane sees of a query row in a stripe of 2,048 reference rows -- 512 distances -- is a SUB-STRIPE of its own: its counts and
// sums in registers for the whole stripe, its candidates (of the median's bracket, of the MAD's bands: ~12 in 100) stored one after
// the other in its own 512 slots of the row's segment with their columns, ONE record (StripeRec) at the end.  No shared counter, no
// atomic, no exchange between lanes on the way (the first version of this kernel filed candidates through LDS counters, one atomic
// round trip each, a lane at a time: 4,650 of its 6,700 cycles a tile went there and it lost to the path that writes the rows).
// Seven comparisons classify a distance: the counts of d < lo, d <= lo, d < hi, d <= hi (what lies below, at and inside the median's
// bracket are differences of these) and of the inner region; a candidate is inside the bracket or in a band.  Neighbour candidates
// (one in a thousand) go to the row's list with an atomic.  fused_finish_kernel<true> reads the records (sub-stripes of 512),
// summary_refine_kernel makes what is reported exact.  lib/Matrix.ml:691-766.
constexpr uint32_t kSubStripe = kStripe / 4;
constexpr uint32_t kNbStage = 1024;  // neighbour candidates a block keeps in LDS until its stripe is done (~260 expected: 128 queries x 2,048 rows at one in a thousand)
template <int KIND, int KS>
__global__ __launch_bounds__(256) void summary_select_mfma_kernel(const double *__restrict__ a, uint32_t r1, const double *__restrict__ bm, uint32_t q, uint32_t n_dims,
                                                                  const double *__restrict__ sa, const double *__restrict__ sb, const FusedThr *__restrict__ thr,
                                                                  double *__restrict__ seg, uint32_t *__restrict__ seg_i, StripeRec *__restrict__ rec,
                                                                  double *__restrict__ part, RowCounts *__restrict__ cnt, uint32_t *__restrict__ nb_idx,
                                                                  double *__restrict__ nb_d, uint32_t n_stripes) {
  constexpr int MI = 2, TS = 4 * KS + 2;
  __shared__ double s_tile[2][16][TS];
  __shared__ double s_sa[2][16];
  // neighbour candidates wait here for the end of the stripe: filed straight into the rows' lists they cost an atomic that RETURNS -- the
  // wavefront then waits for everything it has in flight, the next tile's loads included, four tiles in ten
  __shared__ double s_nbd[kNbStage];
  __shared__ uint32_t s_nbj[kNbStage], s_nbi[kNbStage];
  __shared__ uint32_t s_nbn;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t l15 = (uint32_t)lane & 15u, l4 = (uint32_t)lane >> 4;
  const uint32_t stripe = blockIdx.x, j0 = blockIdx.y * (64u * MI) + (uint32_t)wv * (16u * MI);
  if (threadIdx.x == 0) s_nbn = 0;
  double qf[MI][KS];  // the query rows' fragments: lane = (query row l15 of the tile, dimension 4 ks + l4)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint32_t row = j0 + 16u * mi + l15, c = 4u * ks + l4;
      qf[mi][ks] = (row < q && c < n_dims) ? bm[(uint64_t)row * n_dims + c] : 0.0;
    }
  FusedThr T[MI];
  double sbq[MI];
  const uint32_t ref0 = stripe * kStripe, ref1 = min(r1, ref0 + kStripe);
  uint64_t woff[MI];  // where this lane's next candidate of query mi goes (seg / seg_i share the index)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const uint32_t row = min(j0 + 16u * mi + l15, q - 1u);
    T[mi] = thr[row];
    sbq[mi] = sb[row];
    // (a row's segments: n_stripes x kStripe slots -- the row length rounded up to whole stripes, so that the four sub-stripes of a
    // last, partial stripe have their 512 slots each; rows past the last one file nothing: their distances are made NaNs)
    woff[mi] = (uint64_t)row * ((uint64_t)n_stripes * kStripe) + ref0 + l4 * kSubStripe;
  }
  const uint32_t t0 = ref0 / 16u, t1 = (ref1 + 15u) / 16u;
  constexpr uint32_t PER = (16u * 4u * KS + 255u) / 256u;
  double pre[PER], pre_sa = 0.0;
  auto fetch = [&](uint32_t t) {
#pragma unroll
    for (uint32_t e = 0; e < PER; ++e) {
      const uint32_t idx = threadIdx.x + 256u * e, r = idx / (4u * KS), c = idx % (4u * KS);
      const uint32_t i = min(16u * t + r, r1 - 1u);
      pre[e] = c < n_dims ? a[(uint64_t)i * n_dims + c] : 0.0;
    }
    if (threadIdx.x < 16) pre_sa = sa[min(16u * t + threadIdx.x, r1 - 1u)];
  };
  auto put = [&](int buf) {
#pragma unroll
    for (uint32_t e = 0; e < PER; ++e) {
      const uint32_t idx = threadIdx.x + 256u * e, r = idx / (4u * KS), c = idx % (4u * KS);
      s_tile[buf][r][c] = pre[e];
    }
    if (threadIdx.x < 16) s_sa[buf][threadIdx.x] = pre_sa;
  };
  double sum[MI], sq[MI];
  uint32_t c_lt[MI], c_le[MI], c_lth[MI], c_leh[MI], c_in[MI], c_k[MI];  // d < lo, d <= lo, d < hi, d <= hi, inner region, candidates filed
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    sum[mi] = sq[mi] = 0.0;
    c_lt[mi] = c_le[mi] = c_lth[mi] = c_leh[mi] = c_in[mi] = c_k[mi] = 0u;
  }
  const bool rows_full = j0 + 16u * MI <= q;  // (wavefront-uniform)
  // one distance, where it is made: counted, summed, filed if it is a candidate.  (A NaN -- a pair that is not of the job: a row past
  // the last one, a column past the stripe's end -- is below, at and inside nothing: counted nowhere, filed nowhere.)
  auto take = [&](int mi, double d, bool valid, uint32_t i) {
    const FusedThr &Tq = T[mi];
    sum[mi] = __dadd_rn(sum[mi], valid ? d : 0.0);
    const double dvv = valid ? __dsub_rn(d, Tq.mhat) : 0.0;
    sq[mi] = __dadd_rn(sq[mi], __dmul_rn(dvv, dvv));
    if (!valid) d = __longlong_as_double(0x7FF8000000000000ll);
    const bool lt = d < Tq.lo, le = d <= Tq.lo, lth = d < Tq.hi, leh = d <= Tq.hi;
    const bool in = d > Tq.Lin && d < Tq.Uin;
    c_lt[mi] += lt ? 1u : 0u;
    c_le[mi] += le ? 1u : 0u;
    c_lth[mi] += lth ? 1u : 0u;
    c_leh[mi] += leh ? 1u : 0u;
    c_in[mi] += in ? 1u : 0u;
    if ((!le && lth) || (!in && d >= Tq.Llo && d <= Tq.Uhi)) {
      seg[woff[mi]] = d;
      seg_i[woff[mi]] = i;
      ++woff[mi];
      ++c_k[mi];
    }
    if (d <= Tq.cut) {
      const uint32_t j = j0 + 16u * (uint32_t)mi + l15;
      const uint32_t at = atomicAdd(&s_nbn, 1u);
      if (at < kNbStage) {
        s_nbd[at] = d;
        s_nbj[at] = j;
        s_nbi[at] = i;
      } else {  // (the stage is full: the long way)
        const uint32_t g = atomicAdd(&cnt[j].n_nb, 1u);
        if (g < kNbCap) {
          nb_idx[(uint64_t)j * kNbCap + g] = i;
          nb_d[(uint64_t)j * kNbCap + g] = d;
        }
      }
    }
  };
  fetch(t0);
  put(0);
  __syncthreads();
  for (uint32_t t = t0; t < t1; ++t) {
    const int buf = (int)((t - t0) & 1u);
    fetch(min(t + 1, t1 - 1));
    f64x4m acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[mi] = f64x4m{0.0, 0.0, 0.0, 0.0};
    if constexpr (KS <= 16) {
      double rf[KS];  // the reference tile's fragments: lane = (reference row l15, dimension 4 ks + l4)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) rf[ks] = s_tile[buf][l15][4 * ks + (int)l4];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(rf[ks], qf[mi][ks], acc[mi], 0, 0, 0);  // D[reference row][query row]
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const double rf = s_tile[buf][l15][4 * ks + (int)l4];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(rf, qf[mi][ks], acc[mi], 0, 0, 0);
      }
    }
    double sai[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) sai[rr] = s_sa[buf][l4 + 4u * rr];
    __builtin_amdgcn_s_setprio(2);
    put(buf ^ 1);
    if (rows_full && 16u * t + 16u <= ref1) {  // (uniform) every element of the tile is a pair of the job
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {  // lane holds reference rows l4 + 4 rr of the tile, query row l15 of the M tile
          double u = sai[rr] + sbq[mi] - 2.0 * acc[mi][rr];
          u = u > 0.0 ? u : 0.0;
          const double d = KIND == KPOP_EUCLIDEAN ? sqrt_fast(u) : u * 0.5;
          take(mi, d, true, 16u * t + l4 + 4u * (uint32_t)rr);
        }
    } else {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const uint32_t i = 16u * t + l4 + 4u * (uint32_t)rr;
          double u = sai[rr] + sbq[mi] - 2.0 * acc[mi][rr];
          u = u > 0.0 ? u : 0.0;
          const double d = KIND == KPOP_EUCLIDEAN ? sqrt_fast(u) : u * 0.5;
          take(mi, d, j0 + 16u * (uint32_t)mi + l15 < q && i < ref1, i);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }
  // a record a lane: sub-stripe 4 stripe + l4 of its query rows
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const uint32_t j = j0 + 16u * mi + l15;
    if (j < q) {
      const FusedThr &Tq = T[mi];
      const uint32_t eqlo = c_le[mi] - c_lt[mi], nmed = Tq.lo < Tq.hi ? c_lth[mi] - c_le[mi] : 0u, eqhi = Tq.hi != Tq.lo ? c_leh[mi] - c_lth[mi] : 0u;
      const uint64_t at = (uint64_t)j * (4u * n_stripes) + 4u * stripe + l4;
      rec[at] = StripeRec{c_lt[mi] | (eqlo << 16), eqhi | (nmed << 16), c_in[mi], c_k[mi]};
      part[at * 2 + 0] = sum[mi];
      part[at * 2 + 1] = sq[mi];
    }
  }
  // the stripe's neighbour candidates into their rows' lists
  const uint32_t n_st = min(s_nbn, kNbStage);  // (the loop's last barrier stands between the last append and this read)
  for (uint32_t e = threadIdx.x; e < n_st; e += 256) {
    const uint32_t j = s_nbj[e];
    const uint32_t g = atomicAdd(&cnt[j].n_nb, 1u);
    if (g < kNbCap) {
      nb_idx[(uint64_t)j * kNbCap + g] = s_nbi[e];
      nb_d[(uint64_t)j * kNbCap + g] = s_nbd[e];
    }
  }
}

// the reference's chain for one pair, dimension by dimension (lib/Space.ml:182-205; distance.hip)
template <int KIND>
__device__ __forceinline__ double exact_pair_rows(const double *__restrict__ arow, const double *__restrict__ brow, const double *__restrict__ metric, uint32_t n_dims, double p) {
  double acc = 0.0;
  for (uint32_t c = 0; c < n_dims; ++c) {
    const double diff = __dsub_rn(arow[c], brow[c]);
    acc = __dadd_rn(acc, component<KIND>(diff, metric[c], p));
  }
  return scale_distance<KIND>(acc, p);
}

// ... of two rows that are still to be divided by their norms (x /. norm, lib/Matrix.ml:201-202: the same division element by element
// that the normalised copy holds, so the same bits)
template <int KIND>
__device__ __forceinline__ double exact_pair_rows_div(const double *__restrict__ arow, double na, const double *__restrict__ brow, double nb,
                                                      const double *__restrict__ metric, uint32_t n_dims, double p) {
  double acc = 0.0;
  for (uint32_t c = 0; c < n_dims; ++c) {
    const double diff = __dsub_rn(arow[c] / na, brow[c] / nb);
    acc = __dadd_rn(acc, component<KIND>(diff, metric[c], p));
  }
  return scale_distance<KIND>(acc, p);
}

// The same distances for ANY number of dimensions and any two sets of rows, as a tiled contraction: a block takes 128 query rows x 128
// reference rows, both operands staged through LDS sixteen dimensions at a time (row-major rows, lanes along the dimensions: whole
// 128-byte lines; the next chunk's loads fly under this chunk's 64 MFMAs a wavefront), a wavefront 64 x 64 = 4 x 4 accumulator tiles.
// This is what the reference's own large runs are: 650 K samples x 1,636 classes x 1,635 dimensions (README.md:1054-1060), 300
// neighbours in a database of 650 K x 1,635 (README.md:1101).
// GUARD (kpop_dev_distance_rowwise: the values themselves are the result): d^2 = |a|^2 + |b|^2 - 2 a.b loses what the two rows have in
// common to cancellation -- a pair whose d^2~ is below `tau` of |a|^2 + |b|^2 is recomputed with the reference's chain, dimension by
// dimension (lib/Space.ml:182-205), the others are within (D + 3) 2^-53 / (4 tau) of it, relatively (1.8e-12 at 1,635 dimensions,
// tau = 0.025; measured: a few 1e-14).  Without it (the summary's rows) the values only LOCATE: see the head of the file.
// tile edge; dimensions a chunk; rows' stride in the LDS panels, in PAIRS of dimensions: a panel is [pair of dimensions][row][2] -- a thread
// loads two neighbouring dimensions of a row at once (16 bytes) and stores them with one ds_write_b128; 130 = 2 mod 16 spreads the
// sixteen lanes of a quarter-wavefront (8 pairs x 2 rows) over all the banks, and a fragment read (16 rows x 2 dimensions of a pair)
// covers them exactly
constexpr int kDT = 128, kDK = 16, kDS = kDT + 2;
constexpr size_t kDgLds = (size_t)4 * (kDK / 2) * kDS * 16;  // two buffers x two panels

template <int KIND, bool GUARD>
__global__ __launch_bounds__(256, 2) void distance_gemm_mfma_kernel(const double *__restrict__ a, uint32_t r1, const double *__restrict__ bm, const double *__restrict__ b,
                                                                 uint32_t q, uint32_t n_dims, const double *__restrict__ metric, double p,
                                                                 const double *__restrict__ sa, const double *__restrict__ sb, double *__restrict__ out,
                                                                 uint32_t tiles_m, uint32_t tiles_n, int m_fast, double tau, const double *__restrict__ a_plain,
                                                                 const double *__restrict__ na = nullptr, const double *__restrict__ nb = nullptr,
                                                                 const double *__restrict__ ia = nullptr, const double *__restrict__ ib = nullptr) {
  // (na, nb, ia, ib -- all or none: the operands are still to be divided by their rows' norms na / nb; sa and sb are the sums of squares of
  // the rows AS THEY WILL BE, ia / ib the norms' reciprocals: the contraction runs on the rows as they are and a dot product is scaled
  // where it comes out -- no normalised copy of either operand is made: 0.83 ms of 10 for 100,000 x 1,636 x 1,635)
  // (a, bm: the two panels' sources, ONE of them times the metric -- the caller's choice, the smaller one; a_plain, b: the operands as they
  // are, for the pairs the guard recomputes)
  // two buffers a panel (74 KB a block in all, two blocks a CU): chunk c + 1 is written while chunk c is multiplied, ONE barrier a chunk
  extern __shared__ __attribute__((aligned(16))) double dg_lds[];
  using Panel = double[kDK / 2][kDS][2];
  Panel *Qs = reinterpret_cast<Panel *>(dg_lds);                              // [buffer][pair of dimensions][query row of the tile][2]
  Panel *Rs = reinterpret_cast<Panel *>(dg_lds + 2 * (kDK / 2) * kDS * 2);   // [buffer][pair of dimensions][reference row of the tile][2]
  // Workgroups are dealt to the eight XCDs in turn; the blocks of one XCD take a contiguous run of tiles, decoded with the short
  // side fastest: the tiles that share a panel are neighbours in time on ONE L2.
  uint32_t wgid = blockIdx.x;
  {
    const uint32_t nwg = gridDim.x, q8 = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8;
    wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + blockIdx.x / 8;
  }
  const uint32_t bx = m_fast ? wgid % tiles_m : wgid / tiles_n;  // query tile
  const uint32_t by = m_fast ? wgid / tiles_m : wgid % tiles_n;  // reference tile
  const uint32_t m0 = bx * kDT, n0 = by * kDT;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
  f64x4m acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x4m{0.0, 0.0, 0.0, 0.0};
  constexpr int kQ = kDK / 4;  // 16-byte loads a thread, panel and chunk
  struct __attribute__((aligned(8))) D2 {
    double x, y;
  };
  D2 rq[kQ], rr[kQ];
  const uint32_t kp = threadIdx.x & (kDK / 2 - 1), rbase = threadIdx.x / (kDK / 2);  // the thread's pair of dimensions of a chunk, its first row of a panel
  constexpr uint32_t kRows = 256 / (kDK / 2);                                      // rows between two of its loads
  // Where the thread's loads of a chunk come from: ELEMENT OFFSETS from the tile's first row (rows past the operand's end clamped to its
  // last row, zeroed when they are put), the same for every chunk -- the chunk moves the wavefront-uniform base, so a load costs no
  // address arithmetic of its own.  What is loaded is not looked at before it is put into LDS, AFTER the chunk's MFMAs: masking the
  // registers right after the loads (the first version) made every chunk wait for its successor's loads before its first MFMA --
  // 2.2 of 12.0 ms on 100,000 x 1,636 x 1,635.
  uint32_t oq[kQ], orr[kQ];
#pragma unroll
  for (int u = 0; u < kQ; ++u) {
    const uint32_t row = rbase + kRows * u;
    oq[u] = (min(m0 + row, q - 1u) - m0) * n_dims + 2u * kp;
    orr[u] = (min(n0 + row, r1 - 1u) - n0) * n_dims + 2u * kp;
  }
  const double *qt = bm + (uint64_t)m0 * n_dims, *rt = a + (uint64_t)n0 * n_dims;  // (uniform)
  const bool interior = m0 + kDT <= q && n0 + kDT <= r1;                           // (uniform) every row of both panels exists
  auto prefetch = [&](uint32_t k0) {
    const double *qk = qt + k0, *rk = rt + k0;
    if (k0 + kDK <= n_dims) {  // (uniform) a whole chunk: 16 bytes a load (rows are 8-byte aligned: the hardware does not ask for more)
#pragma unroll
      for (int u = 0; u < kQ; ++u) {
        rq[u] = *reinterpret_cast<const D2 *>(qk + oq[u]);
        rr[u] = *reinterpret_cast<const D2 *>(rk + orr[u]);
      }
    } else {  // the last, partial chunk: dimensions past the end read the row's last one (zeroed when they are put)
      const uint32_t c0 = k0 + 2u * kp, b0 = c0 < n_dims ? 0u : c0 - (n_dims - 1u), b1 = c0 + 1u < n_dims ? 0u : c0 + 1u - (n_dims - 1u);
#pragma unroll
      for (int u = 0; u < kQ; ++u) {
        rq[u].x = qk[oq[u] - b0];
        rq[u].y = qk[oq[u] + 1u - b1];
        rr[u].x = rk[orr[u] - b0];
        rr[u].y = rk[orr[u] + 1u - b1];
      }
    }
  };
  auto put = [&](int buf, uint32_t k0) {  // (k0: the chunk the registers hold)
    if (!interior || k0 + kDK > n_dims) {  // (uniform) an edge: what lies outside the operands is zero
      const bool in0 = k0 + 2u * kp < n_dims, in1 = k0 + 2u * kp + 1u < n_dims;
#pragma unroll
      for (int u = 0; u < kQ; ++u) {
        const uint32_t row = rbase + kRows * u;
        const bool qin = m0 + row < q, rin = n0 + row < r1;
        rq[u].x = (in0 && qin) ? rq[u].x : 0.0;
        rq[u].y = (in1 && qin) ? rq[u].y : 0.0;
        rr[u].x = (in0 && rin) ? rr[u].x : 0.0;
        rr[u].y = (in1 && rin) ? rr[u].y : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < kQ; ++u) {
      double *dq = Qs[buf][kp][rbase + kRows * u], *dr = Rs[buf][kp][rbase + kRows * u];
      *reinterpret_cast<double2 *>(dq) = double2{rq[u].x, rq[u].y};
      *reinterpret_cast<double2 *>(dr) = double2{rr[u].x, rr[u].y};
    }
  };
  prefetch(0);
  put(0, 0);
  __syncthreads();
  for (uint32_t k0 = 0, c = 0; k0 < n_dims; k0 += kDK, ++c) {
    const int buf = (int)(c & 1u);
    const bool more = k0 + kDK < n_dims;  // (uniform)
    if (more) prefetch(k0 + kDK);
#pragma unroll
    for (int ks = 0; ks < kDK; ks += 4) {
      double fa[4], fb[4];
      const int kr = ks + (lane >> 4), cc = lane & 15;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = Qs[buf][kr >> 1][wm + t * 16 + cc][kr & 1];
        fb[t] = Rs[buf][kr >> 1][wn + t * 16 + cc][kr & 1];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) put(buf ^ 1, k0 + kDK);  // (its readers passed the barrier that ended the chunk before)
    __syncthreads();
  }
  // lane holds query rows (l >> 4) + 4 r of accumulator tile i, reference row l & 15 of tile j
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t col = n0 + wn + j * 16 + (lane & 15);
    const double sai = sa[min(col, r1 - 1u)];
    const double iai = ia ? ia[min(col, r1 - 1u)] : 1.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t row = m0 + wm + i * 16 + (lane >> 4) + 4 * r;
        if (row < q && col < r1) {
          const double sbj = sb[row];
          const double dot = (ia || ib) ? acc[i][j][r] * (ib ? iai * ib[row] : iai) : acc[i][j][r];
          double u = sai + sbj - 2.0 * dot;
          u = u > 0.0 ? u : 0.0;
          double d;
          if (GUARD && u < tau * (sai + sbj))
            d = na ? exact_pair_rows_div<KIND>(a_plain + (uint64_t)col * n_dims, na[col], b + (uint64_t)row * n_dims, nb[row], metric, n_dims, p)
                   : exact_pair_rows<KIND>(a_plain + (uint64_t)col * n_dims, b + (uint64_t)row * n_dims, metric, n_dims, p);
          else
            d = KIND == KPOP_EUCLIDEAN ? sqrt_fast(u) : u * 0.5;  // (an ulp or two: far inside what the contraction itself leaves)
          out[(uint64_t)row * r1 + col] = d;
        }
      }
  }
}

// the RowCounts of summary_large.hip as this file sees them: 48 bytes a row, `fail` the tenth word
constexpr uint32_t kRowCountsWords = 12, kRowCountsFail = 9;
constexpr uint32_t kRefNb = 3072, kRefMed = 1024, kRefMad = 2048;  // room of the three bands (rows of the reference set)

template <int KIND>
__device__ __forceinline__ double exact_pair(const double *__restrict__ arow, const double *__restrict__ brow, const double *__restrict__ metric, uint32_t n_dims, double p) {
  double acc = 0.0;
  for (uint32_t c = 0; c < n_dims; ++c) {  // the reference's chain, dimension by dimension (lib/Space.ml:182-205; distance.hip)
    const double diff = __dsub_rn(arow[c], brow[c]);
    acc = __dadd_rn(acc, component<KIND>(diff, metric[c], p));
  }
  return scale_distance<KIND>(acc, p);
}

// ... of a reference row that is still to be divided by its norm (x /. norm element by element: the bits the normalised copy would hold)
template <int KIND>
__device__ __forceinline__ double exact_pair_adiv(const double *__restrict__ arow, double na, const double *__restrict__ brow, const double *__restrict__ metric,
                                                  uint32_t n_dims, double p) {
  double acc = 0.0;
  for (uint32_t c = 0; c < n_dims; ++c) {
    const double diff = __dsub_rn(__ddiv_rn(arow[c], na), brow[c]);
    acc = __dadd_rn(acc, component<KIND>(diff, metric[c], p));
  }
  return scale_distance<KIND>(acc, p);
}

// The same chain by a WAVEFRONT (rows of hundreds of dimensions: a thread a pair walks its own row, 8 bytes a step, every load a line of
// its own -- 2.2 ms of a 14 ms summary of 256 x 650,000 x 1,635): 64 consecutive dimensions a sweep, a lane each -- one coalesced read of
// both rows, the terms in parallel -- then the 64 terms added to the sum ONE AFTER THE OTHER in dimension order (every lane adds the
// same values in the same order and ends with the same sum): the reference's order of additions, lib/Space.ml:182-205.
// na = 0: the reference row is what it is; otherwise it is divided by na element by element as it goes.
// The exact chain by a WAVEFRONT (rows of hundreds of dimensions: a thread a pair walks its own row, 8 bytes a step, every load a line of
// its own -- 2.2 ms of a 14 ms summary of 256 x 650,000 x 1,635): 64 consecutive dimensions a sweep, a lane each -- one coalesced read of
// the rows, the terms in parallel -- then the 64 terms added to the sum ONE AFTER THE OTHER in dimension order (every lane adds the same
// values, read off the lanes with v_readlane, in the same order and ends with the same sum): the reference's order of additions,
// lib/Space.ml:182-205.  na = 0: the reference row is what it is; otherwise it is divided by na element by element as it goes.
// Two pairs at a time (two independent chains of additions; measured level with one: three instructions a term and pair are what it
// costs, 1.0 ms for 256 rows x ~600 band rows x 1,635 against 2.2 ms a thread a pair)
template <int KIND>
__device__ __forceinline__ void exact_pair_wave2(const double *__restrict__ arow0, double na0, const double *__restrict__ arow1, double na1,
                                                 const double *__restrict__ brow, const double *__restrict__ metric, uint32_t n_dims, double p, int lane, double *d0,
                                                 double *d1) {
  double acc0 = 0.0, acc1 = 0.0;
  // (the next sweep's loads are in flight while this sweep's terms are added up)
  uint32_t c = (uint32_t)lane;
  double a0n = c < n_dims ? arow0[c] : 0.0, a1n = c < n_dims ? arow1[c] : 0.0, b_n = c < n_dims ? brow[c] : 0.0, m_n = c < n_dims ? metric[c] : 0.0;
  for (uint32_t c0 = 0; c0 < n_dims; c0 += 64) {
    const double a0c = a0n, a1c = a1n, b_c = b_n, m_c = m_n;
    const uint32_t cn = c0 + 64u + (uint32_t)lane;
    if (cn < n_dims) {
      a0n = arow0[cn];
      a1n = arow1[cn];
      b_n = brow[cn];
      m_n = metric[cn];
    }
    double t0 = 0.0, t1 = 0.0;
    if (c0 + (uint32_t)lane < n_dims) {
      const double av0 = na0 != 0.0 ? __ddiv_rn(a0c, na0) : a0c, av1 = na1 != 0.0 ? __ddiv_rn(a1c, na1) : a1c;
      t0 = component<KIND>(__dsub_rn(av0, b_c), m_c, p);
      t1 = component<KIND>(__dsub_rn(av1, b_c), m_c, p);
    }
    const uint32_t lim = min(64u, n_dims - c0);
    const int t0lo = __double2loint(t0), t0hi = __double2hiint(t0), t1lo = __double2loint(t1), t1hi = __double2hiint(t1);
    if (lim == 64u) {
#pragma unroll
      for (int l = 0; l < 64; ++l) {  // (v_readlane_b32: the terms of lane l as scalars, no trip through the LDS crossbar)
        acc0 = __dadd_rn(acc0, __hiloint2double(__builtin_amdgcn_readlane(t0hi, l), __builtin_amdgcn_readlane(t0lo, l)));
        acc1 = __dadd_rn(acc1, __hiloint2double(__builtin_amdgcn_readlane(t1hi, l), __builtin_amdgcn_readlane(t1lo, l)));
      }
    } else {
      for (uint32_t l = 0; l < lim; ++l) {
        acc0 = __dadd_rn(acc0, __shfl(t0, (int)l, 64));
        acc1 = __dadd_rn(acc1, __shfl(t1, (int)l, 64));
      }
    }
  }
  *d0 = scale_distance<KIND>(acc0, p);
  *d1 = scale_distance<KIND>(acc1, p);
}

__device__ __forceinline__ bool pair_less(double da, uint32_t ia, double db, uint32_t ib) { return da < db || (da == db && ia < ib); }

// One block a query row: see the head of the file.  rows: the approximate distances [q][r1]; stats etc.: what the summary made of
// them (overwritten where exactness matters: median, MAD, the neighbours); rc: RowCounts (its `fail` set for a row left to the fall-back)
template <int KIND>
__global__ __launch_bounds__(1024) void summary_refine_kernel(const double *__restrict__ rows, const double *__restrict__ a, uint32_t r1, const double *__restrict__ b,
                                                              uint32_t n_dims, const double *__restrict__ metric, double p, const double *__restrict__ sb,
                                                              const unsigned long long *__restrict__ smax_bits, uint32_t row0, uint32_t req_len,
                                                              uint32_t max_neighbours, double gamma, double *__restrict__ out_stats, uint32_t *__restrict__ out_n,
                                                              uint32_t *__restrict__ out_idx, double *__restrict__ out_dist, double *__restrict__ out_z,
                                                              uint32_t *__restrict__ rc, uint32_t *__restrict__ n_failed, SummaryLists L,
                                                              const double *__restrict__ na) {
  // (na: the reference rows `a` come as they are, still to be divided by their norms na -- the exact chain divides as it goes; nullptr: they are what they are)
  __shared__ uint32_t s_nb_i[kRefNb], s_med_i[kRefMed], s_mad_i[kRefMad];
  __shared__ double s_nb_d[kRefNb], s_med_d[kRefMed], s_mad_d[kRefMad];
  __shared__ double s_b[128], s_m[128];
  __shared__ double s_nb_x[kRefNb];  // the band's approximate values (what the sums of the mean and the standard deviation hold of them)
  __shared__ uint32_t s_cnt[8];  // [0] neighbours' band, [1] median's band, [2] below it, [3] MAD's band, [4] inside it, [5] failed
  __shared__ double s_val[2];    // the exact median, the exact MAD
  __shared__ double s_corr[16][2];
  const uint32_t jl = blockIdx.x, j = row0 + jl;
  if (rc[(uint64_t)jl * kRowCountsWords + kRowCountsFail]) return;  // (already the fall-back's)
  const double *row = rows + (uint64_t)jl * r1;
  const double *brow = b + (uint64_t)jl * n_dims;
  const bool staged = n_dims <= 128;  // (beyond: the query row and the metric from memory -- L2 hits: every thread reads the same few KB)
  for (uint32_t c = threadIdx.x; staged && c < n_dims; c += 1024) {
    s_b[c] = brow[c];
    s_m[c] = metric[c];
  }
  const double *qb = staged ? s_b : brow, *qm = staged ? s_m : metric;
  if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const double mean_a = out_stats[(uint64_t)j * 4 + 0], sd_a = out_stats[(uint64_t)j * 4 + 1], med_a = out_stats[(uint64_t)j * 4 + 2], mad_a = out_stats[(uint64_t)j * 4 + 3];
  const uint32_t eff_a = out_n[j];
  const double S = sb[jl] + __longlong_as_double((long long)*smax_bits);
  // |u~ - u| <= G for every pair of this row, u = d^2 for the euclidean form and the value itself for the cosine form (d^2 / 2)
  const double G = (KIND == KPOP_EUCLIDEAN ? 1.0 : 0.5) * gamma * S;
  auto sq = [](double x) -> double { return KIND == KPOP_EUCLIDEAN ? x * x : x; };
  // how far an approximate value near x can be from its exact value, in the value's own units (to size the MAD's bands: the
  // certificates below do not depend on it being right)
  auto err_at = [&](double x) -> double {
    if (KIND != KPOP_EUCLIDEAN) return G;
    return x * x >= 4.0 * G ? 1.25 * G / x : 2.0 * sqrt(G);
  };
  bool ok = eff_a >= 1 && eff_a <= max_neighbours && req_len <= max_neighbours && req_len >= 1 && r1 >= 2;
  // the approximate req_len-th smallest (the list the summary wrote is ascending)
  const double t_a = ok ? out_dist[(uint64_t)j * max_neighbours + (min(req_len, eff_a) - 1u)] : 0.0;
  // a neighbour (or a tie of the last) has u~ <= sq(t_a) + 2.5 G.  The band also takes everything so near the query that the
  // cancellation shows in the VALUE (u~ <= 1e5 G: a copy of the query comes out at 1e-7 instead of 0): their exact values
  // replace the approximate ones in the sums of the mean and the standard deviation as well
  const double u_nb = fmax(sq(t_a) + 2.5 * G, 1e5 * G);
  const double u_med_lo = sq(med_a) - 2.5 * G, u_med_hi = sq(med_a) + 2.5 * G;  // below / above: certainly below / above the exact median
  // the MAD: |d - median| < MAD inside (E1, E2) = median -+ MAD.  Brackets for the two edges, W either side:
  const double E2 = med_a + mad_a, E1 = med_a - mad_a;
  const double W = 3.0 * (err_at(med_a) + fmax(err_at(E2), E1 > 0.0 ? err_at(E1) : 0.0)) + 1e-15 * E2;
  const double e2lo = fmax(E2 - W, 0.0), e2hi = E2 + W, e1lo = E1 - W, e1hi = E1 + W;
  const double u_e2lo = sq(e2lo) - 1.25 * G, u_e2hi = sq(e2hi) + 1.25 * G;  // u~ below / above these: certainly below / above the upper edge
  const bool low_edge = e1hi > 0.0;                                        // (no lower edge: the MAD is at least the median)
  const double u_e1hi = low_edge ? sq(e1hi) + 1.25 * G : -1.0, u_e1lo = e1lo > 0.0 ? sq(e1lo) - 1.25 * G : -1.0;
  uint32_t n_below = 0, n_inside = 0;
  auto classify = [&](uint32_t i, double x, bool for_nb, bool for_med, bool for_mad) {
    const double u = sq(x);
    if (for_nb && u <= u_nb) {
      const uint32_t at = atomicAdd(&s_cnt[0], 1u);
      if (at < kRefNb) {
        s_nb_i[at] = i;
        s_nb_x[at] = x;
      }
    }
    if (for_med) {
      if (u < u_med_lo) ++n_below;
      else if (u <= u_med_hi) {
        const uint32_t at = atomicAdd(&s_cnt[1], 1u);
        if (at < kRefMed) s_med_i[at] = i;
      }
    }
    if (for_mad) {
      const bool in_sure = u < u_e2lo && (!low_edge || u > u_e1hi);
      const bool out_sure = u > u_e2hi || (e1lo > 0.0 && u < u_e1lo);
      if (in_sure) ++n_inside;
      else if (!out_sure) {
        const uint32_t at = atomicAdd(&s_cnt[3], 1u);
        if (at < kRefMad) s_mad_i[at] = i;
      }
    }
  };
  // The summary's one pass left lists behind (summary_large.hip: the row's values inside the median's bracket or in the MAD's
  // bands with their columns, those at or below the neighbours' threshold, and counts of everything else).  Where this row's
  // narrow bands lie inside those wide ones, everything here is read off the lists -- a few per cent of the row -- and the row
  // itself is not read again; otherwise it is scanned.
  bool from_lists = false;
  uint32_t base_below = 0, base_inside = 0;
  if (L.info != nullptr) {
    const RowCounts C = L.cnt[jl];
    const FusedThr T = L.thr[jl];
    from_lists = ok && !C.fail && C.n_cand <= L.cap && C.n_nb <= kNbCap;
    // neighbours: every value with u~ <= u_nb is at or below the threshold (values are not negative: sq is monotone)
    from_lists = from_lists && u_nb <= sq(T.cut);
    // the median: its band strictly inside the bracket; below the band = below or at the bracket's lower end + listed ones
    from_lists = from_lists && T.lo >= 0.0 && sq(T.lo) < u_med_lo && u_med_hi < sq(T.hi);
    base_below = C.lt_lo + C.eq_lo;
    // the MAD: the inner region certainly inside, beyond the bands certainly outside
    const bool has_inner = T.Uin > T.Lin;
    if (has_inner) from_lists = from_lists && sq(T.Uin) <= u_e2lo && (!low_edge || (T.Lin >= 0.0 && sq(T.Lin) >= u_e1hi));
    from_lists = from_lists && sq(T.Uhi) >= u_e2hi && (T.Llo <= 0.0 || (e1lo > 0.0 && sq(T.Llo) <= u_e1lo));
    base_inside = C.m_eqlo;
    if (from_lists) {
      const double *cv = L.cand + (uint64_t)jl * L.cap;
      const uint32_t *ci = L.cand_i + (uint64_t)jl * L.cap;
      for (uint32_t e = threadIdx.x; e < C.n_cand; e += 1024) {
        const double x = cv[e];
        const bool medc = x > T.lo && x < T.hi, in = x > T.Lin && x < T.Uin;
        classify(ci[e], x, false, medc, !in);  // (listed and in the inner region: counted there already)
      }
      const double *nv = L.nb_d + (uint64_t)jl * kNbCap;
      const uint32_t *ni = L.nb_idx + (uint64_t)jl * kNbCap;
      for (uint32_t e = threadIdx.x; e < C.n_nb; e += 1024) classify(ni[e], nv[e], true, false, false);
    }
  }
  if (!from_lists && rows == nullptr) {  // (uniform) the one-kernel path wrote no distance rows to scan: the fall-back's
#ifdef KPOP_REFINE_WHY
    if (threadIdx.x == 0 && L.info != nullptr) {
      const FusedThr T = L.thr[jl];
      printf("row %u not from lists: ok %d u_nb %g sq(cut) %g | lo %g hi %g u_med_lo %g u_med_hi %g | Lin %g Uin %g Llo %g Uhi %g u_e2lo %g u_e2hi %g u_e1lo %g u_e1hi %g low_edge %d med_a %g mad_a %g G %g\n", j, (int)ok, u_nb,
             sq(T.cut), T.lo, T.hi, u_med_lo, u_med_hi, T.Lin, T.Uin, T.Llo, T.Uhi, u_e2lo, u_e2hi, u_e1lo, u_e1hi, (int)low_edge, med_a, mad_a, G);
    }
#endif
    if (threadIdx.x == 0) {
      rc[(uint64_t)jl * kRowCountsWords + kRowCountsFail] = 1u;
      atomicAdd(n_failed, 1u);
    }
    return;
  }
  if (!from_lists) {
    base_below = base_inside = 0;
    constexpr int U = 8;  // loads in flight a thread
    for (uint32_t base = 0; base < r1; base += 1024 * U) {
      double xs[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const uint32_t i = base + k * 1024 + threadIdx.x;
        xs[k] = row[min(i, r1 - 1u)];
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const uint32_t i = base + k * 1024 + threadIdx.x;
        if (i < r1) classify(i, xs[k], true, true, true);
      }
    }
  }
  if (threadIdx.x == 0) {
    atomicAdd(&s_cnt[2], base_below);
    atomicAdd(&s_cnt[4], base_inside);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    n_below += (uint32_t)__shfl_xor((int)n_below, o, 64);
    n_inside += (uint32_t)__shfl_xor((int)n_inside, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&s_cnt[2], n_below);
    atomicAdd(&s_cnt[4], n_inside);
  }
  __syncthreads();
  const uint32_t n_nb = s_cnt[0], n_med = s_cnt[1], n_lt = s_cnt[2], n_mad = s_cnt[3], n_in = s_cnt[4];
  const uint32_t r_med = r1 / 2;  // element n / 2 of the sorted row: the upper median (lib/Matrix.ml:664-670)
  ok = ok && n_nb <= kRefNb && n_med <= kRefMed && n_mad <= kRefMad && n_nb >= req_len && r_med >= n_lt && r_med - n_lt < n_med && r_med >= n_in && r_med - n_in < n_mad;
  auto give_up = [&]() {  // the fall-back's
    if (threadIdx.x == 0) {
      rc[(uint64_t)jl * kRowCountsWords + kRowCountsFail] = 1u;
      atomicAdd(n_failed, 1u);
    }
  };
  if (!ok) {  // (uniform)
#ifdef KPOP_REFINE_WHY
    if (threadIdx.x == 0) printf("row %u gives up: n_nb %u n_med %u n_lt %u n_mad %u n_in %u req_len %u r_med %u from_lists %d eff_a %u max_nb %u t_a %g med_a %g mad_a %g mean_a %g\n", j, n_nb, n_med, n_lt, n_mad, n_in, req_len, r_med, (int)from_lists, eff_a, max_neighbours, t_a, med_a, mad_a, mean_a);
#endif
    give_up();
    return;
  }
  // the exact distances of the neighbours' and the median's bands
  double c1 = 0.0, c2 = 0.0;  // what the band's exact values change in sum d and in sum (d - mean)^2
  const bool by_wave = !staged;  // (more than 128 dimensions: a wavefront two pairs, exact_pair_wave2)
  const int lane_ = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
  auto exact_of = [&](uint32_t i) -> double {  // thread a pair
    return na ? exact_pair_adiv<KIND>(a + (uint64_t)i * n_dims, na[i], qb, qm, n_dims, p) : exact_pair<KIND>(a + (uint64_t)i * n_dims, qb, qm, n_dims, p);
  };
  // a wavefront two pairs, e and e + 16 of a band's list (every lane gets both values; a second one past the list's end: the first again)
  auto exact_wave2_of = [&](const uint32_t *idx, uint32_t e, uint32_t n, double *d0, double *d1) {
    const uint32_t i0 = idx[e], i1 = idx[e + 16u < n ? e + 16u : e];
    exact_pair_wave2<KIND>(a + (uint64_t)i0 * n_dims, na ? na[i0] : 0.0, a + (uint64_t)i1 * n_dims, na ? na[i1] : 0.0, qb, qm, n_dims, p, lane_, d0, d1);
  };
  if (by_wave) {
    for (uint32_t e = (uint32_t)wv_; e < n_nb; e += 32) {
      double dx[2];
      exact_wave2_of(s_nb_i, e, n_nb, &dx[0], &dx[1]);
      if (lane_ == 0)
        for (uint32_t t = 0; t < 2u && e + 16u * t < n_nb; ++t) {
          const double x = s_nb_x[e + 16u * t];
          s_nb_d[e + 16u * t] = dx[t];
          c1 += dx[t] - x;
          c2 += (dx[t] - mean_a) * (dx[t] - mean_a) - (x - mean_a) * (x - mean_a);
        }
    }
  } else {
    for (uint32_t e = threadIdx.x; e < n_nb; e += 1024) {
      const double dx = exact_of(s_nb_i[e]), x = s_nb_x[e];
      s_nb_d[e] = dx;
      c1 += dx - x;
      c2 += (dx - mean_a) * (dx - mean_a) - (x - mean_a) * (x - mean_a);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    c1 += __shfl_xor(c1, o, 64);
    c2 += __shfl_xor(c2, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_corr[threadIdx.x >> 6][0] = c1;
    s_corr[threadIdx.x >> 6][1] = c2;
  }
  if (by_wave) {
    for (uint32_t e = (uint32_t)wv_; e < n_med; e += 32) {
      double dx[2];
      exact_wave2_of(s_med_i, e, n_med, &dx[0], &dx[1]);
      if (lane_ == 0)
        for (uint32_t t = 0; t < 2u && e + 16u * t < n_med; ++t) s_med_d[e + 16u * t] = dx[t];
    }
  } else {
    for (uint32_t e = threadIdx.x; e < n_med; e += 1024) s_med_d[e] = exact_of(s_med_i[e]);
  }
  __syncthreads();
  // the median: the band's element of rank r_med - n_lt (ranks by counting: the bands are small)
  for (uint32_t e = threadIdx.x; e < n_med; e += 1024) {
    const double de = s_med_d[e];
    const uint32_t ie = s_med_i[e];
    uint32_t rank = 0;
    for (uint32_t f = 0; f < n_med; ++f) rank += pair_less(s_med_d[f], s_med_i[f], de, ie) ? 1u : 0u;
    if (rank == r_med - n_lt) s_val[0] = de;
  }
  __syncthreads();
  const double median = s_val[0];
  double mean = mean_a, sd = sd_a;
  if (r1 > 1) {
    double t1 = 0.0, t2 = 0.0;
    for (int w = 0; w < 16; ++w) {
      t1 += s_corr[w][0];
      t2 += s_corr[w][1];
    }
    const double n = (double)r1;
    mean = mean_a + t1 / n;
    const double ss = fmax(0.0, sd_a * sd_a * (n - 1.0) + t2 - n * (mean - mean_a) * (mean - mean_a));
    sd = sqrt(ss / (n - 1.0));
  }
  // (the certificate: the values counted as below / above ARE below / above it -- they are G or more away in u)
  if (!(sq(median) >= u_med_lo + G && sq(median) <= u_med_hi - G)) {  // (uniform)
    give_up();
    return;
  }
  // the MAD: exact deviations of its band from the exact median, the element of rank r_med - n_in
  if (by_wave) {
    for (uint32_t e = (uint32_t)wv_; e < n_mad; e += 32) {
      double dx[2];
      exact_wave2_of(s_mad_i, e, n_mad, &dx[0], &dx[1]);
      if (lane_ == 0)
        for (uint32_t t = 0; t < 2u && e + 16u * t < n_mad; ++t) s_mad_d[e + 16u * t] = fabs(__dsub_rn(dx[t], median));
    }
  } else {
    for (uint32_t e = threadIdx.x; e < n_mad; e += 1024) s_mad_d[e] = fabs(__dsub_rn(exact_of(s_mad_i[e]), median));
  }
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < n_mad; e += 1024) {
    const double de = s_mad_d[e];
    const uint32_t ie = s_mad_i[e];
    uint32_t rank = 0;
    for (uint32_t f = 0; f < n_mad; ++f) rank += pair_less(s_mad_d[f], s_mad_i[f], de, ie) ? 1u : 0u;
    if (rank == r_med - n_in) s_val[1] = de;
  }
  // the neighbours: the band sorted by (distance, column) -- every element to its rank --, cut after the req_len-th with its ties
  // (lib/Matrix.ml:641-650: the smallest groups of equal distances until req_len entries are reached)
  __syncthreads();
  const double mad = s_val[1];
  {
    // the certificate: both exact edges lie inside their brackets, so what was counted inside (outside) is
    const double x2 = median + mad, x1 = median - mad;
    const bool good = x2 >= e2lo && x2 <= e2hi && (low_edge ? (x1 <= e1hi && (e1lo <= 0.0 || x1 >= e1lo)) : x1 <= 0.0 || x1 <= e1hi);
    if (!good) {  // (uniform)
#ifdef KPOP_REFINE_WHY
      if (threadIdx.x == 0) printf("row %u: MAD certificate: median %.17g mad %.17g x1 %.17g x2 %.17g e1lo %g e1hi %g e2lo %g e2hi %g low_edge %d n_mad %u n_in %u med_a %.17g mad_a %.17g\n", j, median, mad, x1, x2, e1lo, e1hi, e2lo, e2hi, (int)low_edge, n_mad, n_in, med_a, mad_a);
#endif
      give_up();
      return;
    }
  }
  uint32_t my_rank[(kRefNb + 1023) / 1024];
#pragma unroll
  for (uint32_t s = 0; s < (kRefNb + 1023) / 1024; ++s) {
    const uint32_t e = threadIdx.x + 1024u * s;
    my_rank[s] = 0xFFFFFFFFu;
    if (e < n_nb) {
      const double de = s_nb_d[e];
      const uint32_t ie = s_nb_i[e];
      uint32_t rank = 0;
      for (uint32_t f = 0; f < n_nb; ++f) rank += pair_less(s_nb_d[f], s_nb_i[f], de, ie) ? 1u : 0u;
      my_rank[s] = rank;
      if (rank == req_len - 1u) s_val[0] = de;  // (the median has been read: its room takes the cut)
    }
  }
  __syncthreads();
  const double cut = s_val[0];
  if (!(sq(cut) <= u_nb - G)) {  // (the certificate: no row outside the band is as near as the cut or ties with it)
#ifdef KPOP_REFINE_WHY
    if (threadIdx.x == 0) printf("row %u: neighbours' certificate: cut %.17g u_nb %g G %g n_nb %u t_a %g eff_a %u\n", j, cut, u_nb, G, n_nb, t_a, eff_a);
#endif
    give_up();
    return;
  }
  uint32_t n_eff = 0;
  for (uint32_t e = threadIdx.x; e < n_nb; e += 1024) n_eff += s_nb_d[e] <= cut ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n_eff += (uint32_t)__shfl_xor((int)n_eff, o, 64);
  if ((threadIdx.x & 63) == 0 && n_eff) atomicAdd(&s_cnt[6], n_eff);
  __syncthreads();
  const uint32_t eff = s_cnt[6];
  if (eff > max_neighbours) {  // (a tie group longer than the caller's lists: the fall-back's, and after it the host's long lists)
#ifdef KPOP_REFINE_WHY
    if (threadIdx.x == 0) printf("row %u: eff %u > max_neighbours\n", j, eff);
#endif
    if (threadIdx.x == 0) {
      rc[(uint64_t)jl * kRowCountsWords + kRowCountsFail] = 1u;
      atomicAdd(n_failed, 1u);
    }
    return;
  }
#pragma unroll
  for (uint32_t s = 0; s < (kRefNb + 1023) / 1024; ++s) {
    const uint32_t e = threadIdx.x + 1024u * s;
    if (e < n_nb && my_rank[s] < eff) {
      const double dq = s_nb_d[e];
      out_idx[(uint64_t)j * max_neighbours + my_rank[s]] = s_nb_i[e];
      out_dist[(uint64_t)j * max_neighbours + my_rank[s]] = dq;
      double zz = __dsub_rn(dq, mean) / sd;
      if (zz != zz) zz = __longlong_as_double((long long)0xFFF8000000000000ull);  // x86 invalid-operation NaN, see distance.hip
      out_z[(uint64_t)j * max_neighbours + my_rank[s]] = zz;
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

// ---- host side ------------------------------------------------------------------------------------------------------------
bool summary_mfma_applies(int kind, uint32_t r1, uint32_t n_dims, uint32_t keep_at_most, uint32_t max_neighbours) {
  const uint32_t req_len = keep_at_most ? keep_at_most : r1;
  return ctx().tune_summary_mfma != 0 && (kind == KPOP_EUCLIDEAN || kind == KPOP_COSINE) && n_dims >= 4 && n_dims < 32768 && r1 >= 65536 &&
         req_len <= max_neighbours && max_neighbours <= 2048;
}
// room for: the query rows times the metric, the two sets of norms, the largest of them, the fall-back's flags and its count
static uint64_t mfma_scratch_bytes(uint32_t q, uint32_t r1, uint32_t n_dims, uint32_t bm_rows) {
  return (((uint64_t)r1 * 8 + 255) & ~255ull) + (((uint64_t)bm_rows * n_dims * 8 + 255) & ~255ull) + (((uint64_t)r1 * 8 + 255) & ~255ull) + (((uint64_t)q * 8 + 255) & ~255ull) + 256 +
         (((uint64_t)q * kRowCountsWords * 4 + 255) & ~255ull) + 256;
}
uint64_t summary_mfma_scratch_bytes(uint32_t q, uint32_t r1, uint32_t n_dims) { return mfma_scratch_bytes(q, r1, n_dims, q); }
struct MfmaScratch {
  double *bm, *sa, *sb;
  double *ia;  // the reciprocals of the reference rows' norms (the summary on reference rows that are still to be divided by them)
  unsigned long long *smax;
  uint32_t *rc, *n_failed;
};
static MfmaScratch carve_mfma(void *scratch, uint32_t q, uint32_t r1, uint32_t n_dims, uint32_t bm_rows = 0) {  // (bm_rows: rows of the copy times the metric, q unless said)
  char *p = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(scratch) + 255) & ~(uintptr_t)255);
  MfmaScratch M;
  M.bm = reinterpret_cast<double *>(p);
  p += ((uint64_t)(bm_rows ? bm_rows : q) * n_dims * 8 + 255) & ~255ull;
  M.sa = reinterpret_cast<double *>(p);
  p += ((uint64_t)r1 * 8 + 255) & ~255ull;
  M.ia = reinterpret_cast<double *>(p);
  p += ((uint64_t)r1 * 8 + 255) & ~255ull;
  M.sb = reinterpret_cast<double *>(p);
  p += ((uint64_t)q * 8 + 255) & ~255ull;
  M.smax = reinterpret_cast<unsigned long long *>(p);
  p += 256;
  M.rc = reinterpret_cast<uint32_t *>(p);
  p += ((uint64_t)q * kRowCountsWords * 4 + 255) & ~255ull;
  M.n_failed = reinterpret_cast<uint32_t *>(p);
  return M;
}

// the tiled contraction's 74 KB of dynamic LDS, allowed once a device slot
static int distance_gemm_lds_attr() {
  static PerSlotOnce once;
  if (!once()) {
    KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&distance_gemm_mfma_kernel<KPOP_EUCLIDEAN, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDgLds));
    KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&distance_gemm_mfma_kernel<KPOP_COSINE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDgLds));
    KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&distance_gemm_mfma_kernel<KPOP_EUCLIDEAN, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDgLds));
    KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&distance_gemm_mfma_kernel<KPOP_COSINE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDgLds));
    once() = true;
  }
  return 0;
}

// the reference set's norms (once per call: they do not depend on the chunk of query rows)
// na, s_raw (both or neither): the reference rows come as they are, still to be divided by their norms na; s_raw their sums of squares as
// they are (the norms' pass has them): sa = s_raw / na^2, ia = 1 / na -- no pass over the reference set here at all
__global__ __launch_bounds__(256) void reference_scales_kernel(const double *__restrict__ s_raw, const double *__restrict__ n, uint32_t rows, double *__restrict__ sa,
                                                               double *__restrict__ ia, unsigned long long *__restrict__ smax) {
  __shared__ unsigned long long s_max;
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
  unsigned long long mine = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) {
    const double r = 1.0 / n[i], v = s_raw[i] * r * r;
    ia[i] = r;
    sa[i] = v;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);  // (not negative: bit patterns order as the values do)
    mine = mine > bits ? mine : bits;
  }
  atomicMax(&s_max, mine);
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(smax, s_max);
}
int launch_mfma_reference_norms(const double *a, uint32_t r1, uint32_t n_dims, const double *metric, void *scratch, uint32_t q_room, hipStream_t st,
                                const double *na, const double *s_raw) {
  const MfmaScratch M = carve_mfma(scratch, q_room, r1, n_dims);
  KPOP_HIP(hipMemsetAsync(M.smax, 0, 256, st));
  if (na && s_raw)
    reference_scales_kernel<<<dim3(std::min(div_up(r1, 256u), 1024u)), dim3(256), 0, st>>>(s_raw, na, r1, M.sa, M.ia, M.smax);
  else
    row_sumsq_kernel<<<dim3(std::min(div_up(r1, 16), 4096u)), dim3(256), 0, st>>>(a, r1, n_dims, metric, M.sa, nullptr, M.smax);
  KPOP_LAUNCH_CHECK();
  return 0;
}

// ... into a second scratch of the same shape (a second chain of kernels of the same call: distance.hip, the two lanes of the summary)
int launch_mfma_copy_reference_norms(const void *from, void *to, uint32_t r1, uint32_t n_dims, uint32_t q_room, hipStream_t st) {
  const MfmaScratch A = carve_mfma(const_cast<void *>(from), q_room, r1, n_dims), B = carve_mfma(to, q_room, r1, n_dims);
  KPOP_HIP(hipMemcpyAsync(B.sa, A.sa, (uint64_t)r1 * 8, hipMemcpyDeviceToDevice, st));
  KPOP_HIP(hipMemcpyAsync(B.ia, A.ia, (uint64_t)r1 * 8, hipMemcpyDeviceToDevice, st));  // (filled or not: copied either way)
  KPOP_HIP(hipMemcpyAsync(B.smax, A.smax, 256, hipMemcpyDeviceToDevice, st));
  return 0;
}

template <int KIND>
static int launch_rows_mfma(const double *a, uint32_t r1, const double *b, uint32_t q, uint32_t n_dims, const double *metric, double *rows, const MfmaScratch &M,
                            hipStream_t st, bool a_raw) {
  const double *ia = a_raw ? M.ia : nullptr;  // (a_raw: the reference rows still to be divided by their norms, launch_mfma_reference_norms)
  row_sumsq_kernel<<<dim3(std::min(div_up(q, 16), 4096u)), dim3(256), 0, st>>>(b, q, n_dims, metric, M.sb, M.bm, nullptr);
  KPOP_LAUNCH_CHECK();
  if (n_dims > 128) {  // the tiled contraction (any number of dimensions): 128 query rows x 128 reference rows a block
    const uint32_t tiles_m = div_up(q, (uint32_t)kDT), tiles_n = div_up(r1, (uint32_t)kDT);
    KPOP_TRY(distance_gemm_lds_attr());
    distance_gemm_mfma_kernel<KIND, false><<<dim3(tiles_m * tiles_n), dim3(256), kDgLds, st>>>(a, r1, M.bm, b, q, n_dims, metric, 2.0, M.sa, M.sb, rows, tiles_m, tiles_n,
                                                                                                tiles_m <= 16 ? 1 : 0, 0.0, a, nullptr, nullptr, ia, nullptr);
    KPOP_LAUNCH_CHECK();
    return 0;
  }
  // every block resident at once (three a CU for up to 64 dimensions, two beyond), each an equal run of tiles
  const uint32_t n_tiles = div_up(r1, 16);
  const bool small = n_dims <= 64;
  const uint32_t rows_per_block = small ? 128u : 256u, ny = div_up(q, rows_per_block);
  const uint32_t resident = (uint32_t)ctx().n_cus * (small ? 3u : 2u);
  const uint32_t gx = std::max(1u, std::min(n_tiles, std::max(1u, resident / ny)));
  const uint32_t tpb = div_up(n_tiles, gx);
  const dim3 grid(div_up(n_tiles, tpb), ny);
  if (small)
    distance_rows_mfma_kernel<KIND, 16, 2><<<grid, dim3(256), 0, st>>>(a, r1, M.bm, q, n_dims, M.sa, M.sb, rows, tpb, ia);
  else
    distance_rows_mfma_kernel<KIND, 32, 4><<<grid, dim3(256), 0, st>>>(a, r1, M.bm, q, n_dims, M.sa, M.sb, rows, tpb, ia);
  KPOP_LAUNCH_CHECK();
  return 0;
}

// kpop_dev_distance_rowwise on the matrix cores (euclidean / cosine, operands already divided by their norms where the caller asked):
// out[j][i] for every pair, the pairs near enough for cancellation to show recomputed with the reference's chain (GUARD above)
bool distance_mfma_applies(int kind, uint32_t r1, uint32_t r2, uint32_t n_dims) {
  return ctx().tune_distance_mfma != 0 && (kind == KPOP_EUCLIDEAN || kind == KPOP_COSINE) && n_dims >= 16 && n_dims < 32768 && r1 >= 64 && r2 >= 64 &&
         (uint64_t)r1 * r2 * n_dims >= (1ull << 32);
}
double distance_mfma_tau(uint32_t n_dims) { return std::min(0.5, std::max(1.0 / 16.0, (double)(n_dims + 3) * 1.11e-4)); }
// s[i] = s_raw[i] / n[i]^2 (the sum of squares of the row divided by its norm), inv[i] = 1 / n[i]
__global__ __launch_bounds__(256) void norm_scales_kernel(double *__restrict__ s, const double *__restrict__ s_raw, const double *__restrict__ n, uint32_t rows,
                                                          double *__restrict__ inv) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < rows) {
    const double r = 1.0 / n[i];
    inv[i] = r;
    s[i] = s_raw[i] * r * r;
  }
}
// scaled = x m, nothing else (the copy of the smaller operand the panels of its side are loaded from)
__global__ __launch_bounds__(256) void scale_rows_kernel(const double *__restrict__ x, uint64_t n, uint32_t n_dims, const double *__restrict__ metric, double *__restrict__ scaled) {
  for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (uint64_t)gridDim.x * 256) scaled[e] = x[e] * metric[e % n_dims];
}

// a, b: the operands AS THEY ARE; n1, n2: their rows' norms when they are to be divided by them first (lib/Matrix.ml:201-202), nullptr otherwise;
// s1, s2: with the norms, the rows' sums of squares (the norms' pass has them: no pass of its own over the operands)
int launch_distance_rowwise_mfma(int kind, const double *a, uint32_t r1, const double *b, uint32_t r2, uint32_t n_dims, const double *metric, double p, double *out,
                                 hipStream_t st, const double *n1, const double *n2, const double *s1, const double *s2) {
  void *ws = nullptr;
  const uint32_t bm_rows = std::min(r1, r2);
  const uint64_t m_bytes = (mfma_scratch_bytes(r2, r1, n_dims, bm_rows) + 511) & ~255ull, inv_bytes = (((uint64_t)r1 + r2) * 8 + 511) & ~255ull;
  KPOP_TRY(ctx().ws_for(st).ensure(m_bytes + inv_bytes + 512, &ws));
  const MfmaScratch M = carve_mfma(ws, r2, r1, n_dims, bm_rows);
  double *ia = reinterpret_cast<double *>(reinterpret_cast<char *>(ws) + m_bytes), *ib = ia + r1;
  // the metric goes onto the SMALLER operand (its copy times the metric is what the panels of that side are loaded from: 21 MB for 1,636
  // classes where the copy of 100,000 samples was 1.3 GB written and read back -- half a millisecond of 11)
  const bool scale_a = r1 <= r2;
  if (n1 && s1 && s2) {
    const uint64_t n_small = (uint64_t)(scale_a ? r1 : r2) * n_dims;
    scale_rows_kernel<<<dim3((uint32_t)std::min<uint64_t>(div_up(n_small, (uint64_t)256), 8192)), dim3(256), 0, st>>>(scale_a ? a : b, n_small, n_dims, metric, M.bm);
    KPOP_LAUNCH_CHECK();
    norm_scales_kernel<<<dim3(div_up(r1, 256u)), dim3(256), 0, st>>>(M.sa, s1, n1, r1, ia);
    KPOP_LAUNCH_CHECK();
    norm_scales_kernel<<<dim3(div_up(r2, 256u)), dim3(256), 0, st>>>(M.sb, s2, n2, r2, ib);
    KPOP_LAUNCH_CHECK();
  } else {
    row_sumsq_kernel<<<dim3(std::min(div_up(r1, 16), 4096u)), dim3(256), 0, st>>>(a, r1, n_dims, metric, M.sa, scale_a ? M.bm : nullptr, nullptr);
    KPOP_LAUNCH_CHECK();
    row_sumsq_kernel<<<dim3(std::min(div_up(r2, 16), 4096u)), dim3(256), 0, st>>>(b, r2, n_dims, metric, M.sb, scale_a ? nullptr : M.bm, nullptr);
    KPOP_LAUNCH_CHECK();
    if (n1) {
      norm_scales_kernel<<<dim3(div_up(r1, 256u)), dim3(256), 0, st>>>(M.sa, M.sa, n1, r1, ia);
      KPOP_LAUNCH_CHECK();
      norm_scales_kernel<<<dim3(div_up(r2, 256u)), dim3(256), 0, st>>>(M.sb, M.sb, n2, r2, ib);
      KPOP_LAUNCH_CHECK();
    } else {
      ia = ib = nullptr;
    }
  }
  const double *pa = scale_a ? M.bm : a, *pb = scale_a ? b : M.bm;  // the panels' sources
  const uint32_t tiles_m = div_up(r2, (uint32_t)kDT), tiles_n = div_up(r1, (uint32_t)kDT);
  if ((uint64_t)tiles_m * tiles_n >= (1ull << 31)) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "distance_rowwise: %u x %u tiles", tiles_m, tiles_n);
  const double tau = distance_mfma_tau(n_dims);
  const int m_fast = tiles_m <= tiles_n ? 1 : 0;  // (the shorter side fastest: its panel stays in the L2s)
  KPOP_TRY(distance_gemm_lds_attr());
  if (kind == KPOP_EUCLIDEAN)
    distance_gemm_mfma_kernel<KPOP_EUCLIDEAN, true><<<dim3(tiles_m * tiles_n), dim3(256), kDgLds, st>>>(pa, r1, pb, b, r2, n_dims, metric, p, M.sa, M.sb, out, tiles_m, tiles_n, m_fast, tau, a, n1, n2, ia, ib);
  else
    distance_gemm_mfma_kernel<KPOP_COSINE, true><<<dim3(tiles_m * tiles_n), dim3(256), kDgLds, st>>>(pa, r1, pb, b, r2, n_dims, metric, p, M.sa, M.sb, out, tiles_m, tiles_n, m_fast, tau, a, n1, n2, ia, ib);
  KPOP_LAUNCH_CHECK();
  return 0;
}

// the one-kernel path's pass (declared in summary_large.hip, which owns its scratch): q query rows (their fragments and norms in the
// matrix-core scratch: launch_mfma_query_prep) against all r1 reference rows, a block a (stripe, 128 query rows)
int launch_select_mfma(int kind, const double *a, uint32_t r1, uint32_t q, uint32_t n_dims, const void *mscratch, uint32_t q_room, const FusedThr *thr, double *seg,
                       uint32_t *seg_i, StripeRec *rec, double *part, RowCounts *cnt, uint32_t *nb_idx, double *nb_d, uint32_t n_stripes, hipStream_t st) {
  const MfmaScratch M = carve_mfma(const_cast<void *>(mscratch), q_room, r1, n_dims);
  const dim3 grid(n_stripes, div_up(q, 128u));
#define KPOP_SEL(K, KSV) summary_select_mfma_kernel<K, KSV><<<grid, dim3(256), 0, st>>>(a, r1, M.bm, q, n_dims, M.sa, M.sb, thr, seg, seg_i, rec, part, cnt, nb_idx, nb_d, n_stripes)
  if (kind == KPOP_EUCLIDEAN) {
    if (n_dims <= 64) KPOP_SEL(KPOP_EUCLIDEAN, 16); else KPOP_SEL(KPOP_EUCLIDEAN, 32);
  } else {
    if (n_dims <= 64) KPOP_SEL(KPOP_COSINE, 16); else KPOP_SEL(KPOP_COSINE, 32);
  }
#undef KPOP_SEL
  KPOP_LAUNCH_CHECK();
  return 0;
}
// the chunk's query rows times the metric and their norms into the scratch, the fall-back's flags cleared
int launch_mfma_query_prep(const double *b, uint32_t q, uint32_t r1, uint32_t n_dims, const double *metric, void *scratch, uint32_t q_room, hipStream_t st) {
  const MfmaScratch M = carve_mfma(scratch, q_room, r1, n_dims);
  KPOP_HIP(hipMemsetAsync(M.rc, 0, (uint64_t)q * kRowCountsWords * 4, st));
  KPOP_HIP(hipMemsetAsync(M.n_failed, 0, 256, st));
  row_sumsq_kernel<<<dim3(std::min(div_up(q, 16), 4096u)), dim3(256), 0, st>>>(b, q, n_dims, metric, M.sb, M.bm, nullptr);
  KPOP_LAUNCH_CHECK();
  return 0;
}
// approximate distance rows of the prepared query rows against ANOTHER set of rows (the sample of the reference set: `as`, its norms `sas`)
int launch_rows_mfma_against(int kind, const double *as, const double *sas, uint32_t s, uint32_t q, uint32_t n_dims, double *rows, void *scratch, uint32_t q_room,
                             uint32_t r1, hipStream_t st, const double *ias) {
  // (ias: the sample's rows come as they are, still to be divided by their norms -- sas the sums of squares of the rows as they will be, ias the
  // norms' reciprocals; nullptr: the rows are what they are)
  MfmaScratch M = carve_mfma(scratch, q_room, r1, n_dims);
  if (n_dims > 128) {  // the tiled contraction
    const uint32_t tiles_m = div_up(q, (uint32_t)kDT), tiles_n = div_up(s, (uint32_t)kDT);
    KPOP_TRY(distance_gemm_lds_attr());
    if (kind == KPOP_EUCLIDEAN)
      distance_gemm_mfma_kernel<KPOP_EUCLIDEAN, false><<<dim3(tiles_m * tiles_n), dim3(256), kDgLds, st>>>(as, s, M.bm, nullptr, q, n_dims, nullptr, 2.0, sas, M.sb, rows, tiles_m, tiles_n,
                                                                                                          tiles_m <= 16 ? 1 : 0, 0.0, nullptr, nullptr, nullptr, ias, nullptr);
    else
      distance_gemm_mfma_kernel<KPOP_COSINE, false><<<dim3(tiles_m * tiles_n), dim3(256), kDgLds, st>>>(as, s, M.bm, nullptr, q, n_dims, nullptr, 2.0, sas, M.sb, rows, tiles_m, tiles_n,
                                                                                                       tiles_m <= 16 ? 1 : 0, 0.0, nullptr, nullptr, nullptr, ias, nullptr);
    KPOP_LAUNCH_CHECK();
    return 0;
  }
  const uint32_t n_tiles = div_up(s, 16);
  const bool small = n_dims <= 64;
  const uint32_t rows_per_block = small ? 128u : 256u, ny = div_up(q, rows_per_block);
  const uint32_t resident = (uint32_t)ctx().n_cus * (small ? 3u : 2u);
  const uint32_t gx = std::max(1u, std::min(n_tiles, std::max(1u, resident / ny)));
  const uint32_t tpb = div_up(n_tiles, gx);
  const dim3 grid(div_up(n_tiles, tpb), ny);
#define KPOP_ROWS(K) \
  do { \
    if (small) distance_rows_mfma_kernel<K, 16, 2><<<grid, dim3(256), 0, st>>>(as, s, M.bm, q, n_dims, sas, M.sb, rows, tpb, ias); \
    else distance_rows_mfma_kernel<K, 32, 4><<<grid, dim3(256), 0, st>>>(as, s, M.bm, q, n_dims, sas, M.sb, rows, tpb, ias); \
  } while (0)
  if (kind == KPOP_EUCLIDEAN) KPOP_ROWS(KPOP_EUCLIDEAN); else KPOP_ROWS(KPOP_COSINE);
#undef KPOP_ROWS
  KPOP_LAUNCH_CHECK();
  return 0;
}
// the reference set's sums of squares (and, when it is taken as it is, its norms' reciprocals) of the SAMPLE of its rows -- rows
// floor(i r1 / s), launch_sample_gather's -- out of the scratch launch_mfma_reference_norms filled
__global__ __launch_bounds__(256) void gather_sample_scalars_kernel(const double *__restrict__ sa, const double *__restrict__ ia, uint32_t r1, uint32_t s,
                                                                    double *__restrict__ sas, double *__restrict__ ias) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= s) return;
  const uint64_t src = ((uint64_t)i * r1) / s;
  sas[i] = sa[src];
  if (ias) ias[i] = ia[src];
}
int launch_mfma_sample_scalars(const void *scratch, uint32_t q_room, uint32_t r1, uint32_t n_dims, uint32_t s, double *sas, double *ias, hipStream_t st) {
  const MfmaScratch M = carve_mfma(const_cast<void *>(scratch), q_room, r1, n_dims);
  gather_sample_scalars_kernel<<<dim3(div_up(s, 256u)), dim3(256), 0, st>>>(M.sa, M.ia, r1, s, sas, ias);
  KPOP_LAUNCH_CHECK();
  return 0;
}
int launch_row_sumsq(const double *x, uint32_t rows, uint32_t n_dims, const double *metric, double *out, hipStream_t st) {
  row_sumsq_kernel<<<dim3(std::min(div_up(rows, 16), 4096u)), dim3(256), 0, st>>>(x, rows, n_dims, metric, out, nullptr, nullptr);
  KPOP_LAUNCH_CHECK();
  return 0;
}

// the chunk's approximate distance rows (into `rows`), and the flags of its fall-back cleared
int launch_distance_rows_mfma(int kind, const double *a, uint32_t r1, const double *b, uint32_t q, uint32_t n_dims, const double *metric, double *rows,
                              void *scratch, uint32_t q_room, hipStream_t st, bool a_raw) {
  const MfmaScratch M = carve_mfma(scratch, q_room, r1, n_dims);
  KPOP_HIP(hipMemsetAsync(M.rc, 0, (uint64_t)q * kRowCountsWords * 4, st));
  KPOP_HIP(hipMemsetAsync(M.n_failed, 0, 256, st));
  return kind == KPOP_EUCLIDEAN ? launch_rows_mfma<KPOP_EUCLIDEAN>(a, r1, b, q, n_dims, metric, rows, M, st, a_raw)
                                : launch_rows_mfma<KPOP_COSINE>(a, r1, b, q, n_dims, metric, rows, M, st, a_raw);
}

// the refinement; *gate = the device word that counts the rows left to the fall-back, *row_counts = their flags (RowCounts)
int launch_summary_refine(int kind, const double *rows, const double *a, uint32_t r1, const double *b, uint32_t q, uint32_t n_dims, const double *metric,
                          double p, uint32_t row0, uint32_t keep_at_most, uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx,
                          double *out_dist, double *out_z, void *scratch, uint32_t q_room, hipStream_t st, const SummaryLists &lists, const uint32_t **gate,
                          const void **row_counts, const double *na) {
  const MfmaScratch M = carve_mfma(scratch, q_room, r1, n_dims);
  const uint32_t req_len = keep_at_most ? keep_at_most : r1;
  // |u~ - u| <= gamma (|a|^2 + |b|^2): n_dims products and additions of the contraction and of the two norms at 2^-53 each, the
  // chain's own roundings, and a factor of ten on top
  const double gamma = 4e-15 * (double)std::max(n_dims, 16u);
  SummaryLists L = lists;
  if (!ctx().tune_summary_mfma_lists || !L.cand_i) L = SummaryLists{};
  if (kind == KPOP_EUCLIDEAN)
    summary_refine_kernel<KPOP_EUCLIDEAN><<<dim3(q), dim3(1024), 0, st>>>(rows, a, r1, b, n_dims, metric, p, M.sb, M.smax, row0, req_len, max_neighbours, gamma, out_stats,
                                                                          out_n, out_idx, out_dist, out_z, M.rc, M.n_failed, L, na);
  else
    summary_refine_kernel<KPOP_COSINE><<<dim3(q), dim3(1024), 0, st>>>(rows, a, r1, b, n_dims, metric, p, M.sb, M.smax, row0, req_len, max_neighbours, gamma, out_stats,
                                                                       out_n, out_idx, out_dist, out_z, M.rc, M.n_failed, L, na);
  KPOP_LAUNCH_CHECK();
  *gate = M.n_failed;
  *row_counts = M.rc;
  return 0;
}

}  // namespace kpop

// distance.hip -- norms, rowwise distances and the per-row summary.
//
//   row_norms_kernel         Base.get_normalizations   lib/Matrix.ml:42-76
//                            Space.compute_norm        lib/Space.ml:166-181
//   distance_rowwise_kernel  Base.get_distance_rowwise lib/Matrix.ml:191-266
//                            Space.Distance.compute    lib/Space.ml:182-205
//   distance_summary_kernel  summarize_rowwise / summarize_distance_matrix_row
//                                                      lib/Matrix.ml:691-766,632-690
//
// Numerics follow the reference operation for operation: every pair's sum runs
// over the dimensions in ascending order in ONE thread, diff*diff*m is
// evaluated left to right, nothing is fused, and a/n_i, b/n_j are the same
// IEEE divisions the reference performs per element (lib/Matrix.ml:247-249) --
// done once per row here instead of once per pair.
#include <math.h>

#include <algorithm>
#include <vector>

#include <string.h>

#include "common.h"
#include "summary_types.h"
#include "radix_sort.h"
#include "space_ops.h"
#include "wave_sort.h"

namespace kpop {

// ---------------------------------------------------------------------------
// norms + pre-normalised rows.  One thread per row walks the dimensions in
// order; 64-row x 32-dim tiles go through LDS so global traffic is coalesced.
// ---------------------------------------------------------------------------
constexpr int kNormRows = 64, kNormDims = 32;

// norms[i] = scale(sum_c m_c g(a_ic)), 0 -> 1 (lib/Matrix.ml:67); when `normalised` is non-null the
// block then re-reads its 64 rows (still in L2) and writes a_ic / n_i (adaptor_a/_b, lib/Matrix.ml:248)
template <int KIND>
__device__ __forceinline__ void row_norms_block(const double *__restrict__ m, uint32_t rows, uint32_t n_dims,
                                                const double *__restrict__ metric, double p,
                                                double *__restrict__ norms, double *__restrict__ normalised, uint32_t block,
                                                double *__restrict__ sumsq = nullptr) {
  // (sumsq: the sum itself, before the scale -- sum_c m_c a_ic^2 for the euclidean and the cosine form: what the matrix-core path wants of a row)
  __shared__ double tile[kNormRows][kNormDims + 1];
  __shared__ double s_metric[kNormDims];
  __shared__ double s_norm[kNormRows];
  const uint32_t row0 = block * kNormRows;
  double acc = 0.0;
  // (the next tile's loads fly while 64 of the block's threads walk this one: a tile at a time left the kernel at 2 TB/s on 1M x 64)
  constexpr int kPer = kNormRows * kNormDims / 256;
  double pre[kPer];
  auto fetch = [&](uint32_t c0) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const uint32_t e = threadIdx.x + 256u * u, i = e / kNormDims, c = e % kNormDims;
      // (addresses clamped into the matrix, what lies outside zeroed when the tile is stored: nothing here looks at a loaded value)
      pre[u] = m[(uint64_t)min(row0 + i, rows - 1u) * n_dims + min(c0 + c, n_dims - 1u)];
    }
  };
  fetch(0);
  for (uint32_t c0 = 0; c0 < n_dims; c0 += kNormDims) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const uint32_t e = threadIdx.x + 256u * u, i = e / kNormDims, c = e % kNormDims;
      tile[i][c] = (row0 + i < rows && c0 + c < n_dims) ? pre[u] : 0.0;
    }
    if (threadIdx.x < kNormDims) s_metric[threadIdx.x] = (c0 + threadIdx.x < n_dims) ? metric[c0 + threadIdx.x] : 0.0;
    __syncthreads();
    if (c0 + kNormDims < n_dims) fetch(c0 + kNormDims);
    if (threadIdx.x < kNormRows) {
      const uint32_t lim = min((uint32_t)kNormDims, n_dims - c0);
      for (uint32_t c = 0; c < lim; ++c) {
        double el = tile[threadIdx.x][c];
        // lib/Space.ml:169-178: acc +. (el *. el *. m_i)  |  acc +. ((|el| ** p) *. m_i)
        acc = __dadd_rn(acc, component<KIND>(el, s_metric[c], p));
      }
    }
  }
  if (threadIdx.x < kNormRows) {
    double nv = scale_distance<KIND>(acc, p);
    nv = (nv == 0.0) ? 1.0 : nv;  // lib/Matrix.ml:67
    s_norm[threadIdx.x] = nv;
    if (row0 + threadIdx.x < rows) {
      norms[row0 + threadIdx.x] = nv;
      if (sumsq) sumsq[row0 + threadIdx.x] = acc;
    }
  }
  if (!normalised) return;
  __syncthreads();
  const uint32_t nrows = min((uint32_t)kNormRows, rows - row0);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (uint32_t i = wv; i < nrows; i += 4) {  // one wave per row: coalesced, no integer division
    const double *src = m + (uint64_t)(row0 + i) * n_dims;
    double *dst = normalised + (uint64_t)(row0 + i) * n_dims;
    const double nv = s_norm[i];
    for (uint32_t c = lane; c < n_dims; c += 64) dst[c] = src[c] / nv;
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void row_norms_kernel(const double *__restrict__ m, uint32_t rows, uint32_t n_dims,
                                                        const double *__restrict__ metric, double p,
                                                        double *__restrict__ norms, double *__restrict__ normalised) {
  row_norms_block<KIND>(m, rows, n_dims, metric, p, norms, normalised, blockIdx.x);
}

// both operands of a distance in ONE launch (the first blocks take the first operand's rows): a launch of its own for 65 class
// vectors was 9 us of start-up in front of 12 us of work on the 100,000 rows
template <int KIND>
__global__ __launch_bounds__(256) void row_norms_pair_kernel(const double *__restrict__ m1, uint32_t r1, double *__restrict__ n1,
                                                             double *__restrict__ a_div, const double *__restrict__ m2, uint32_t r2,
                                                             double *__restrict__ n2, double *__restrict__ b_div, uint32_t n_dims,
                                                             const double *__restrict__ metric, double p, uint32_t blocks1,
                                                             double *__restrict__ s1 = nullptr, double *__restrict__ s2 = nullptr) {
  const bool first = blockIdx.x < blocks1;  // (uniform)
  row_norms_block<KIND>(first ? m1 : m2, first ? r1 : r2, n_dims, metric, p, first ? n1 : n2, first ? a_div : b_div, first ? blockIdx.x : blockIdx.x - blocks1,
                        first ? s1 : s2);
}

template <int KIND>
static int launch_row_norms_pair(const double *m1, uint32_t r1, double *n1, double *a_div, const double *m2, uint32_t r2, double *n2, double *b_div,
                                 uint32_t n_dims, const double *metric, double p, hipStream_t st, double *s1 = nullptr, double *s2 = nullptr) {
  const uint32_t blocks1 = m1 ? div_up(r1, kNormRows) : 0u, blocks2 = m2 ? div_up(r2, kNormRows) : 0u;
  if (blocks1 + blocks2 == 0) return 0;
  row_norms_pair_kernel<KIND><<<dim3(blocks1 + blocks2), dim3(256), 0, st>>>(m1, r1, n1, a_div, m2, r2, n2, b_div, n_dims, metric, p, blocks1, s1, s2);
  KPOP_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// rowwise distances.  a, b are the pre-normalised operands.  A block owns a tile
// of `w` m1-rows (columns of the result) x TJ m2-rows; each thread owns one
// 4(j) x 4(i) micro-tile and walks the dimensions in ascending order, 16 at a
// time through LDS ([dim][row], so a thread's 4 rows are one 32-byte read).
// The tile shape adapts to r1: n_cg = ceil(w/4) column groups, n_rg =
// floor(256/n_cg) row groups (<= 64), so r1 = 65 runs as 17 x 15 groups at 95 %
// lane use instead of paying for a second, empty 64-wide tile.
// ---------------------------------------------------------------------------
constexpr int kDC = 16, kMaxW = 128, kMaxTJ = 256;

// SLAB (spectral distances, n_dims in the millions and few rows): blockIdx.z owns dimensions [z*slab, (z+1)*slab) and
// writes its raw partial sums to slab z of `out` ([gridDim.z][r2][r1]); reduce_slabs_kernel adds the slabs in order and
// applies the scale.  SLAB = false is the hot path: one block walks all the dimensions, the sum is the reference's.
// MAXTJ: the most second-operand rows a block stages (TJ <= MAXTJ).  With 128 (one column tile of 61..127 columns: distances to a
// set of classes) and the rows' norms in LDS instead of 48 registers the kernel fits 168 VGPRs: three wavefronts a SIMD, not two.
template <int KIND, bool SLAB = false, int TY = 4, int MAXTJ = 256>
__global__ __launch_bounds__(256, MAXTJ == 128 ? 3 : 1) void distance_rowwise_kernel(const double *__restrict__ a, uint32_t w, uint32_t r1,
                                                               const double *__restrict__ b, uint32_t r2,
                                                               uint32_t n_dims, const double *__restrict__ metric,
                                                               double p, double *__restrict__ out, uint32_t n_cg,
                                                               uint32_t n_rg, uint32_t slab = 0, const double *__restrict__ na = nullptr,
                                                               const double *__restrict__ nb = nullptr, const uint32_t *__restrict__ gate = nullptr) {
  if (gate && *gate == 0) return;  // (a fallback that is launched whatever happens and runs only when something failed: summary_large_impl)
  __shared__ __attribute__((aligned(16))) double As[kDC][kMaxW + 2];
  __shared__ __attribute__((aligned(16))) double Bs[kDC][MAXTJ + 2];
  __shared__ double s_metric[kDC];
  __shared__ double s_na[kMaxW], s_nb[MAXTJ];  // the rows' norms (read as the rows are staged)
  const uint32_t TJ = TY * n_rg;
  const uint32_t i0 = blockIdx.x * w, j0 = blockIdx.y * TJ;
  const uint32_t i1 = min(r1, i0 + w);
  const uint32_t cg = threadIdx.x % n_cg, rg = threadIdx.x / n_cg;
  const bool worker = rg < n_rg;
  const uint32_t ti = cg * 4, tj = (worker ? rg : 0) * TY;
  double acc[TY][4];
#pragma unroll
  for (int y = 0; y < TY; ++y)
#pragma unroll
    for (int x = 0; x < 4; ++x) acc[y][x] = 0.0;
  // Staging is software-pipelined through registers: the global loads of chunk c+1 are issued before
  // the arithmetic of chunk c and land in LDS after it, so HBM/L2 latency hides behind the f64 work.
  constexpr int NA = kMaxW * kDC / 256, NB = MAXTJ * kDC / 256;
  double ra[NA], rb[NB];
  const uint32_t sc = threadIdx.x % kDC, rbase = threadIdx.x / kDC;  // 16 rows per sweep of the block
  const uint32_t d_begin = SLAB ? blockIdx.z * slab : 0u;
  const uint32_t d_end = SLAB ? min(n_dims, d_begin + slab) : n_dims;
  // na / nb: the norms of the rows (lib/Matrix.ml:247-249 divides every element by its row's norm before the difference).
  // The quotients are taken here, as the rows are staged, instead of being written out as a second copy of both operands
  // and read back: the same IEEE division of the same operands.
  const bool divide = na != nullptr;
  if (divide) {
    for (uint32_t i = threadIdx.x; i < kMaxW; i += 256) s_na[i] = na[min(i0 + i, r1 - 1)];
    for (uint32_t j = threadIdx.x; j < (uint32_t)MAXTJ; j += 256) s_nb[j] = nb ? nb[min(j0 + j, r2 - 1)] : 1.0;
  }
  auto prefetch = [&](uint32_t c0) {
    const bool cok = c0 + sc < d_end;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const uint32_t row = rbase + q * 16;
      ra[q] = (cok && row < w && i0 + row < i1) ? a[(uint64_t)(i0 + row) * n_dims + c0 + sc] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const uint32_t row = rbase + q * 16;
      rb[q] = (cok && row < TJ && j0 + row < r2) ? b[(uint64_t)(j0 + row) * n_dims + c0 + sc] : 0.0;
    }
  };
  prefetch(d_begin);
  for (uint32_t c0 = d_begin; c0 < d_end; c0 += kDC) {
    __syncthreads();  // the previous chunk's readers are done
    if (divide) {
#pragma unroll
      for (int q = 0; q < NA; ++q) As[sc][rbase + q * 16] = __ddiv_rn(ra[q], s_na[rbase + q * 16]);
#pragma unroll
      for (int q = 0; q < NB; ++q) Bs[sc][rbase + q * 16] = __ddiv_rn(rb[q], s_nb[rbase + q * 16]);
    } else {
#pragma unroll
      for (int q = 0; q < NA; ++q) As[sc][rbase + q * 16] = ra[q];
#pragma unroll
      for (int q = 0; q < NB; ++q) Bs[sc][rbase + q * 16] = rb[q];
    }
    if (threadIdx.x < kDC) s_metric[threadIdx.x] = (c0 + threadIdx.x < d_end) ? metric[c0 + threadIdx.x] : 0.0;
    __syncthreads();
    if (c0 + kDC < d_end) prefetch(c0 + kDC);
    const uint32_t lim = min((uint32_t)kDC, d_end - c0);
    if (worker) {
      for (uint32_t cc = 0; cc < lim; ++cc) {
        double av[4], bv[TY];
#pragma unroll
        for (int x = 0; x < 4; ++x) av[x] = As[cc][ti + x];
#pragma unroll
        for (int y = 0; y < TY; ++y) bv[y] = Bs[cc][tj + y];
        const double mc = s_metric[cc];
#pragma unroll
        for (int y = 0; y < TY; ++y)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            // lib/Space.ml:192-200: diff = a -. b ; acc +. (diff *. diff *. m)
            double diff = __dsub_rn(av[x], bv[y]);
            acc[y][x] = __dadd_rn(acc[y][x], component<KIND>(diff, mc, p));
          }
      }
    }
  }
  if (!worker) return;
#pragma unroll
  for (int y = 0; y < TY; ++y) {
    const uint32_t j = j0 + tj + y;
    if (j >= r2) continue;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const uint32_t i = i0 + ti + x;
      if (i < i1) {
        if (SLAB) out[((uint64_t)blockIdx.z * r2 + j) * r1 + i] = acc[y][x];
        else out[(uint64_t)j * r1 + i] = scale_distance<KIND>(acc[y][x], p);  // data.(j).@(i), lib/Matrix.ml:253
      }
    }
  }
}

// ---------------------------------------------------------------------------
// summary: one block per m2 row.  LDS: dist[NP] (column order), kd[NP]/ki[NP]
// (sorted by distance then column: the FloatIntMultimap of lib/Matrix.ml:631).
// ---------------------------------------------------------------------------
constexpr uint32_t kSummaryMaxR1 = 4096;

struct DistIdx {
  double d;
  uint32_t i;
};

__device__ __forceinline__ bool pair_less(double da, uint32_t ia, double db, uint32_t ib) {
  return (da < db) || (da == db && ia < ib);
}

// block-wide bitonic sort of (kd, ki) pairs, NP a power of two
__device__ void block_sort_pairs(double *kd, uint32_t *ki, uint32_t NP) {
  for (uint32_t s = 2; s <= NP; s <<= 1) {
    for (uint32_t t = s >> 1; t > 0; t >>= 1) {
      __syncthreads();
      for (uint32_t q = threadIdx.x; q < NP / 2; q += blockDim.x) {
        const uint32_t i = 2 * q - (q & (t - 1)), j = i + t;
        const bool asc = (i & s) == 0;
        const double di = kd[i], dj = kd[j];
        const uint32_t ii = ki[i], ij = ki[j];
        const bool gt = pair_less(dj, ij, di, ii);
        if (gt == asc) {
          kd[i] = dj; kd[j] = di;
          ki[i] = ij; ki[j] = ii;
        }
      }
    }
  }
  __syncthreads();
}

// the last phase of the network: a bitonic sequence into ascending order
__device__ void block_merge_keys(double *kd, uint32_t NP) {
  for (uint32_t t = NP >> 1; t > 0; t >>= 1) {
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < NP / 2; q += blockDim.x) {
      const uint32_t i = 2 * q - (q & (t - 1)), j = i + t;
      const double di = kd[i], dj = kd[j];
      if (dj < di) {
        kd[i] = dj; kd[j] = di;
      }
    }
  }
  __syncthreads();
}

// PRE = true: `a` is a ready r2 x r1 distance matrix (summarize_distance, lib/Matrix.ml:767-810) and row j is
// only loaded; PRE = false: distances of m2 row j to every m1 row are computed here (summarize_rowwise).
template <int KIND, bool PRE>
__global__ __launch_bounds__(1024) void distance_summary_kernel(
    const double *__restrict__ a, uint32_t r1, const double *__restrict__ b, uint32_t r2, uint32_t n_dims,
    const double *__restrict__ metric, double p, uint32_t NP, uint32_t req_len, uint32_t max_neighbours,
    double *__restrict__ out_stats, uint32_t *__restrict__ out_n, uint32_t *__restrict__ out_idx,
    double *__restrict__ out_dist, double *__restrict__ out_z) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *dist = reinterpret_cast<double *>(smem);     // [NP] column order
  double *kd = dist + NP;                               // [NP]
  uint32_t *ki = reinterpret_cast<uint32_t *>(kd + NP);  // [NP]
  __shared__ double s_stats[4];
  __shared__ uint32_t s_eff;
  const uint32_t j = blockIdx.x;
  const double *brow = PRE ? nullptr : b + (uint64_t)j * n_dims;
  const double inf = __longlong_as_double(0x7FF0000000000000ll);
  // distances of row j to every m1 row (lib/Matrix.ml:744-749), one thread per m1 row
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) {
    double d = inf;
    if (PRE) {
      if (i < r1) d = a[(uint64_t)j * r1 + i];
    } else if (i < r1) {
      const double *arow = a + (uint64_t)i * n_dims;
      double acc = 0.0;
      for (uint32_t c = 0; c < n_dims; ++c) {
        double diff = __dsub_rn(arow[c], brow[c]);
        acc = __dadd_rn(acc, component<KIND>(diff, metric[c], p));
      }
      d = scale_distance<KIND>(acc, p);
    }
    dist[i] = d;
    kd[i] = d;
    ki[i] = (i < r1) ? i : 0xFFFFFFFFu;
  }
  block_sort_pairs(kd, ki, NP);
  // lib/Matrix.ml:640-655: the multimap is walked in ascending order one distinct distance at a time and set_len *. dist
  // added up in that order.  The TERMS are worked out by all threads (a position that starts a group of equal distances
  // finds its end; the others contribute +0.0, exact), the chain that adds them stays one thread's, in order -- but over
  // terms that lie ready in LDS, eight loads ahead, instead of a walk whose every step waited for the one before
  // (4,000 columns: ~0.2 ms a row; 100,000 rows against 1,000 / 4,000 columns: 13 -> 5.8 / 282 -> 98 ms).
  if (threadIdx.x == 0) s_eff = r1;
  __syncthreads();
  for (uint32_t s = threadIdx.x; s < r1; s += blockDim.x) {
    const double dd = kd[s];
    const bool head = s == 0 || kd[s - 1] != dd;
    double term = 0.0;
    if (head) {
      uint32_t e = s + 1;
      while (e < r1 && kd[e] == dd) ++e;
      term = __dmul_rn((double)(e - s), dd);
      if (s >= req_len) atomicMin(&s_eff, s);  // groups are added while eff_len < req_len (:648-649): the first boundary at or beyond it
    }
    dist[s] = term;  // (dist is put back in column order below)
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double acc = 0.0;
#pragma unroll 8
    for (uint32_t s = 0; s < r1; ++s) acc = __dadd_rn(acc, dist[s]);
    s_stats[0] = (r1 > 0) ? acc / (double)r1 : 0.0;
    s_stats[2] = (r1 > 0) ? kd[r1 / 2] : 0.0;  // the upper median: the group that holds position r1 / 2 (:645-647)
  }
  __syncthreads();
  for (uint32_t s = threadIdx.x; s < r1; s += blockDim.x) dist[ki[s]] = kd[s];  // column order again
  __syncthreads();
  if (threadIdx.x == 0) {
    // :657-670 squared deviations in column order
    const double mean = s_stats[0];
    double acc = 0.0;
#pragma unroll 8
    for (uint32_t c = 0; c < r1; ++c) {
      const double dv = __dsub_rn(dist[c], mean);
      acc = __dadd_rn(acc, __dmul_rn(dv, dv));
    }
    s_stats[1] = (r1 > 1) ? sqrt(acc / ((double)r1 - 1.0)) : 0.0;  // :679-683
  }
  __syncthreads();
  // neighbours: the first eff_len entries of the multimap (:685-689)
  {
    const double mean = s_stats[0], sd = s_stats[1];
    const uint32_t n_out = min(s_eff, max_neighbours);
    for (uint32_t q = threadIdx.x; q < n_out; q += blockDim.x) {
      out_idx[(uint64_t)j * max_neighbours + q] = ki[q];
      out_dist[(uint64_t)j * max_neighbours + q] = kd[q];
      double zz = __dsub_rn(kd[q], mean) / sd;
      // sd = 0 makes this 0/0 (lib/Matrix.ml:688 is unguarded).  The reference runs on x86-64, whose invalid
      // operations return the sign-set quiet NaN ("-nan" through %.15g); gfx950 returns the positive one.
      if (zz != zz) zz = __longlong_as_double((long long)0xFFF8000000000000ull);
      out_z[(uint64_t)j * max_neighbours + q] = zz;
    }
  }
  __syncthreads();
  // MAD: |d - median| sorted, element n/2 (:659-678).  kd is sorted: |kd[i] - median| falls to the median's position and
  // rises after it, padding last -- a bitonic sequence, which the network's last phase alone puts in order
  {
    const double median = s_stats[2];
    for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) kd[i] = (i < r1) ? fabs(__dsub_rn(kd[i], median)) : inf;
  }
  block_merge_keys(kd, NP);
  if (threadIdx.x == 0) {
    out_stats[(uint64_t)j * 4 + 0] = s_stats[0];
    out_stats[(uint64_t)j * 4 + 1] = s_stats[1];
    out_stats[(uint64_t)j * 4 + 2] = s_stats[2];
    out_stats[(uint64_t)j * 4 + 3] = (r1 > 0) ? kd[r1 / 2] : 0.0;
    out_n[j] = s_eff;
  }
}

// ---------------------------------------------------------------------------
// summary against a SMALL first operand (the classifier's case, README.md:656: a few dozen class vectors): one
// WAVEFRONT per m2 row instead of one block.  The first operand sits in LDS, transposed ([dim][row], so the lanes of a
// wave -- one m1 row each -- read consecutive words), and a block walks over m2 rows grid-stride, four at a time.  The
// (distance, column) pairs are sorted in registers (wave_bitonic_sort_pairs), and the walk of the multimap
// (lib/Matrix.ml:640-655) and the squared deviations (:657-670) stay the reference's sequential chains, run by one lane
// over at most 64 R values: same operations in the same order as distance_summary_kernel, so the same bits.
// Needs r1 <= 512 and (PRE or r1 x n_dims doubles and four waves' buffers within a CU's LDS); a first operand of up to 512
// rows that does not fit goes through distance rows in the workspace and the PRE variant (summary_impl).
// ---------------------------------------------------------------------------
constexpr size_t kWaveSummaryHalf = 78u << 10, kWaveSummaryWhole = 159u << 10;
constexpr int kSummaryWaves = 16;  // per block: they share the LDS copy of the first operand, and 2 blocks fill a CU's 32 wave slots

__device__ __forceinline__ uint64_t dist_key(double x) {  // order-preserving map of an f64 to u64
  const uint64_t b = (uint64_t)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_dist(uint64_t k) {
  const uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

constexpr int kWaveTail = 8;  // columns beyond 64 R that the TAIL variant takes in by rank instead of doubling the network

__device__ __forceinline__ double wave_readlane_f64(double v, int l) {  // l uniform
  return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), l) << 32) |
                              (unsigned int)__builtin_amdgcn_readlane((int)__double_as_longlong(v), l));
}
__device__ __forceinline__ uint64_t wave_readlane_u64(uint64_t v, int l) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}

// R: the 64-column slots sorted by the network (a power of two).  TAIL: 1..kWaveTail columns more (r1 = 64 R + T), whose
// distances are worked out apart -- for eight rows of the wave at a time, a lane per (row, column) -- and take their places
// among the sorted 64 R by rank.  65 classes (the bench's, and any set "a few more than 64") then cost one slot's network
// and one slot's distances, not two with 63 of the second slot's 64 lanes idle.
template <int KIND, bool PRE, int R, bool TAIL>
__global__ __launch_bounds__(64 * kSummaryWaves) void distance_summary_wave_kernel(
    const double *__restrict__ a, uint32_t r1, const double *__restrict__ b, uint32_t r2, uint32_t n_dims,
    const double *__restrict__ metric, double p, uint32_t req_len, uint32_t max_neighbours,
    double *__restrict__ out_stats, uint32_t *__restrict__ out_n, uint32_t *__restrict__ out_idx,
    double *__restrict__ out_dist, double *__restrict__ out_z, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int RS = R + (TAIL ? 1 : 0);                    // stripes of 64 sorted positions
  constexpr int RP = RS <= 1 ? 1 : RS <= 2 ? 2 : RS <= 4 ? 4 : RS <= 8 ? 8 : 16;  // ... rounded up to a power of two (the MAD's merge)
  constexpr int N = 64 * RS;
  // per wave: chain[N] (terms of the sequential sums), kd[N] (sorted), ki[N], brow[n_dims]; then the transposed first operand
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t per_wave = (uint32_t)N * 20 + (PRE ? 0u : n_dims * 8);
  unsigned char *mine = smem + (size_t)wv * ((per_wave + 15) & ~15u);
  double *chain = reinterpret_cast<double *>(mine);
  double *kd = chain + N;
  uint32_t *ki = reinterpret_cast<uint32_t *>(kd + N);
  double *s_b = reinterpret_cast<double *>(ki + N);
  const uint32_t n_waves = blockDim.x >> 6, n_threads = blockDim.x;
  double *As = reinterpret_cast<double *>(smem + (size_t)n_waves * ((per_wave + 15) & ~15u));  // [n_dims][r1] + metric[n_dims]
  double *s_metric = As + (size_t)n_dims * r1;
  if (!PRE) {
    for (uint32_t e = threadIdx.x; e < r1 * n_dims; e += n_threads) {
      const uint32_t i = e / n_dims, c = e % n_dims;  // coalesced read of row-major m1
      As[(size_t)c * r1 + i] = a[e];
    }
    for (uint32_t c = threadIdx.x; c < n_dims; c += n_threads) s_metric[c] = metric[c];
  }
  __syncthreads();
  const double inf = __longlong_as_double(0x7FF0000000000000ll);
  const uint32_t n_tail = TAIL ? r1 - 64u * R : 0u;  // 1..kWaveTail
  const uint32_t j_stride = gridDim.x * n_waves;
  double tail_batch = 0.0;  // TAIL, not PRE: lane l holds the distance of row (l >> 3) of the batch to column 64 R + (l & 7)
  uint32_t it = 0;
  for (uint32_t j = blockIdx.x * n_waves + wv; j < r2; j += j_stride, ++it) {
    // distances of row j to every m1 row (lib/Matrix.ml:744-749): lane owns columns lane, lane + 64, ...
    double dv[R];
    double dt = inf;  // TAIL: lane t < n_tail holds the distance to column 64 R + t
    if (PRE) {
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const uint32_t i = (uint32_t)lane + 64u * q;
        dv[q] = i < r1 ? a[(uint64_t)j * r1 + i] : inf;
      }
      if (TAIL && (uint32_t)lane < n_tail) dt = a[(uint64_t)j * r1 + 64u * R + lane];
    } else {
      if (TAIL) {
        if ((it & 7u) == 0) {  // the tail columns of this and the wave's next seven rows
          const uint32_t jr = j + (uint32_t)(lane >> 3) * j_stride, t = (uint32_t)lane & 7u;
          const bool ok = jr < r2 && t < n_tail;
          const double *brow = b + (uint64_t)(ok ? jr : j) * n_dims;
          const double *acol = As + 64u * R + (ok ? t : 0u);
          double acc = 0.0;
#pragma unroll 8
          for (uint32_t c = 0; c < n_dims; ++c) {
            const double diff = __dsub_rn(acol[(size_t)c * r1], brow[c]);
            acc = __dadd_rn(acc, component<KIND>(diff, s_metric[c], p));
          }
          tail_batch = ok ? scale_distance<KIND>(acc, p) : inf;
        }
        const double mine_t = __shfl(tail_batch, (int)((it & 7u) * 8u) + (lane & 7), 64);
        dt = (uint32_t)lane < n_tail ? mine_t : inf;
      }
      for (uint32_t c = lane; c < n_dims; c += 64) s_b[c] = b[(uint64_t)j * n_dims + c];
      __builtin_amdgcn_wave_barrier();
      double acc[R];
#pragma unroll
      for (int q = 0; q < R; ++q) acc[q] = 0.0;
      for (uint32_t c = 0; c < n_dims && !(dbg & 8); ++c) {
        const double bc = s_b[c], mc = s_metric[c];
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const uint32_t i = (uint32_t)lane + 64u * q;
          const double av = i < r1 ? As[(size_t)c * r1 + i] : 0.0;
          const double diff = __dsub_rn(av, bc);
          acc[q] = __dadd_rn(acc[q], component<KIND>(diff, mc, p));
        }
      }
#pragma unroll
      for (int q = 0; q < R; ++q) dv[q] = ((uint32_t)lane + 64u * q) < r1 ? scale_distance<KIND>(acc[q], p) : inf;
    }
    uint64_t key[R];
    uint32_t val[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const uint32_t i = (uint32_t)lane + 64u * q;
      key[q] = i < r1 ? dist_key(dv[q]) : ~0ull;  // padding sorts last (a NaN distance would too; none arises from finite rows)
      val[q] = i < r1 ? i : 0xFFFFFFFFu;
    }
    if (!(dbg & 1)) wave_bitonic_sort_pairs<R, uint64_t>(key, val, lane);
    if (!TAIL) {
#pragma unroll
      for (int q = 0; q < R; ++q) {  // sorted position of (lane, q) is lane * R + q
        kd[lane * R + q] = key_dist(key[q]);
        ki[lane * R + q] = val[q];
      }
    } else {
      // the tail's columns come after every sorted one (larger index): a tail element goes behind the sorted elements <= it
      // and the tail elements before it in (distance, column) order; a sorted element moves up by the tail elements < it
      const uint64_t tkey = (uint32_t)lane < n_tail ? dist_key(dt) : ~0ull;
      uint32_t shift[R], my_pos = 0;
#pragma unroll
      for (int q = 0; q < R; ++q) shift[q] = 0;
      for (uint32_t t = 0; t < n_tail; ++t) {
        const uint64_t tk = wave_readlane_u64(tkey, (int)t);
        uint32_t pos = 0;
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const bool before = key[q] <= tk;
          pos += (uint32_t)__popcll(__ballot(before));
          shift[q] += before ? 0u : 1u;
        }
        pos += (uint32_t)__popcll(__ballot((uint32_t)lane < n_tail && (tkey < tk || (tkey == tk && (uint32_t)lane < t))));
        if ((uint32_t)lane == t) my_pos = pos;
      }
#pragma unroll
      for (int q = 0; q < R; ++q) {
        kd[lane * R + q + shift[q]] = key_dist(key[q]);
        ki[lane * R + q + shift[q]] = val[q];
      }
      if ((uint32_t)lane < n_tail) {
        kd[my_pos] = dt;
        ki[my_pos] = 64u * R + lane;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // The walk of the multimap (lib/Matrix.ml:640-655) and the squared deviations (:657-670) are sequential chains of
    // f64 additions in the reference, and stay so here: the sorted values come back striped (position q * 64 + lane), each
    // lane works out the term of its position, the terms go to LDS and the chain reads them back, every lane the same
    // word -- one ds_read per term (two v_readlane and their hazards per term made the chains a quarter of the kernel).
    double sv[RS];     // sorted distance at position q * 64 + lane (inf beyond r1)
    uint64_t hm[RS];   // lanes of stripe q that start a group of equal distances
#pragma unroll
    for (int q = 0; q < RS; ++q) {
      const uint32_t pos = (uint32_t)q * 64u + lane;
      sv[q] = pos < r1 ? kd[pos] : inf;
      const double prev = pos > 0 && pos < r1 ? kd[pos - 1] : 0.0;
      hm[q] = __ballot(pos < r1 && (pos == 0 || prev != sv[q]));
    }
    // position of the next group start after this one (r1 if none): the length of a group is that minus its start
#pragma unroll
    for (int q = 0; q < RS; ++q) {
      uint32_t next = r1;
      const uint64_t above = lane == 63 ? 0ull : (hm[q] >> (lane + 1));
      if (above) next = (uint32_t)q * 64u + lane + 1u + (uint32_t)__ffsll((long long)above) - 1u;
      else {
#pragma unroll
        for (int q2 = RS - 1; q2 > q; --q2)
          if (hm[q2]) next = (uint32_t)q2 * 64u + (uint32_t)__ffsll((long long)hm[q2]) - 1u;
      }
      const uint32_t pos = (uint32_t)q * 64u + lane;
      const bool head = (hm[q] >> lane) & 1ull;
      chain[pos] = head ? __dmul_rn((double)(next - pos), sv[q]) : 0.0;  // set_len *. dist (:643)
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t chain_len = (dbg & 2) ? min(r1, 1u) : r1;
    double acc = 0.0;
#pragma unroll 8
    for (uint32_t i = 0; i < chain_len; ++i) acc = __dadd_rn(acc, chain[i]);  // a position that starts no group adds +0.0: exact, the sum is never -0.0
    const double mean = (r1 > 0) ? acc / (double)r1 : 0.0;
    // upper median = the value at sorted position r1 / 2 (:645-647 picks the group that holds it)
    const double median = r1 > 0 ? kd[r1 / 2] : 0.0;
    // groups are added while eff_len < req_len (:648-649): the first group boundary at or beyond req_len
    uint32_t eff = r1;
    if (req_len < r1) {
#pragma unroll
      for (int q = RS - 1; q >= 0; --q) {
        const uint32_t lo = req_len > (uint32_t)q * 64u ? req_len - (uint32_t)q * 64u : 0u;  // lanes of this stripe at or beyond req_len
        const uint64_t m = lo >= 64u ? 0ull : (hm[q] >> lo) << lo;
        if (m) eff = (uint32_t)q * 64u + (uint32_t)__ffsll((long long)m) - 1u;
      }
    }
    // squared deviations in column order (:657-670); dv[q] is column lane + 64 q, dt column 64 R + lane
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const double d0 = __dsub_rn(dv[q], mean);
      chain[q * 64 + lane] = __dmul_rn(d0, d0);  // (columns beyond r1 are not read)
    }
    if (TAIL && (uint32_t)lane < n_tail) {
      const double d0 = __dsub_rn(dt, mean);
      chain[64 * R + lane] = __dmul_rn(d0, d0);
    }
    __builtin_amdgcn_wave_barrier();
    acc = 0.0;
#pragma unroll 8
    for (uint32_t i = 0; i < chain_len; ++i) acc = __dadd_rn(acc, chain[i]);
    const double sd = (r1 > 1) ? sqrt(acc / ((double)r1 - 1.0)) : 0.0;  // :679-683
    const uint32_t n_out = min(eff, max_neighbours);
    for (uint32_t q = lane; q < n_out; q += 64) {  // :685-689
      out_idx[(uint64_t)j * max_neighbours + q] = ki[q];
      out_dist[(uint64_t)j * max_neighbours + q] = kd[q];
      double zz = __dsub_rn(kd[q], mean) / sd;
      if (zz != zz) zz = __longlong_as_double((long long)0xFFF8000000000000ull);  // see distance_summary_kernel
      out_z[(uint64_t)j * max_neighbours + q] = zz;
    }
    // MAD: |d - median| sorted, element n/2 (:659-678).  Over the SORTED distances |d - median| falls down to the median's
    // position and rises after it (the subtraction and fabs are monotone), padding last: a bitonic sequence, which one merge
    // -- log2 N stages, not a sort's 28 -- puts in order.  The same values as over the columns, so the same element.
    uint64_t mk[RP];
#pragma unroll
    for (int q = 0; q < RP; ++q) mk[q] = ~0ull;
#pragma unroll
    for (int q = 0; q < RS; ++q)
      if (((uint32_t)q * 64u + lane) < r1) mk[q] = dist_key(fabs(__dsub_rn(sv[q], median)));
    if (!(dbg & 4)) wave_bitonic_merge_striped<RP, uint64_t>(mk, lane);
    const uint32_t mpos = r1 / 2;  // sorted position q * 64 + lane
    double mad = 0.0;
#pragma unroll
    for (int q = 0; q < RS; ++q)
      if (r1 > 0 && (mpos >> 6) == (uint32_t)q) mad = wave_readlane_f64(key_dist(mk[q]), (int)(mpos & 63u));
    if (lane == 0) {
      out_stats[(uint64_t)j * 4 + 0] = mean;
      out_stats[(uint64_t)j * 4 + 1] = sd;
      out_stats[(uint64_t)j * 4 + 2] = median;
      out_stats[(uint64_t)j * 4 + 3] = mad;
      out_n[j] = eff;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int KIND, bool PRE, int R, bool TAIL>
static int launch_summary_wave_r(const double *a, uint32_t r1, const double *b, uint32_t r2, uint32_t n_dims, const double *metric,
                                 double p, uint32_t req_len, uint32_t max_neighbours, double *out_stats, uint32_t *out_n,
                                 uint32_t *out_idx, double *out_dist, double *out_z, hipStream_t st) {
  const uint32_t per_wave = ((uint32_t)(64 * (R + (TAIL ? 1 : 0))) * 20 + (PRE ? 0u : n_dims * 8) + 15) & ~15u;
  const size_t shared = PRE ? 0 : ((size_t)n_dims * r1 + n_dims) * 8;
  // as many waves per block as the LDS left beside the shared operand allows: two blocks per CU (78 KB each) while eight
  // waves still fit beside it, one block with the whole LDS above that (100 classes x 64 dimensions: 51 KB of operand)
  const bool two = shared + 8 * (size_t)per_wave <= kWaveSummaryHalf;
  const size_t budget = two ? kWaveSummaryHalf : kWaveSummaryWhole;
  const uint32_t waves = (uint32_t)std::max<size_t>(1, std::min<size_t>(kSummaryWaves, (budget - shared) / per_wave));
  const size_t smem = (size_t)waves * per_wave + shared;
  static PerSlotOnce attr_once;
  bool &attr_set = attr_once();
  if (!attr_set) {
    KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&distance_summary_wave_kernel<KIND, PRE, R, TAIL>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    attr_set = true;
  }
  const uint32_t blocks = std::min<uint32_t>(div_up(r2, waves), (uint32_t)ctx().n_cus * (two ? 2 : 1));
  distance_summary_wave_kernel<KIND, PRE, R, TAIL><<<dim3(blocks), dim3(64 * waves), smem, st>>>(
      a, r1, b, r2, n_dims, metric, p, req_len, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z, (ctx().tune_dbg >> 16) & 15);
  KPOP_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// The same summary with JB second-operand rows of a wavefront in hand at once (round 4).  What a row's summary shares with its
// neighbours' is worth more than it looks:
//  * the distances: a lane reads its column of the first operand from LDS ONCE per dimension for JB rows (the one-row kernel read
//    512 B per row and dimension -- with four SIMDs on a CU exactly the LDS's 128 B a clock, so it ran at the LDS's rate, not the
//    vector pipe's); the rows' own values are broadcast reads of [dimension][JB] words;
//  * everything that is ONE value per row -- the two sequential chains (mean: lib/Matrix.ml:640-655; squared deviations:
//    :657-670), the divisions and square roots behind mean and standard deviation, the z-scores -- is done by lane r for row r,
//    JB rows in the time of one (a wave64 instruction costs the same for one lane as for 64);
//  * the MAD (:659-678) needs no network: over the sorted distances, |d - median| is two sorted runs (falling before the
//    median's position, rising from it), and the element of rank r1 / 2 of their union is found by testing all splits at once
//    (one lane a split, one ballot), reading the runs where they lie in LDS;
//  * the sort's exchanges run in the vector pipe (wave_sort.h).
// Same operations in the same order on every value: the same bits as distance_summary_wave_kernel and distance_summary_kernel.
// Batches go to wavefronts block-cyclically (batch = it * W + wave * blocks + block), so the blocks of a launch differ by at most
// one batch whatever r2 is.
// ---------------------------------------------------------------------------
template <int KIND, bool PRE, int R, bool TAIL, int JB>
__global__ __launch_bounds__(64 * kSummaryWaves) void distance_summary_batch_kernel(
    const double *__restrict__ a, uint32_t r1, const double *__restrict__ b, uint32_t r2, uint32_t n_dims,
    const double *__restrict__ metric, double p, uint32_t req_len, uint32_t max_neighbours, uint32_t stride, uint32_t per_wave,
    double *__restrict__ out_stats, uint32_t *__restrict__ out_n, uint32_t *__restrict__ out_idx,
    double *__restrict__ out_dist, double *__restrict__ out_z, int dbg, const double *__restrict__ nb) {
  // nb (not PRE): the norms of the second operand's rows, which then come as they are and are divided as they are staged
  // (lib/Matrix.ml:247-249: the same quotients a divided copy holds -- one pass over the rows less to write and to read)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int RS = R + (TAIL ? 1 : 0);  // stripes of 64 sorted positions
  constexpr int TW = 64 / JB;             // lanes a row has in the passes that take one lane per (row, item)
  static_assert(!TAIL || kWaveTail <= TW, "a row's tail columns fit its lanes");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t n_waves = blockDim.x >> 6, n_threads = blockDim.x;
  // per wave: kd[JB][stride] (sorted distances), chain[JB][cs] (terms of the sequential sums; before that the batch's
  // second-operand rows, [dimension][JB]), ki[JB][stride] (sorted columns), one word per row of mean, sd, median, MAD, eff_len
  const uint32_t cs = PRE ? stride : max(stride, n_dims);
  unsigned char *mine = smem + (size_t)wv * per_wave;
  double *kd = reinterpret_cast<double *>(mine);
  double *chain = kd + (size_t)JB * stride;
  uint32_t *ki = reinterpret_cast<uint32_t *>(chain + (size_t)JB * cs);
  double *s_mean = reinterpret_cast<double *>(ki + (size_t)JB * stride);
  double *s_sd = s_mean + JB, *s_med = s_sd + JB, *s_mad = s_med + JB;
  uint32_t *s_eff = reinterpret_cast<uint32_t *>(s_mad + JB);
  double *s_bT = chain;
  double *As = reinterpret_cast<double *>(smem + (size_t)n_waves * per_wave);  // [n_dims][r1] + metric[n_dims]
  double *s_metric = As + (size_t)n_dims * r1;
  if (!PRE) {
    for (uint32_t e = threadIdx.x; e < r1 * n_dims; e += n_threads) {
      const uint32_t i = e / n_dims, c = e % n_dims;  // coalesced read of row-major m1
      As[(size_t)c * r1 + i] = a[e];
    }
    for (uint32_t c = threadIdx.x; c < n_dims; c += n_threads) s_metric[c] = metric[c];
  }
  __syncthreads();
  const double inf = __longlong_as_double(0x7FF0000000000000ll);
  const uint32_t n_tail = TAIL ? r1 - 64u * R : 0u;  // 1..kWaveTail
  const uint32_t n_batches = (r2 + JB - 1) / JB, W = gridDim.x * n_waves;
  const uint32_t m = r1 / 2;  // the upper median's position (:645-647)
  for (uint32_t bt = (uint32_t)wv * gridDim.x + blockIdx.x; bt < n_batches; bt += W) {
    const uint32_t j0 = bt * JB, nv = min((uint32_t)JB, r2 - j0);  // (rows beyond nv repeat the last one and are not written)
    double dv[JB][R];  // distance of row r to column lane + 64 q (lib/Matrix.ml:744-749)
    double dt[JB];     // TAIL: lane t < n_tail holds row r's distance to column 64 R + t
#pragma unroll
    for (int r = 0; r < JB; ++r) dt[r] = inf;
    if (PRE) {
#pragma unroll
      for (int r = 0; r < JB; ++r) {
        const double *arow = a + (uint64_t)(j0 + min((uint32_t)r, nv - 1)) * r1;
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const uint32_t i = (uint32_t)lane + 64u * q;
          dv[r][q] = i < r1 ? arow[i] : inf;
        }
        if (TAIL && (uint32_t)lane < n_tail) dt[r] = arow[64u * R + lane];
      }
    } else {
#pragma unroll
      for (int r = 0; r < JB; ++r) {
        const uint32_t jr = j0 + min((uint32_t)r, nv - 1);
        const double *brow = b + (uint64_t)jr * n_dims;
        if (nb) {
          const double nrm = nb[jr];
          for (uint32_t c = lane; c < n_dims; c += 64) s_bT[(size_t)c * JB + r] = __ddiv_rn(brow[c], nrm);
        } else {
          for (uint32_t c = lane; c < n_dims; c += 64) s_bT[(size_t)c * JB + r] = brow[c];
        }
      }
      __builtin_amdgcn_wave_barrier();
      double acc[JB][R];
#pragma unroll
      for (int r = 0; r < JB; ++r)
#pragma unroll
        for (int q = 0; q < R; ++q) acc[r][q] = 0.0;
      uint32_t col[R];
#pragma unroll
      for (int q = 0; q < R; ++q) col[q] = min((uint32_t)lane + 64u * q, r1 - 1);  // (lanes beyond r1 work on a copy of the last column)
      // TAIL: the columns beyond 64 R ride along, lane (r, t) = (lane / TW, lane % TW) working out row r's distance to column 64 R + t
      const uint32_t tr = (uint32_t)lane / TW, tt = (uint32_t)lane % TW;
      const double *acol = As + 64u * R + (TAIL ? min(tt, n_tail - 1) : 0u);
      double tacc = 0.0;
      // the operands of dimension c + 1 are asked for before dimension c's arithmetic starts: an LDS read is ~100 cycles, a
      // dimension's arithmetic 64 x JB / 4, and the compiler on its own waits for every read right after issuing it
      struct Dim {
        double mc, bv[JB], av[R], ta, tb;
      };
      auto fetch = [&](uint32_t c) {
        Dim d;
        d.mc = s_metric[c];
#pragma unroll
        for (int r = 0; r < JB; ++r) d.bv[r] = s_bT[(size_t)c * JB + r];
#pragma unroll
        for (int q = 0; q < R; ++q) d.av[q] = As[(size_t)c * r1 + col[q]];
        d.ta = TAIL ? acol[(size_t)c * r1] : 0.0;
        d.tb = TAIL ? s_bT[(size_t)c * JB + tr] : 0.0;
        return d;
      };
      auto work = [&](const Dim &d) {
#pragma unroll
        for (int r = 0; r < JB; ++r)
#pragma unroll
          for (int q = 0; q < R; ++q) {
            const double diff = __dsub_rn(d.av[q], d.bv[r]);
            acc[r][q] = __dadd_rn(acc[r][q], component<KIND>(diff, d.mc, p));
          }
        if (TAIL) {
          const double diff = __dsub_rn(d.ta, d.tb);
          tacc = __dadd_rn(tacc, component<KIND>(diff, d.mc, p));
        }
      };
      const uint32_t n_d = (dbg & 8) ? 0u : n_dims;
      if (n_d) {
        Dim cur = fetch(0);
        for (uint32_t c = 0; c + 1 < n_d; ++c) {
          const Dim nxt = fetch(c + 1);
          work(cur);
          cur = nxt;
        }
        work(cur);
      }
#pragma unroll
      for (int r = 0; r < JB; ++r)
#pragma unroll
        for (int q = 0; q < R; ++q) dv[r][q] = ((uint32_t)lane + 64u * q) < r1 ? scale_distance<KIND>(acc[r][q], p) : inf;
      if (TAIL) {
        const double td = scale_distance<KIND>(tacc, p);
#pragma unroll
        for (int r = 0; r < JB; ++r) {
          const double mine_t = __shfl(td, r * TW + (lane & (TW - 1)), 64);
          dt[r] = (uint32_t)lane < n_tail ? mine_t : inf;
        }
      }
      __builtin_amdgcn_wave_barrier();  // the rows' values are done with: their words become the chains'
    }
    // ---- a row at a time: sort, the multimap's terms, eff_len, median, MAD
#pragma unroll
    for (int r = 0; r < JB; ++r) {
      double *kd_r = kd + (size_t)r * stride, *chain_r = chain + (size_t)r * cs;
      uint32_t *ki_r = ki + (size_t)r * stride;
      uint64_t key[R];
      uint32_t val[R];
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const uint32_t i = (uint32_t)lane + 64u * q;
        key[q] = i < r1 ? dist_key(dv[r][q]) : ~0ull;  // padding sorts last
        val[q] = i < r1 ? i : 0xFFFFFFFFu;
      }
      if (!(dbg & 1)) wave_bitonic_sort_pairs<R, uint64_t>(key, val, lane);
      if (!TAIL) {
#pragma unroll
        for (int q = 0; q < R; ++q) {  // sorted position of (lane, q) is lane * R + q; padding lies beyond r1
          const uint32_t pos = (uint32_t)lane * R + q;
          if (pos < r1) {
            kd_r[pos] = key_dist(key[q]);
            ki_r[pos] = val[q];
          }
        }
      } else {
        // (as in distance_summary_wave_kernel) a tail element goes behind the sorted elements <= it and the tail elements before it
        // in (distance, column) order; a sorted element moves up by the tail elements < it
        const uint64_t tkey = (uint32_t)lane < n_tail ? dist_key(dt[r]) : ~0ull;
        uint32_t shift[R], my_pos = 0;
#pragma unroll
        for (int q = 0; q < R; ++q) shift[q] = 0;
        for (uint32_t t = 0; t < n_tail; ++t) {
          const uint64_t tk = wave_readlane_u64(tkey, (int)t);
          uint32_t pos = 0;
#pragma unroll
          for (int q = 0; q < R; ++q) {
            const bool before = key[q] <= tk;
            pos += (uint32_t)__popcll(__ballot(before));
            shift[q] += before ? 0u : 1u;
          }
          pos += (uint32_t)__popcll(__ballot((uint32_t)lane < n_tail && (tkey < tk || (tkey == tk && (uint32_t)lane < t))));
          if ((uint32_t)lane == t) my_pos = pos;
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
          kd_r[lane * R + q + shift[q]] = key_dist(key[q]);
          ki_r[lane * R + q + shift[q]] = val[q];
        }
        if ((uint32_t)lane < n_tail) {
          kd_r[my_pos] = dt[r];
          ki_r[my_pos] = 64u * R + lane;
        }
      }
      __builtin_amdgcn_wave_barrier();
      // the terms of the multimap's walk (:640-655): a position that starts a group of equal distances holds set_len *. dist
      double sv[RS];
      uint64_t hm[RS];
#pragma unroll
      for (int q = 0; q < RS; ++q) {
        const uint32_t pos = (uint32_t)q * 64u + lane;
        sv[q] = pos < r1 ? kd_r[pos] : inf;
        const double prev = pos > 0 && pos < r1 ? kd_r[pos - 1] : 0.0;
        hm[q] = __ballot(pos < r1 && (pos == 0 || prev != sv[q]));
      }
#pragma unroll
      for (int q = 0; q < RS; ++q) {
        uint32_t next = r1;
        const uint64_t above = lane == 63 ? 0ull : (hm[q] >> (lane + 1));
        if (above) next = (uint32_t)q * 64u + lane + 1u + (uint32_t)__ffsll((long long)above) - 1u;
        else {
#pragma unroll
          for (int q2 = RS - 1; q2 > q; --q2)
            if (hm[q2]) next = (uint32_t)q2 * 64u + (uint32_t)__ffsll((long long)hm[q2]) - 1u;
        }
        const uint32_t pos = (uint32_t)q * 64u + lane;
        const bool head = (hm[q] >> lane) & 1ull;
        if (pos < r1) chain_r[pos] = head ? __dmul_rn((double)(next - pos), sv[q]) : 0.0;  // set_len *. dist (:643)
      }
      // groups are added while eff_len < req_len (:648-649): the first group boundary at or beyond req_len
      uint32_t eff = r1;
      if (req_len < r1) {
#pragma unroll
        for (int q = RS - 1; q >= 0; --q) {
          const uint32_t lo = req_len > (uint32_t)q * 64u ? req_len - (uint32_t)q * 64u : 0u;
          const uint64_t mm = lo >= 64u ? 0ull : (hm[q] >> lo) << lo;
          if (mm) eff = (uint32_t)q * 64u + (uint32_t)__ffsll((long long)mm) - 1u;
        }
      }
      const double median = kd_r[m];
      // MAD = the element of rank m of |d - median| (:659-678).  Before the median's position the values fall, from it they
      // rise: u_i = |kd[m - 1 - i] - median| (i < m) and w_i = |kd[m + i] - median| (i < r1 - m) are both ascending.  Taking i
      // from u and m + 1 - i from w gives the m + 1 smallest when u_{i-1} <= w_{m+1-i} (true up to some i, false after); the
      // largest such i is the count of lanes that say yes, less one, and the element of rank m is max(u_{i-1}, w_{m-i}).
      uint32_t yes = 0;
      for (uint32_t i0 = 0; i0 <= m && !(dbg & 4); i0 += 64) {
        const uint32_t i = i0 + lane;
        bool ok = false;
        if (i <= m) {
          ok = i == 0 || 2 * m + 1 - i >= r1;  // nothing taken from u yet / w has run out: w_{m+1-i} counts as +inf
          if (!ok) ok = fabs(__dsub_rn(kd_r[m - i], median)) <= fabs(__dsub_rn(kd_r[2 * m + 1 - i], median));
        }
        yes += (uint32_t)__popcll(__ballot(ok));
      }
      const uint32_t is = yes ? yes - 1 : 0;
      double mad = fabs(__dsub_rn(kd_r[min(2 * m - is, r1 - 1)], median));  // w_{m-is} (there whenever is is the largest yes)
      if (is >= 1) mad = fmax(mad, fabs(__dsub_rn(kd_r[m - is], median)));
      if (lane == r) {
        s_med[r] = median;
        s_mad[r] = mad;
        s_eff[r] = eff;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- lane r for row r: the walk's sum in the multimap's order (:640-655)
    const int rr = min(lane, JB - 1);
    const uint32_t chain_len = (dbg & 2) ? min(r1, 1u) : r1;
    {
      const double *my = chain + (size_t)rr * cs;
      double acc = 0.0;
#pragma unroll 8
      for (uint32_t i = 0; i < chain_len; ++i) acc = __dadd_rn(acc, my[i]);  // a position that starts no group adds +0.0: exact
      if (lane < JB) s_mean[lane] = acc / (double)r1;
    }
    __builtin_amdgcn_wave_barrier();
    // squared deviations in column order (:657-670)
#pragma unroll
    for (int r = 0; r < JB; ++r) {
      const double mean = s_mean[r];
      double *chain_r = chain + (size_t)r * cs;
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const uint32_t i = (uint32_t)lane + 64u * q;
        const double d0 = __dsub_rn(dv[r][q], mean);
        if (i < r1) chain_r[i] = __dmul_rn(d0, d0);
      }
      if (TAIL && (uint32_t)lane < n_tail) {
        const double d0 = __dsub_rn(dt[r], mean);
        chain_r[64 * R + lane] = __dmul_rn(d0, d0);
      }
    }
    __builtin_amdgcn_wave_barrier();
    {
      const double *my = chain + (size_t)rr * cs;
      double acc = 0.0;
#pragma unroll 8
      for (uint32_t i = 0; i < chain_len; ++i) acc = __dadd_rn(acc, my[i]);
      const double sd = (r1 > 1) ? sqrt(acc / ((double)r1 - 1.0)) : 0.0;  // :679-683
      if (lane < JB) s_sd[lane] = sd;
      if ((uint32_t)lane < nv) {
        double *st = out_stats + (uint64_t)(j0 + lane) * 4;
        st[0] = s_mean[lane];
        st[1] = sd;
        st[2] = s_med[lane];
        st[3] = s_mad[lane];
        out_n[j0 + lane] = s_eff[lane];
      }
    }
    __builtin_amdgcn_wave_barrier();
    // neighbours: the first eff_len entries of the multimap (:685-689), one lane per (row, entry) while the lists are short
    uint32_t longest = 0;
#pragma unroll
    for (int r = 0; r < JB; ++r) longest = max(longest, (uint32_t)r < nv ? min(s_eff[r], max_neighbours) : 0u);
    auto put = [&](uint32_t r, uint32_t q) {
      const uint64_t o = (uint64_t)(j0 + r) * max_neighbours + q;
      const double d = kd[(size_t)r * stride + q];
      out_idx[o] = ki[(size_t)r * stride + q];
      out_dist[o] = d;
      double zz = __dsub_rn(d, s_mean[r]) / s_sd[r];
      if (zz != zz) zz = __longlong_as_double((long long)0xFFF8000000000000ull);  // see distance_summary_kernel
      out_z[o] = zz;
    };
    if (longest <= (uint32_t)TW) {
      const uint32_t r = (uint32_t)lane / TW, q = (uint32_t)lane % TW;
      if (r < nv && q < min(s_eff[r], max_neighbours)) put(r, q);
    } else {
      for (uint32_t r = 0; r < nv; ++r) {
        const uint32_t n_out = min(s_eff[r], max_neighbours);
        for (uint32_t q = lane; q < n_out; q += 64) put(r, q);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int R, bool TAIL>
constexpr int summary_batch_rows() {  // JB: rows in hand at once, by the registers and LDS a row takes
  return R + (TAIL ? 1 : 0) <= 2 ? 4 : 2;
}
static inline uint32_t summary_batch_stride(uint32_t r1) { return (r1 + 1) & ~1u; }
static inline size_t summary_batch_per_wave(uint32_t r1, uint32_t n_dims, bool pre, int jb) {
  const size_t stride = summary_batch_stride(r1), cs = pre ? stride : std::max<size_t>(stride, n_dims);
  return ((size_t)jb * (stride * 12 + cs * 8 + 36) + 15) & ~(size_t)15;
}
static inline bool summary_fits_batch(uint32_t r1, uint32_t n_dims, bool pre, int jb) {  // eight wavefronts beside the first operand
  const size_t shared = pre ? 0 : ((size_t)n_dims * r1 + n_dims) * 8;  // (fewer: the one-row kernel's smaller buffers keep more of them on the CU -- 200 classes x 64 dimensions 1.17 against 1.48 ms)
  return shared + 8 * summary_batch_per_wave(r1, n_dims, pre, jb) <= kWaveSummaryWhole;
}

template <int KIND, bool PRE, int R, bool TAIL>
static int launch_summary_batch_r(const double *a, uint32_t r1, const double *b, uint32_t r2, uint32_t n_dims, const double *metric,
                                  double p, uint32_t req_len, uint32_t max_neighbours, double *out_stats, uint32_t *out_n,
                                  uint32_t *out_idx, double *out_dist, double *out_z, hipStream_t st, const double *nb) {
  constexpr int JB = summary_batch_rows<R, TAIL>();
  const size_t per_wave = summary_batch_per_wave(r1, n_dims, PRE, JB);
  const size_t shared = PRE ? 0 : ((size_t)n_dims * r1 + n_dims) * 8;
  // a multiple of four wavefronts a block (a SIMD each), two blocks a CU when both fit its LDS
  const bool two = 2 * (shared + kSummaryWaves * per_wave) <= kWaveSummaryWhole;
  uint32_t waves = (uint32_t)std::min<size_t>(kSummaryWaves, (kWaveSummaryWhole - shared) / per_wave);
  waves = std::max(4u, waves & ~3u);
  const size_t smem = (size_t)waves * per_wave + shared;
  static PerSlotOnce attr_once;
  bool &attr_set = attr_once();
  if (!attr_set) {
    KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&distance_summary_batch_kernel<KIND, PRE, R, TAIL, JB>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    attr_set = true;
  }
  const uint32_t blocks = std::min<uint32_t>(div_up(div_up(r2, JB), waves), (uint32_t)ctx().n_cus * (two ? 2 : 1));
  distance_summary_batch_kernel<KIND, PRE, R, TAIL, JB><<<dim3(blocks), dim3(64 * waves), smem, st>>>(
      a, r1, b, r2, n_dims, metric, p, req_len, max_neighbours, summary_batch_stride(r1), (uint32_t)per_wave, out_stats, out_n, out_idx,
      out_dist, out_z, (ctx().tune_dbg >> 16) & 15, PRE ? nullptr : nb);
  KPOP_LAUNCH_CHECK();
  return 0;
}

// the stripes of 64 sorted positions a row of r1 distances takes in the wave kernel (R, + 1 with a tail)
static inline uint32_t wave_summary_stripes(uint32_t r1, bool tails) {
  for (uint32_t r = 1; r <= 4; r <<= 1) {
    if (r1 <= 64 * r) return r;
    if (r1 <= 64 * r + kWaveTail && tails) return r + 1;
  }
  return 8;
}
static inline bool summary_fits_wave(uint32_t r1, uint32_t n_dims, bool pre) {
  if (r1 < 1 || r1 > 512) return false;
  if (pre) return true;
  const size_t per_wave = ((size_t)64 * wave_summary_stripes(r1, !(ctx().tune_dbg & 32768)) * 20 + (size_t)n_dims * 8 + 15) & ~(size_t)15;
  return ((size_t)n_dims * r1 + n_dims) * 8 + 4 * per_wave <= kWaveSummaryWhole;  // the first operand and at least four waves' buffers
}

// which instance of distance_summary_batch_kernel takes a first operand of r1 rows: 1 (R 1), 2 (R 1 + tail), 3 (R 2), 4 (R 2 + tail),
// 5 (R 4); 0: none (round 3's kernel, one row of a wavefront at a time)
static int summary_batch_case(uint32_t r1, uint32_t n_dims, bool pre) {
  if (ctx().tune_dbg & (1 << 30)) return 0;  // (round 3's kernel, for A/B)
  const bool tails = !(ctx().tune_dbg & 32768);
  int c = 0;
  if (r1 <= 64) c = 1;
  else if (r1 <= 64 + kWaveTail && tails) c = 2;
  else if (r1 <= 128) c = 3;
  else if (r1 <= 128 + kWaveTail && tails) c = 4;
  else if (r1 <= 256) c = 5;
  if (!c) return 0;
  const int jb = c <= 3 ? 4 : 2;  // (summary_batch_rows)
  return summary_fits_batch(r1, n_dims, pre, jb) ? c : 0;
}

template <int KIND, bool PRE>
static int launch_summary_wave(const double *a, uint32_t r1, const double *b, uint32_t r2, uint32_t n_dims, const double *metric,
                               double p, uint32_t req_len, uint32_t max_neighbours, double *out_stats, uint32_t *out_n,
                               uint32_t *out_idx, double *out_dist, double *out_z, hipStream_t st, const double *nb = nullptr) {
#define KPOP_WAVE(RR, TT) \
  return launch_summary_wave_r<KIND, PRE, RR, TT>(a, r1, b, r2, n_dims, metric, p, req_len, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z, st)
  const bool tails = !(ctx().tune_dbg & 32768);  // (32768: the doubled network for a few columns beyond 64 R, for A/B)
#define KPOP_BATCH(RR, TT) \
  return launch_summary_batch_r<KIND, PRE, RR, TT>(a, r1, b, r2, n_dims, metric, p, req_len, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z, st, nb)
  switch (summary_batch_case(r1, n_dims, PRE)) {
    case 1: KPOP_BATCH(1, false);
    case 2: KPOP_BATCH(1, true);
    case 3: KPOP_BATCH(2, false);
    case 4: KPOP_BATCH(2, true);
    case 5: KPOP_BATCH(4, false);
    default: break;
  }
#undef KPOP_BATCH
  if (r1 <= 64) KPOP_WAVE(1, false);
  if (r1 <= 64 + kWaveTail && tails) KPOP_WAVE(1, true);
  if (r1 <= 128) KPOP_WAVE(2, false);
  if (r1 <= 128 + kWaveTail && tails) KPOP_WAVE(2, true);
  if (r1 <= 256) KPOP_WAVE(4, false);
  if (r1 <= 256 + kWaveTail && tails) KPOP_WAVE(4, true);
  KPOP_WAVE(8, false);
#undef KPOP_WAVE
}

// ---------------------------------------------------------------------------
// host-side orchestration
// ---------------------------------------------------------------------------
struct DistWork {
  double *n1, *n2, *a, *b;
};

static DistWork carve(void *work, uint32_t r1, uint32_t r2, uint32_t n_dims) {
  double *w = reinterpret_cast<double *>(work);
  DistWork d;
  d.n1 = w;
  d.n2 = d.n1 + r1;
  d.a = d.n2 + r2;
  d.b = d.a + (uint64_t)r1 * n_dims;
  return d;
}

template <int KIND>
static int prepare_operands(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                            const double *metric, double p, int normalize, void *work, const double **a,
                            const double **b, hipStream_t st) {
  if (!normalize) {  // n1 = n2 = 1 (lib/Matrix.ml:201-202): x /. 1. = x
    *a = m1;
    *b = m2;
    return 0;
  }
  DistWork w = carve(work, r1, r2, n_dims);
  KPOP_TRY(launch_row_norms_pair<KIND>(r1 ? m1 : nullptr, r1, w.n1, w.a, r2 ? m2 : nullptr, r2, w.n2, w.b, n_dims, metric, p, st));
  *a = w.a;
  *b = w.b;
  return 0;
}

// launches the rowwise kernel on prepared (normalised) operands
// ---------------------------------------------------------------------------
// long rows (n_dims >= kLongD): norms and distances summed slab by slab
// ---------------------------------------------------------------------------
constexpr uint32_t kLongD = 32768, kSlabDims = 4096;

static uint32_t long_slabs(uint32_t r1, uint32_t r2, uint32_t n_dims) {
  const uint64_t pairs = std::max<uint64_t>(1, (uint64_t)r1 * r2);
  const uint64_t cap = std::max<uint64_t>(1, (128ull << 20) / pairs);  // partials stay below 1 GB
  return (uint32_t)std::min<uint64_t>(std::min<uint64_t>(div_up(n_dims, kSlabDims), cap), 65535);
}
static uint32_t slab_dims(uint32_t n_dims, uint32_t slabs) { return (div_up(n_dims, slabs) + kDC - 1) / kDC * kDC; }

// partial[row][z] = sum over the slab of m_c g(a_c)   (block tree; lib/Space.ml:169-178 term by term)
template <int KIND>
__global__ __launch_bounds__(256) void row_norm_partial_kernel(const double *__restrict__ m, uint32_t n_dims, uint32_t slab,
                                                               const double *__restrict__ metric, double p,
                                                               double *__restrict__ partial) {
  const uint32_t row = blockIdx.y, z = blockIdx.x;
  const uint32_t d0 = z * slab, d1 = min(n_dims, d0 + slab);
  const double *src = m + (uint64_t)row * n_dims;
  double acc = 0.0;
  for (uint32_t c = d0 + threadIdx.x; c < d1; c += 256) acc = __dadd_rn(acc, component<KIND>(src[c], metric[c], p));
  __shared__ double sh[4];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(uint64_t)row * gridDim.x + z] = sh[0] + sh[1] + sh[2] + sh[3];
}

template <int KIND>
__global__ void row_norm_final_kernel(const double *__restrict__ partial, uint32_t rows, uint32_t slabs, double p,
                                      double *__restrict__ norms) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  double acc = 0.0;
  for (uint32_t z = 0; z < slabs; ++z) acc += partial[(uint64_t)row * slabs + z];
  const double nv = scale_distance<KIND>(acc, p);
  norms[row] = (nv == 0.0) ? 1.0 : nv;  // lib/Matrix.ml:67
}

__global__ __launch_bounds__(256) void divide_rows_kernel(const double *__restrict__ m, uint32_t n_dims, const double *__restrict__ norms,
                                                          double *__restrict__ out) {
  const uint32_t row = blockIdx.y;
  const double nv = norms[row];
  const uint64_t base = (uint64_t)row * n_dims;
  for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < n_dims; c += gridDim.x * 256) out[base + c] = m[base + c] / nv;
}

template <int KIND>
__global__ void reduce_slabs_kernel(const double *__restrict__ partial, uint64_t pairs, uint32_t slabs, double p,
                                    double *__restrict__ out) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= pairs) return;
  double acc = 0.0;
  for (uint32_t z = 0; z < slabs; ++z) acc += partial[(uint64_t)z * pairs + e];
  out[e] = scale_distance<KIND>(acc, p);
}

template <int KIND>
static int rowwise_block(const double *a, uint32_t r1, const double *b, uint32_t r2, uint32_t n_dims, const double *metric,
                         double p, double *out, hipStream_t st, const double *na = nullptr, const double *nb = nullptr,
                         const uint32_t *gate = nullptr) {
  // A few rows against a very long first operand (the chunks of the large-reference summary: 256-512 query rows against
  // 10^6): tiles of 32 columns x 256 rows, so the long operand is read from HBM ONCE per 256 rows (tiles of 64 x 32 read it
  // eight times for 256 rows: 4.1 GB moved for 2.56) while the short one stays in L2.
  if (r1 >= 65536 && r2 <= 4096 && !(ctx().tune_dbg & 8192)) {
    const uint32_t w = 32, n_cg = 8, n_rg = 32, TJ = 8 * n_rg;
    distance_rowwise_kernel<KIND, false, 8><<<dim3(div_up(r1, w), div_up(r2, TJ)), dim3(256), 0, st>>>(a, w, r1, b, r2, n_dims, metric, p, out, n_cg,
                                                                                                       n_rg, 0, na, nb, gate);
    KPOP_LAUNCH_CHECK();
    return 0;
  }
  // balanced column tiles of 64..127 columns (one tile below 128)
  const uint32_t n_tiles = std::max(1u, r1 / 64);
  const uint32_t w = div_up(r1, n_tiles);
  const uint32_t n_cg = div_up(w, 4);
  // a thread takes 4 columns x TY rows of the tile: TY = 8 reads a quarter fewer LDS words per pair than 4 x 4, and the LDS
  // pipe is as busy as the VALU in this kernel; 4 keeps more row groups when the second operand is short
  const bool tall = r2 >= 8192 && !(ctx().tune_dbg & 4096);
  const uint32_t ty = tall ? 8 : 4;
  const uint32_t n_rg = std::min(256u / n_cg, (uint32_t)kMaxTJ / ty);
  const uint32_t TJ = ty * n_rg;
  const uint32_t rows_per_launch = 65535u * TJ;  // m2 rows ride on grid.y
  for (uint32_t j0 = 0; j0 < r2; j0 += rows_per_launch) {
    const uint32_t nr = std::min(rows_per_launch, r2 - j0);
    if (tall && TJ <= 128 && !(ctx().tune_dbg & 1024))  // (1024: the 256-row staging with two wavefronts a SIMD, for A/B)
      distance_rowwise_kernel<KIND, false, 8, 128><<<dim3(div_up(r1, w), div_up(nr, TJ)), dim3(256), 0, st>>>(
          a, w, r1, b + (uint64_t)j0 * n_dims, nr, n_dims, metric, p, out + (uint64_t)j0 * r1, n_cg, n_rg, 0, na, nb ? nb + j0 : nullptr, gate);
    else if (tall)
      distance_rowwise_kernel<KIND, false, 8><<<dim3(div_up(r1, w), div_up(nr, TJ)), dim3(256), 0, st>>>(
          a, w, r1, b + (uint64_t)j0 * n_dims, nr, n_dims, metric, p, out + (uint64_t)j0 * r1, n_cg, n_rg, 0, na, nb ? nb + j0 : nullptr, gate);
    else
      distance_rowwise_kernel<KIND><<<dim3(div_up(r1, w), div_up(nr, TJ)), dim3(256), 0, st>>>(
          a, w, r1, b + (uint64_t)j0 * n_dims, nr, n_dims, metric, p, out + (uint64_t)j0 * r1, n_cg, n_rg, 0, na, nb ? nb + j0 : nullptr, gate);
    KPOP_LAUNCH_CHECK();
  }
  return 0;
}

// distance_mfma.hip: every pair's distance as a tiled contraction on the f64 matrix cores
bool distance_mfma_applies(int kind, uint32_t r1, uint32_t r2, uint32_t n_dims);
int launch_distance_rowwise_mfma(int kind, const double *a, uint32_t r1, const double *b, uint32_t r2, uint32_t n_dims, const double *metric, double p, double *out,
                                 hipStream_t st, const double *n1, const double *n2, const double *s1, const double *s2);

template <int KIND>
static int rowwise_impl(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                        const double *metric, double p, int normalize, void *work, double *out, hipStream_t st,
                        const double *norms1 = nullptr) {
  if (n_dims >= kLongD) {  // spectral distances: a few rows over millions of k-mers
    const uint32_t slabs = long_slabs(r1, r2, n_dims), slab = slab_dims(n_dims, slabs);
    DistWork w = carve(work, r1, r2, n_dims);
    double *partial = w.b + (uint64_t)r2 * n_dims;  // [slabs][r2][r1], then [r1 + r2][slabs] for the norms
    double *npart = partial + (uint64_t)slabs * r1 * r2;
    const double *a = m1, *b = m2;
    if (normalize) {
      const double *src[2] = {m1, m2};
      double *nrm[2] = {w.n1, w.n2}, *dst[2] = {w.a, w.b};
      const uint32_t rows[2] = {r1, r2};
      for (int o = 0; o < 2; ++o) {
        if (rows[o] > 65535) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "distance_rowwise: more than 65535 rows of %u dimensions", n_dims);
        row_norm_partial_kernel<KIND><<<dim3(slabs, rows[o]), dim3(256), 0, st>>>(src[o], n_dims, slab, metric, p, npart);
        KPOP_LAUNCH_CHECK();
        row_norm_final_kernel<KIND><<<dim3(div_up(rows[o], 256)), dim3(256), 0, st>>>(npart, rows[o], slabs, p, nrm[o]);
        KPOP_LAUNCH_CHECK();
        divide_rows_kernel<<<dim3(std::min<uint32_t>(div_up(n_dims, 256), 1024), rows[o]), dim3(256), 0, st>>>(src[o], n_dims, nrm[o], dst[o]);
        KPOP_LAUNCH_CHECK();
      }
      a = w.a;
      b = w.b;
    }
    const uint32_t n_tiles = std::max(1u, r1 / 64);
    const uint32_t wt = div_up(r1, n_tiles), n_cg = div_up(wt, 4);
    const uint32_t n_rg = std::min(256u / n_cg, (uint32_t)kMaxTJ / 4), TJ = 4 * n_rg;
    if (div_up(r2, TJ) > 65535) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "distance_rowwise: too many rows in the second operand (%u)", r2);
    distance_rowwise_kernel<KIND, true><<<dim3(div_up(r1, wt), div_up(r2, TJ), slabs), dim3(256), 0, st>>>(
        a, wt, r1, b, r2, n_dims, metric, p, partial, n_cg, n_rg, slab);
    KPOP_LAUNCH_CHECK();
    const uint64_t pairs = (uint64_t)r1 * r2;
    reduce_slabs_kernel<KIND><<<dim3(div_up(pairs, 256)), dim3(256), 0, st>>>(partial, pairs, slabs, p, out);
    KPOP_LAUNCH_CHECK();
    return 0;
  }
  // One column tile (r1 < 128: distances to a set of classes): every row is divided about once either way, and dividing while
  // staging saves writing and re-reading a copy of both operands -- 0.144 -> 0.132 ms at 65 x 100,000 x 64, 0.031 -> 0.023 ms at
  // 10 x 100,000 x 9.  With more column tiles every tile divides the second operand's rows again (4,096^2: 5 % slower), so
  // larger first operands keep the copies.  (16384: always the copies, for A/B.)
  if (!normalize || r1 >= 128 || (ctx().tune_dbg & 16384)) {
    // large jobs (the reference's own: 650 K samples x 1,636 classes x 1,635 dimensions, README.md:1054-1060) on the matrix cores: the rows'
    // norms only -- the contraction takes the operands as they are and scales a dot product where it comes out (distance_mfma.hip)
    if (distance_mfma_applies(KIND, r1, r2, n_dims)) {
      if (!normalize) return launch_distance_rowwise_mfma(KIND, m1, r1, m2, r2, n_dims, metric, p, out, st, nullptr, nullptr, nullptr, nullptr);
      // (the norms' pass hands over the rows' sums of squares as well -- the same sums before their scale: the larger operand is read ONCE
      // before the contraction; they go where the normalised copies used to)
      DistWork w = carve(work, r1, r2, n_dims);
      KPOP_TRY(launch_row_norms_pair<KIND>(m1, r1, w.n1, nullptr, m2, r2, w.n2, nullptr, n_dims, metric, p, st, w.a, w.b));
      return launch_distance_rowwise_mfma(KIND, m1, r1, m2, r2, n_dims, metric, p, out, st, norms1 ? norms1 : w.n1, w.n2, w.a, w.b);
    }
    const double *a, *b;
    KPOP_TRY(prepare_operands<KIND>(m1, r1, m2, r2, n_dims, metric, p, normalize, work, &a, &b, st));
    return rowwise_block<KIND>(a, r1, b, r2, n_dims, metric, p, out, st);
  }
  // norms only; the rowwise kernel divides as it stages the rows
  DistWork w = carve(work, r1, r2, n_dims);
  // (a caller that keeps the first operand -- the class vectors of the streaming pipeline -- brings its norms along)
  KPOP_TRY(launch_row_norms_pair<KIND>((r1 && !norms1) ? m1 : nullptr, r1, w.n1, nullptr, r2 ? m2 : nullptr, r2, w.n2, nullptr, n_dims, metric, p, st));
  return rowwise_block<KIND>(m1, r1, m2, r2, n_dims, metric, p, out, st, norms1 ? norms1 : w.n1, w.n2);
}

// summary_large.hip
int launch_summary_large(const double *rows, uint32_t n_rows, uint32_t r1, uint32_t row0, uint32_t keep_at_most,
                         uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx,
                         double *out_dist, double *out_z, hipStream_t st, void *scratch, SummaryLists *lists = nullptr, bool plain_rows = false,
                         const double *srow = nullptr, uint32_t srow_n = 0);
uint64_t summary_large_scratch_bytes(uint32_t n_rows, uint32_t r1);
bool summary_fused_applies(uint32_t r1, uint32_t keep_at_most);
uint32_t summary_fused_sample_rows(uint32_t r1);
uint64_t summary_fused_scratch_bytes(uint32_t n_rows, uint32_t r1);
uint64_t summary_select_scratch_bytes(uint32_t n_rows, uint32_t r1);
int launch_sample_gather(const double *a, uint32_t r1, uint32_t n_dims, uint32_t s, double *out, hipStream_t st);
int launch_summary_fused(int kind, const double *a, uint32_t r1, const double *b, uint32_t n_rows, uint32_t n_dims, const double *metric, double p,
                         const double *srow, uint32_t s, uint32_t row0, uint32_t keep_at_most, uint32_t max_neighbours, double *out_stats,
                         uint32_t *out_n, uint32_t *out_idx, double *out_dist, double *out_z, double *seg, void *scratch, hipStream_t st,
                         const uint32_t **gate);
// distance_mfma.hip: the large-reference summary's distances on the matrix cores, and what makes its results exact again
bool summary_mfma_applies(int kind, uint32_t r1, uint32_t n_dims, uint32_t keep_at_most, uint32_t max_neighbours);
uint64_t summary_mfma_scratch_bytes(uint32_t q, uint32_t r1, uint32_t n_dims);
int launch_mfma_reference_norms(const double *a, uint32_t r1, uint32_t n_dims, const double *metric, void *scratch, uint32_t q_room, hipStream_t st,
                                const double *na = nullptr, const double *s_raw = nullptr);
int launch_mfma_copy_reference_norms(const void *from, void *to, uint32_t r1, uint32_t n_dims, uint32_t q_room, hipStream_t st);
int launch_distance_rows_mfma(int kind, const double *a, uint32_t r1, const double *b, uint32_t q, uint32_t n_dims, const double *metric, double *rows,
                              void *scratch, uint32_t q_room, hipStream_t st, bool a_raw = false);
int launch_summary_refine(int kind, const double *rows, const double *a, uint32_t r1, const double *b, uint32_t q, uint32_t n_dims, const double *metric,
                          double p, uint32_t row0, uint32_t keep_at_most, uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx,
                          double *out_dist, double *out_z, void *scratch, uint32_t q_room, hipStream_t st, const SummaryLists &lists, const uint32_t **gate,
                          const void **row_counts, const double *na = nullptr);
int launch_summary_flagged_rows(const double *rows, uint32_t n_rows, uint32_t r1, uint32_t row0, uint32_t keep_at_most, uint32_t max_neighbours,
                                double *out_stats, uint32_t *out_n, uint32_t *out_idx, double *out_dist, double *out_z, const void *flags, hipStream_t st);
// ... the same without distance rows: the summary's pass inside the contraction (summary_large.hip / distance_mfma.hip)
bool summary_select_mfma_applies(uint32_t r1, uint32_t keep_at_most);
int launch_mfma_query_prep(const double *b, uint32_t q, uint32_t r1, uint32_t n_dims, const double *metric, void *scratch, uint32_t q_room, hipStream_t st);
int launch_rows_mfma_against(int kind, const double *as, const double *sas, uint32_t s, uint32_t q, uint32_t n_dims, double *rows, void *scratch, uint32_t q_room,
                             uint32_t r1, hipStream_t st, const double *ias = nullptr);
int launch_mfma_sample_scalars(const void *scratch, uint32_t q_room, uint32_t r1, uint32_t n_dims, uint32_t s, double *sas, double *ias, hipStream_t st);
int launch_row_sumsq(const double *x, uint32_t rows, uint32_t n_dims, const double *metric, double *out, hipStream_t st);
int launch_summary_fused_mfma(int kind, const double *a, uint32_t r1, uint32_t n_rows, uint32_t n_dims, const double *srow, uint32_t s, uint32_t row0,
                              uint32_t keep_at_most, uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx, double *out_dist,
                              double *out_z, double *seg, uint32_t *seg_i, void *scratch, const void *mscratch, uint32_t q_room, hipStream_t st,
                              SummaryLists *lists);
int launch_summary_failed_rows(const double *rows, uint32_t n_rows, uint32_t r1, uint32_t row0, uint32_t keep_at_most, uint32_t max_neighbours,
                               double *out_stats, uint32_t *out_n, uint32_t *out_idx, double *out_dist, double *out_z, void *scratch,
                               hipStream_t st);

// kpop_tune("summary_audit", 1): the rows a chunk leaves to the exact fall-back are counted (a synchronisation and a 4-byte copy a chunk:
// for tests and probes -- a fall-back that runs when it should not costs 4 ms a chunk and changes no result, so nothing else shows it)
static int audit_fallback(const uint32_t *gate, hipStream_t st) {
  if (!ctx().tune_summary_audit || !gate) return 0;
  uint32_t g = 0;
  KPOP_HIP(hipStreamSynchronize(st));
  KPOP_HIP(hipMemcpy(&g, gate, 4, hipMemcpyDeviceToHost));
  ctx().summary_fallback_rows += g;
  return 0;
}
extern "C" int kpop_debug_summary_fallbacks(uint64_t *rows) {
  KPOP_TRY(require_init());
  if (!rows) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_debug_summary_fallbacks: null argument");
  *rows = ctx().summary_fallback_rows;
  ctx().summary_fallback_rows = 0;
  return KPOP_OK;
}

// r1 > kSummaryMaxR1: query rows in chunks, distance rows of a chunk in the library workspace
template <int KIND>
static int summary_large_impl(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                              const double *metric, double p, int normalize, uint32_t keep_at_most,
                              uint32_t max_neighbours, void *work, double *out_stats, uint32_t *out_n,
                              uint32_t *out_idx, double *out_dist, double *out_z, hipStream_t st) {
  const double *a, *b;
  const uint64_t budget = 4096ull << 20;
  const bool mfma = summary_mfma_applies(KIND, r1, n_dims, keep_at_most, max_neighbours);
  const bool mfma_select = mfma && ctx().tune_summary_mfma == 2 && n_dims <= 128 && summary_select_mfma_applies(r1, keep_at_most);
  // The matrix-core path (its default form) takes the REFERENCE set as it is: no normalised copy of it is made (8.5 GB read and written
  // for 650,000 x 1,635, a quarter of a 256-row call) -- the norms' pass keeps the rows' sums of squares, a dot product is scaled by the
  // norm's reciprocal where it comes out, the exact chains of the refinement and of the fall-back divide element by element as the copy
  // did (the same bits).  The query rows, a few hundred, are divided as before.  kpop_tune("summary_rawref", 0): the copy, as before.
  const double *na = nullptr, *s_raw = nullptr;  // the reference rows' norms and raw sums of squares when `a` is NOT divided
  if (mfma && !mfma_select && normalize && ctx().tune_summary_rawref) {
    DistWork w = carve(work, r1, r2, n_dims);
    KPOP_TRY(launch_row_norms_pair<KIND>(m1, r1, w.n1, nullptr, m2, r2, w.n2, w.b, n_dims, metric, p, st, w.a, nullptr));
    a = m1;
    b = w.b;
    na = w.n1;
    s_raw = w.a;  // (r1 doubles at the head of the room the copy would have taken)
  } else {
    KPOP_TRY(prepare_operands<KIND>(m1, r1, m2, r2, n_dims, metric, p, normalize, work, &a, &b, st));
  }
  if (mfma) {
    // the distances on the matrix cores, approximately, to LOCATE what the summary reports; what is reported is recomputed with the
    // reference's chain (distance_mfma.hip).  Rows the refinement cannot vouch for: exact distance rows and the one-block-per-row
    // kernel over them, both launched whatever happened and both returning at once when nothing was flagged.
    // Up to 128 dimensions (kpop_tune("summary_mfma", 2), the default) no approximate row is WRITTEN either: thresholds from the
    // distances to a sample of the reference rows, then ONE kernel that classifies every distance in the accumulators' registers
    // (summary_select_mfma_kernel); beyond, and under kpop_tune("summary_mfma", 1), round 5's path: rows, then the summary's pass over them.
    const bool select = mfma_select;
    uint32_t chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(r2, 2 * budget / ((uint64_t)r1 * 8)));  // (1,024 rows against a million)
    if (chunk > 128) chunk = chunk / 128 * 128;
    // TWO LANES (kpop_tune("summary_lanes", 2); 512 query rows and more): batches of 256 rows alternately on the caller's stream and on one of
    // the library's own, a lane's contraction waiting for the other lane's.  Meant to run a batch's sample / finish / refinement kernels (one
    // block a row, a block a CU, chains of dependent steps) under the next batch's contraction and pass; measured level with one batch after
    // the other (common.h), which stays the default.  Results do not depend on the lanes: every row is its own computation.
    const bool two_lanes = !select && ctx().tune_summary_lanes == 2 && r2 >= 512 && chunk >= 512;
    if (two_lanes) {
      chunk = 256;
      const uint64_t row_bytes = ((uint64_t)chunk * r1 * 8 + 255) & ~255ull;
      const uint64_t sum_bytes = (summary_large_scratch_bytes(chunk, r1) + 511) & ~255ull, m_bytes = (summary_mfma_scratch_bytes(chunk, r1, n_dims) + 511) & ~255ull;
      const bool row_sample2 = ctx().tune_summary_sample == 1;  // (the brackets from a sample of the reference rows: see the one-lane form below)
      const uint32_t s_rows2 = row_sample2 ? (n_dims > 128 ? std::min(32768u, summary_fused_sample_rows(r1)) : summary_fused_sample_rows(r1)) : 0u;
      const uint64_t srow2_bytes = ((uint64_t)chunk * s_rows2 * 8 + 255) & ~255ull, as2_bytes = ((uint64_t)s_rows2 * n_dims * 8 + 255) & ~255ull,
                     sas2_bytes = ((uint64_t)s_rows2 * 8 + 255) & ~255ull;
      const uint64_t lane_bytes = row_bytes + sum_bytes + m_bytes + srow2_bytes + 256;
      void *ws = nullptr;
      KPOP_TRY(ctx().ws_for(st).ensure(2 * lane_bytes + as2_bytes + 2 * sas2_bytes + 512, &ws));
      Context::AuxLane *aux = nullptr;
      KPOP_TRY(ctx().aux_for(st, &aux));
      char *wp = reinterpret_cast<char *>(ws);
      struct Lane {
        hipStream_t s;
        double *rows;
        void *scratch, *mscratch;
        double *srow;
      } lane[2];
      for (int l = 0; l < 2; ++l) {
        char *base = wp + (uint64_t)l * lane_bytes;
        lane[l] = Lane{l ? aux->stream : st, reinterpret_cast<double *>(base), base + row_bytes, base + row_bytes + sum_bytes,
                       reinterpret_cast<double *>(base + row_bytes + sum_bytes + m_bytes)};
      }
      double *a_s2 = reinterpret_cast<double *>(wp + 2 * lane_bytes), *sa_s2 = reinterpret_cast<double *>(wp + 2 * lane_bytes + as2_bytes),
             *ia_s2 = na ? reinterpret_cast<double *>(wp + 2 * lane_bytes + as2_bytes + sas2_bytes) : nullptr;
      KPOP_TRY(launch_mfma_reference_norms(a, r1, n_dims, metric, lane[0].mscratch, chunk, st, na, s_raw));
      KPOP_TRY(launch_mfma_copy_reference_norms(lane[0].mscratch, lane[1].mscratch, r1, n_dims, chunk, st));
      if (row_sample2) {
        KPOP_TRY(launch_sample_gather(a, r1, n_dims, s_rows2, a_s2, st));
        KPOP_TRY(launch_mfma_sample_scalars(lane[0].mscratch, chunk, r1, n_dims, s_rows2, sa_s2, ia_s2, st));
      }
      KPOP_HIP(hipEventRecord(aux->fork, st));
      KPOP_HIP(hipStreamWaitEvent(aux->stream, aux->fork, 0));
      int rc = 0;
      uint32_t bi = 0;
      for (uint32_t q0 = 0; q0 < r2 && !rc; q0 += chunk, ++bi) {
        const Lane &L = lane[bi & 1u];
        const uint32_t q = std::min(chunk, r2 - q0);
        const double *bq = b + (uint64_t)q0 * n_dims;
        SummaryLists lists;
        const uint32_t *gate = nullptr;
        const void *flags = nullptr;
        auto batch = [&]() -> int {
          if (bi > 0) KPOP_HIP(hipStreamWaitEvent(L.s, aux->step[(bi - 1) & 1u], 0));  // this batch's contraction after the one before it
          KPOP_TRY(launch_distance_rows_mfma(KIND, a, r1, bq, q, n_dims, metric, L.rows, L.mscratch, chunk, L.s, na != nullptr));
          KPOP_HIP(hipEventRecord(aux->step[bi & 1u], L.s));
          if (row_sample2) KPOP_TRY(launch_rows_mfma_against(KIND, a_s2, sa_s2, s_rows2, q, n_dims, L.srow, L.mscratch, chunk, r1, L.s, ia_s2));
          KPOP_TRY(launch_summary_large(L.rows, q, r1, q0, keep_at_most, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z, L.s, L.scratch, &lists, true,
                                        row_sample2 ? L.srow : nullptr, s_rows2));
          KPOP_TRY(launch_summary_refine(KIND, L.rows, a, r1, bq, q, n_dims, metric, p, q0, keep_at_most, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z,
                                         L.mscratch, chunk, L.s, lists, &gate, &flags, na));
          KPOP_TRY(audit_fallback(gate, L.s));
          KPOP_TRY(audit_fallback(lists.n_failed, L.s));
          KPOP_TRY(rowwise_block<KIND>(a, r1, bq, q, n_dims, metric, p, L.rows, L.s, na, nullptr, gate));  // (na: the kernel divides the reference rows as it stages them)
          KPOP_TRY(launch_summary_flagged_rows(L.rows, q, r1, q0, keep_at_most, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z, flags, L.s));
          return 0;
        };
        rc = batch();
      }
      // the join, whatever happened: the caller's stream goes on only when the library's own has drained
      const hipError_t e1 = hipEventRecord(aux->join, aux->stream), e2 = hipStreamWaitEvent(st, aux->join, 0);
      if (rc) return rc;
      if (e1 != hipSuccess || e2 != hipSuccess) KPOP_FAIL(KPOP_ERR_HIP, "kpop_dev_distance_summary: joining the second lane: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
      return 0;
    }
    void *ws = nullptr;
    // the brackets and bands from the query rows' distances to a SAMPLE OF THE REFERENCE ROWS (even spacing), whatever the layout of the database
    // (launch_summary_large); kpop_tune("summary_sample", 0): from runs of the distance rows, as until late in round 6
    const bool row_sample = !select && ctx().tune_summary_sample == 1;
    // (beyond 128 dimensions half the sample: its contraction is s / r1 of the main one -- a tenth for 650,000 rows -- while the candidates its
    // wider brackets add, 12 -> 17 % of a row, cost little beside a contraction of 1,635 dimensions)
    const uint32_t s_rows = select ? summary_fused_sample_rows(r1) : row_sample ? (n_dims > 128 ? std::min(32768u, summary_fused_sample_rows(r1)) : summary_fused_sample_rows(r1)) : 0;
    const uint64_t r1_seg = select ? (((uint64_t)r1 + 2047) & ~2047ull) : r1;  // (the select kernel's segments: a row rounded up to whole stripes of 2,048)
    const uint64_t row_bytes = ((uint64_t)chunk * r1_seg * 8 + 255) & ~255ull, segi_bytes = select ? (((uint64_t)chunk * r1_seg * 4 + 255) & ~255ull) : 0;
    const uint64_t sum_bytes = (std::max(summary_large_scratch_bytes(chunk, r1), select ? summary_select_scratch_bytes(chunk, r1) : 0) + 511) & ~255ull;
    const uint64_t m_bytes = (summary_mfma_scratch_bytes(chunk, r1, n_dims) + 511) & ~255ull;
    const uint64_t as_bytes = ((uint64_t)s_rows * n_dims * 8 + 255) & ~255ull, sas_bytes = 2 * (((uint64_t)s_rows * 8 + 255) & ~255ull),  // (sums of squares, reciprocals)
                   srow_bytes = ((uint64_t)chunk * s_rows * 8 + 255) & ~255ull;
    KPOP_TRY(ctx().ws_for(st).ensure(row_bytes + segi_bytes + sum_bytes + m_bytes + as_bytes + sas_bytes + srow_bytes + 512, &ws));
    char *wp = reinterpret_cast<char *>(ws);
    double *rows = reinterpret_cast<double *>(wp);
    uint32_t *seg_i = reinterpret_cast<uint32_t *>(wp + row_bytes);
    void *scratch = wp + row_bytes + segi_bytes, *mscratch = wp + row_bytes + segi_bytes + sum_bytes;
    double *a_s = reinterpret_cast<double *>(wp + row_bytes + segi_bytes + sum_bytes + m_bytes);
    double *sa_s = reinterpret_cast<double *>(wp + row_bytes + segi_bytes + sum_bytes + m_bytes + as_bytes);
    double *srow = reinterpret_cast<double *>(wp + row_bytes + segi_bytes + sum_bytes + m_bytes + as_bytes + sas_bytes);
    KPOP_TRY(launch_mfma_reference_norms(a, r1, n_dims, metric, mscratch, chunk, st, na, s_raw));
    double *ia_s = nullptr;
    if (row_sample) {  // the sample of the reference rows, its sums of squares and (the set taken as it is) its norms' reciprocals: once a call
      KPOP_TRY(launch_sample_gather(a, r1, n_dims, s_rows, a_s, st));
      if (na) ia_s = sa_s + (sas_bytes / 16);
      KPOP_TRY(launch_mfma_sample_scalars(mscratch, chunk, r1, n_dims, s_rows, sa_s, ia_s, st));
    }
    if (select) {  // the sample of the reference rows and its norms: once a call
      KPOP_TRY(launch_sample_gather(a, r1, n_dims, s_rows, a_s, st));
      KPOP_TRY(launch_row_sumsq(a_s, s_rows, n_dims, metric, sa_s, st));
    }
    for (uint32_t q0 = 0; q0 < r2; q0 += chunk) {
      const uint32_t q = std::min(chunk, r2 - q0);
      const double *bq = b + (uint64_t)q0 * n_dims;
      SummaryLists lists;
      const uint32_t *gate = nullptr;
      const void *flags = nullptr;
      if (select) {
        KPOP_TRY(launch_mfma_query_prep(bq, q, r1, n_dims, metric, mscratch, chunk, st));
        KPOP_TRY(launch_rows_mfma_against(KIND, a_s, sa_s, s_rows, q, n_dims, srow, mscratch, chunk, r1, st));
        KPOP_TRY(launch_summary_fused_mfma(KIND, a, r1, q, n_dims, srow, s_rows, q0, keep_at_most, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z, rows,
                                           seg_i, scratch, mscratch, chunk, st, &lists));
        KPOP_TRY(launch_summary_refine(KIND, nullptr, a, r1, bq, q, n_dims, metric, p, q0, keep_at_most, max_neighbours, out_stats, out_n, out_idx, out_dist,
                                       out_z, mscratch, chunk, st, lists, &gate, &flags));
        KPOP_TRY(audit_fallback(gate, st));
      } else {
        KPOP_TRY(launch_distance_rows_mfma(KIND, a, r1, bq, q, n_dims, metric, rows, mscratch, chunk, st, na != nullptr));
        if (row_sample) KPOP_TRY(launch_rows_mfma_against(KIND, a_s, sa_s, s_rows, q, n_dims, srow, mscratch, chunk, r1, st, ia_s));  // (the chunk's query rows are prepared: the call above)
        KPOP_TRY(launch_summary_large(rows, q, r1, q0, keep_at_most, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z, st, scratch, &lists, true,
                                      row_sample ? srow : nullptr, s_rows));
        KPOP_TRY(launch_summary_refine(KIND, rows, a, r1, bq, q, n_dims, metric, p, q0, keep_at_most, max_neighbours, out_stats, out_n, out_idx, out_dist,
                                       out_z, mscratch, chunk, st, lists, &gate, &flags, na));
        KPOP_TRY(audit_fallback(gate, st));
        KPOP_TRY(audit_fallback(lists.n_failed, st));  // (rows whose sample-based brackets missed: the ten-pass kernel's)
      }
      KPOP_TRY(rowwise_block<KIND>(a, r1, bq, q, n_dims, metric, p, rows, st, na, nullptr, gate));  // (na: the kernel divides the reference rows as it stages them)
      KPOP_TRY(launch_summary_flagged_rows(rows, q, r1, q0, keep_at_most, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z, flags, st));
    }
    return 0;
  }
  if (summary_fused_applies(r1, keep_at_most)) {
    // 131,072 reference rows and more: no distance rows at all (summary_large.hip, "second step").  The workspace holds, per
    // chunk of query rows, the candidates' segments (the same room distance rows would take, mostly untouched), the sample of
    // the first operand and the distances to it, and the lists.
    const uint32_t s = summary_fused_sample_rows(r1);
    const uint64_t fixed = (((uint64_t)s * n_dims * 8 + 255) & ~255ull) + (1u << 20);
    uint32_t chunk = r2;
    while (chunk > 1 && fixed + (uint64_t)chunk * ((uint64_t)r1 + s) * 8 + summary_fused_scratch_bytes(chunk, r1) > budget) chunk = chunk > 256 ? (chunk - 1) / 256 * 256 : chunk / 2;
    const uint64_t bytes_seg = ((uint64_t)chunk * r1 * 8 + 255) & ~255ull, bytes_as = ((uint64_t)s * n_dims * 8 + 255) & ~255ull,
                   bytes_srow = ((uint64_t)chunk * s * 8 + 255) & ~255ull;
    void *ws = nullptr;
    KPOP_TRY(ctx().ws_for(st).ensure(bytes_seg + bytes_as + bytes_srow + summary_fused_scratch_bytes(chunk, r1), &ws));
    char *wp = reinterpret_cast<char *>(ws);
    double *seg = reinterpret_cast<double *>(wp), *a_s = reinterpret_cast<double *>(wp + bytes_seg),
           *srow = reinterpret_cast<double *>(wp + bytes_seg + bytes_as);
    void *scratch = wp + bytes_seg + bytes_as + bytes_srow;
    KPOP_TRY(launch_sample_gather(a, r1, n_dims, s, a_s, st));
    for (uint32_t q0 = 0; q0 < r2; q0 += chunk) {
      const uint32_t q = std::min(chunk, r2 - q0);
      const double *bq = b + (uint64_t)q0 * n_dims;
      KPOP_TRY(rowwise_block<KIND>(a_s, s, bq, q, n_dims, metric, p, srow, st));
      const uint32_t *gate = nullptr;
      KPOP_TRY(launch_summary_fused(KIND, a, r1, bq, q, n_dims, metric, p, srow, s, q0, keep_at_most, max_neighbours, out_stats, out_n, out_idx,
                                    out_dist, out_z, seg, scratch, st, &gate));
      KPOP_TRY(audit_fallback(gate, st));
      // the rows it flagged (a bracket that missed, a list that overflowed): their distance rows into the segments' room and the
      // one-block-per-row kernel over them -- both launched whatever happened, both return at once when nothing was flagged
      KPOP_TRY(rowwise_block<KIND>(a, r1, bq, q, n_dims, metric, p, seg, st, nullptr, nullptr, gate));
      KPOP_TRY(launch_summary_failed_rows(seg, q, r1, q0, keep_at_most, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z, scratch, st));
    }
    return 0;
  }
  // distance rows of a chunk of queries live in the library workspace: enough of them (one 1024-thread block each) to
  // put two blocks on every CU when the first operand is large
  const uint32_t chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(r2, budget / ((uint64_t)r1 * 8)));
  void *ws = nullptr;
  const uint64_t row_bytes = ((uint64_t)chunk * r1 * 8 + 255) & ~255ull;
  // (262,144 reference rows and more: the brackets and bands from the query rows' distances to a sample of the REFERENCE ROWS at even spacing,
  // as in the matrix-core form above -- the same chain for every pair, a sixteenth of the distances again or less; runs of the distance rows
  // speak for a few lineages of a database laid out lineage by lineage.  kpop_tune("summary_sample", 0): the runs)
  const uint32_t s_rows = (ctx().tune_summary_sample == 1 && ctx().tune_summary2 == 1 && r1 >= 262144) ? summary_fused_sample_rows(r1) : 0;
  const uint64_t as_bytes = ((uint64_t)s_rows * n_dims * 8 + 255) & ~255ull, srow_bytes = ((uint64_t)chunk * s_rows * 8 + 255) & ~255ull;
  const uint64_t sum_bytes = (summary_large_scratch_bytes(chunk, r1) + 511) & ~255ull;
  KPOP_TRY(ctx().ws_for(st).ensure(row_bytes + sum_bytes + as_bytes + srow_bytes + 256, &ws));
  double *rows = reinterpret_cast<double *>(ws);
  void *scratch = reinterpret_cast<char *>(ws) + row_bytes;
  double *a_s = reinterpret_cast<double *>(reinterpret_cast<char *>(ws) + row_bytes + sum_bytes);
  double *srow = reinterpret_cast<double *>(reinterpret_cast<char *>(ws) + row_bytes + sum_bytes + as_bytes);
  if (s_rows) KPOP_TRY(launch_sample_gather(a, r1, n_dims, s_rows, a_s, st));
  for (uint32_t q0 = 0; q0 < r2; q0 += chunk) {
    const uint32_t q = std::min(chunk, r2 - q0);
    KPOP_TRY(rowwise_block<KIND>(a, r1, b + (uint64_t)q0 * n_dims, q, n_dims, metric, p, rows, st));
    if (s_rows) KPOP_TRY(rowwise_block<KIND>(a_s, s_rows, b + (uint64_t)q0 * n_dims, q, n_dims, metric, p, srow, st));
    KPOP_TRY(launch_summary_large(rows, q, r1, q0, keep_at_most, max_neighbours, out_stats, out_n, out_idx, out_dist,
                                  out_z, st, scratch, nullptr, false, s_rows ? srow : nullptr, s_rows));
  }
  return 0;
}

template <int KIND, bool PRE>
static int launch_summary(const double *a, uint32_t r1, const double *b, uint32_t r2, uint32_t n_dims,
                          const double *metric, double p, uint32_t keep_at_most, uint32_t max_neighbours,
                          double *out_stats, uint32_t *out_n, uint32_t *out_idx, double *out_dist, double *out_z,
                          hipStream_t st, const double *nb = nullptr) {  // nb: see summary_divides_in_flight
  if (summary_fits_wave(r1, n_dims, PRE) && ctx().tune_dbg != 4)
    return launch_summary_wave<KIND, PRE>(a, r1, b, r2, n_dims, metric, p, keep_at_most ? keep_at_most : r1, max_neighbours, out_stats,
                                          out_n, out_idx, out_dist, out_z, st, nb);
  uint32_t NP = 64;
  while (NP < r1) NP <<= 1;
  const size_t smem = (size_t)NP * (8 + 8 + 4);
  static PerSlotOnce attr_once;
  bool &attr_set = attr_once();
  if (!attr_set) {
    KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&distance_summary_kernel<KIND, PRE>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kSummaryMaxR1 * 20)));
    attr_set = true;
  }
  const uint32_t req_len = keep_at_most ? keep_at_most : r1;  // lib/Matrix.ml:723-726,774-777
  // (a row of 4,096 slots fills the LDS of a CU by itself: 1,024 threads on its sort -- 98 -> 64 ms for 100,000 rows against
  // 4,000; below that several 256-thread blocks share a CU and more threads a block measured slower)
  const uint32_t threads = NP >= 4096 ? 1024u : 256u;
  distance_summary_kernel<KIND, PRE><<<dim3(r2), dim3(threads), smem, st>>>(a, r1, b, r2, n_dims, metric, p, NP, req_len,
                                                                        max_neighbours, out_stats, out_n, out_idx,
                                                                        out_dist, out_z);
  KPOP_LAUNCH_CHECK();
  return 0;
}

template <int KIND>
static int summary_impl(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                        const double *metric, double p, int normalize, uint32_t keep_at_most, uint32_t max_neighbours,
                        void *work, double *out_stats, uint32_t *out_n, uint32_t *out_idx, double *out_dist,
                        double *out_z, hipStream_t st) {
  const double *a, *b;
  if (normalize && r2 && summary_fits_wave(r1, n_dims, false) && ctx().tune_dbg != 4 && summary_batch_case(r1, n_dims, false)) {
    // The kernel that takes four rows of a wavefront at a time stages the second operand's rows itself: they go in as they are,
    // with their norms, and are divided on the way (the same quotients); only the few rows of the first operand get a divided copy.
    DistWork w = carve(work, r1, r2, n_dims);
    KPOP_TRY(launch_row_norms_pair<KIND>(m1, r1, w.n1, w.a, m2, r2, w.n2, nullptr, n_dims, metric, p, st));
    return launch_summary<KIND, false>(w.a, r1, m2, r2, n_dims, metric, p, keep_at_most, max_neighbours, out_stats, out_n, out_idx, out_dist,
                                       out_z, st, w.n2);
  }
  KPOP_TRY(prepare_operands<KIND>(m1, r1, m2, r2, n_dims, metric, p, normalize, work, &a, &b, st));
  if (r1 >= 1 && r1 <= kSummaryMaxR1 && !summary_fits_wave(r1, n_dims, false) && ctx().tune_dbg != 4) {
    // A first operand that does not fit LDS (100 x 200 dimensions, 500 x 64, anything of 513..4,096 rows): the distances of
    // a chunk of second-operand rows into the workspace (the tiled kernel), then the summary over them -- one wavefront per
    // row up to 512 columns, one block per row sorting in LDS above.  Round 2's block kernel worked its distances out
    // itself, every block reading the whole first operand: 100,000 rows against 1,000 / 4,000 took 23 / 320 ms.
    const uint32_t chunk = (uint32_t)std::min<uint64_t>(r2, std::max<uint64_t>(1024, (256ull << 20) / ((uint64_t)r1 * 8)));
    void *ws = nullptr;
    KPOP_TRY(ctx().ws_for(st).ensure((uint64_t)chunk * r1 * 8, &ws));
    double *rows = reinterpret_cast<double *>(ws);
    for (uint32_t q0 = 0; q0 < r2; q0 += chunk) {
      const uint32_t q = std::min(chunk, r2 - q0);
      KPOP_TRY(rowwise_block<KIND>(a, r1, b + (uint64_t)q0 * n_dims, q, n_dims, metric, p, rows, st));
      KPOP_TRY((launch_summary<KIND, true>(rows, r1, nullptr, q, n_dims, metric, p, keep_at_most, max_neighbours, out_stats + (uint64_t)q0 * 4,
                                           out_n + q0, out_idx + (uint64_t)q0 * max_neighbours, out_dist + (uint64_t)q0 * max_neighbours,
                                           out_z + (uint64_t)q0 * max_neighbours, st)));
    }
    return 0;
  }
  return launch_summary<KIND, false>(a, r1, b, r2, n_dims, metric, p, keep_at_most, max_neighbours, out_stats, out_n,
                                     out_idx, out_dist, out_z, st);
}

// out[i][c] = m[i][c] * w[c]  (Base.get_embeddings, lib/Matrix.ml:104)
__global__ __launch_bounds__(256) void scale_columns_kernel(const double *__restrict__ m, uint64_t total, uint32_t n_dims,
                                                            const double *__restrict__ w, double *__restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) out[e] = __dmul_rn(m[e], w[e % n_dims]);
}

template <int KIND>
static int embeddings_impl(const double *d_m, uint32_t rows, uint32_t n_dims, const double *d_metric, const double *d_w, double p,
                           int normalize, double *d_scaled, double *d_norms, double *d_out, hipStream_t st) {
  const uint64_t total = (uint64_t)rows * n_dims;
  scale_columns_kernel<<<dim3(capped_grid(div_up(total, 256))), dim3(256), 0, st>>>(d_m, total, n_dims, d_w, normalize ? d_scaled : d_out);
  KPOP_LAUNCH_CHECK();
  if (normalize) {  // v / norm(v) with the 0 -> 1 rule standing in for `if norm <> 0.` (lib/Matrix.ml:105-109)
    row_norms_kernel<KIND><<<dim3(div_up(rows, kNormRows)), dim3(256), 0, st>>>(d_scaled, rows, n_dims, d_metric, p, d_norms, d_out);
    KPOP_LAUNCH_CHECK();
  }
  return 0;
}

// Against more than kSummaryMaxR1 rows the summary kernels return at most kLargeNeighbours neighbours per row
// (summary_large.hip); out_n is the reference's eff_len whatever its size.  A row whose list is longer
// (--summary-keep-at-most all: req_len = r1, lib/Matrix.ml:723-726; or a tie group of thousands, :648-649) is completed
// here, by the host entry points: the row's distances -- given, or computed by the kernel the summary itself uses -- are
// sorted by (distance, column) with the device-wide radix sort (stable, the columns travelling as values: equal distances
// keep column order, which is the multimap's, :641-650), and the first eff_len written over the row's stride with their
// z-scores.  One row at a time: long lists are the rare request, and a row of a million distances is 24 small launches.
constexpr uint32_t kLargeNeighbours = 2048;
__global__ void long_list_keys_kernel(const double *__restrict__ d, uint32_t r1, uint64_t *__restrict__ key, uint32_t *__restrict__ val) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < r1) {
    const double x = d[i];
    key[i] = dist_key(x == 0.0 ? 0.0 : x);  // the kernels' own order (-0 as +0): a caller's matrix may hold negative entries or NaNs
    val[i] = i;
  }
}

struct LongListSource {  // where a row's distances come from
  const double *d_rows = nullptr;  // r2 x r1, given (kpop_summarize_distances) ...
  const double *d1 = nullptr, *d2 = nullptr, *dm = nullptr;  // ... or the operands (kpop_distance_summary)
  uint32_t n_dims = 0;
  int kind = 0, normalize = 0;
  double p = 2.0;
};

static int fill_long_lists(const LongListSource &src, uint32_t r1, uint32_t r2, uint32_t max_neighbours, const double *out_stats, const uint32_t *out_n,
                           uint32_t *out_idx, double *out_dist, double *out_z, hipStream_t st) {
  if (r1 <= kSummaryMaxR1 || max_neighbours <= kLargeNeighbours) return KPOP_OK;
  std::vector<uint32_t> todo;
  for (uint32_t j = 0; j < r2; ++j)
    if (out_n[j] > kLargeNeighbours) todo.push_back(j);
  if (todo.empty()) return KPOP_OK;
  if (!out_idx || !out_dist || !out_z) KPOP_FAIL(KPOP_ERR_INVALID, "summary: neighbour outputs are null");
  DevBuf drow, dw, ka, kb, va, vb, scr;
  if (!src.d_rows) {
    KPOP_TRY(drow.alloc((uint64_t)r1 * 8));
    KPOP_TRY(dw.alloc(kpop_dev_distance_workspace_bytes(r1, 1, src.n_dims)));
  }
  KPOP_TRY(ka.alloc((uint64_t)r1 * 8));
  KPOP_TRY(kb.alloc((uint64_t)r1 * 8));
  KPOP_TRY(va.alloc((uint64_t)r1 * 4));
  KPOP_TRY(vb.alloc((uint64_t)r1 * 4));
  KPOP_TRY(scr.alloc(radix_scratch_bytes(r1)));
  std::vector<uint64_t> hk;
  for (uint32_t j : todo) {
    const double *row = src.d_rows ? src.d_rows + (uint64_t)j * r1 : drow.as<double>();
    if (!src.d_rows)
      KPOP_TRY(kpop_dev_distance_rowwise(src.d1, r1, src.d2 + (uint64_t)j * src.n_dims, 1, src.n_dims, src.dm, src.kind, src.p, src.normalize, dw.p,
                                         drow.as<double>(), st));
    long_list_keys_kernel<<<dim3(div_up(r1, 256)), dim3(256), 0, st>>>(row, r1, ka.as<uint64_t>(), va.as<uint32_t>());
    KPOP_LAUNCH_CHECK();
    uint64_t *sk = nullptr;
    uint32_t *sv = nullptr;
    KPOP_TRY(radix_sort_pairs_u64(ka.as<uint64_t>(), kb.as<uint64_t>(), va.as<uint32_t>(), vb.as<uint32_t>(), r1, 64, scr.p, st, &sk, &sv));
    const uint32_t m = std::min(out_n[j], max_neighbours);
    hk.resize(m);
    KPOP_HIP(hipMemcpyAsync(hk.data(), sk, (uint64_t)m * 8, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipMemcpyAsync(out_idx + (uint64_t)j * max_neighbours, sv, (uint64_t)m * 4, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipStreamSynchronize(st));
    const double mean = out_stats[(uint64_t)j * 4 + 0], sd = out_stats[(uint64_t)j * 4 + 1];
    for (uint32_t q = 0; q < m; ++q) {
      const uint64_t kq = hk[q], bq = (kq >> 63) ? (kq & 0x7FFFFFFFFFFFFFFFull) : ~kq;  // key_dist on the host
      double dq;
      memcpy(&dq, &bq, 8);
      out_dist[(uint64_t)j * max_neighbours + q] = dq;
      volatile double num = dq - mean;  // (two roundings, as the kernels': no contraction)
      double zz = num / sd;
      if (zz != zz) {  // the x86 invalid-operation NaN the reference's arithmetic gives (see the kernels)
        const uint64_t nanbits = 0xFFF8000000000000ull;
        memcpy(&zz, &nanbits, 8);
      }
      out_z[(uint64_t)j * max_neighbours + q] = zz;
    }
  }
  return KPOP_OK;
}

static int check_kind(int kind, double p, const char *who) {
  if (kind != KPOP_EUCLIDEAN && kind != KPOP_COSINE && kind != KPOP_MINKOWSKI)
    KPOP_FAIL(KPOP_ERR_INVALID, "%s: unknown distance kind %d", who, kind);
  if (kind == KPOP_MINKOWSKI && !(p >= 0.0)) KPOP_FAIL(KPOP_ERR_INVALID, "%s: negative Minkowski power", who);  // lib/Space.ml:222-223
  return 0;
}

}  // namespace kpop

using namespace kpop;

// Base.get_embeddings, lib/Matrix.ml:78-128: rows scaled by metric ** (1/2 | 1/p), then divided by their norm
extern "C" int kpop_embeddings(const double *m, uint32_t rows, uint32_t n_dims, const double *metric, int kind, double p,
                               int normalize, double *out) {
  KPOP_TRY(require_init());
  ArenaScope scratch;
  KPOP_TRY(check_kind(kind, p, "kpop_embeddings"));
  if (rows == 0 || n_dims == 0) return KPOP_OK;
  if (!m || !metric || !out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_embeddings: null argument");
  const double inv_power = kind == KPOP_MINKOWSKI ? 1. / p : 0.5;  // :84-87
  std::vector<double> w(n_dims);
  for (uint32_t c = 0; c < n_dims; ++c) w[c] = pow(metric[c], inv_power);  // host pow: the libm the reference's ** calls
  hipStream_t st = nullptr;
  const uint64_t bytes = (uint64_t)rows * n_dims * 8;
  DevBuf dm, dmet, dw, ds, dn, dout;
  KPOP_TRY(dm.alloc(bytes));
  KPOP_TRY(dmet.alloc((uint64_t)n_dims * 8));
  KPOP_TRY(dw.alloc((uint64_t)n_dims * 8));
  KPOP_TRY(ds.alloc(normalize ? bytes : 8));
  KPOP_TRY(dn.alloc((uint64_t)rows * 8));
  KPOP_TRY(dout.alloc(bytes));
  KPOP_HIP(hipMemcpyAsync(dm.p, m, bytes, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(dmet.p, metric, (uint64_t)n_dims * 8, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(dw.p, w.data(), (uint64_t)n_dims * 8, hipMemcpyHostToDevice, st));
  int rc;
  if (kind == KPOP_EUCLIDEAN)
    rc = embeddings_impl<KPOP_EUCLIDEAN>(dm.as<double>(), rows, n_dims, dmet.as<double>(), dw.as<double>(), p, normalize, ds.as<double>(),
                                         dn.as<double>(), dout.as<double>(), st);
  else if (kind == KPOP_COSINE)
    rc = embeddings_impl<KPOP_COSINE>(dm.as<double>(), rows, n_dims, dmet.as<double>(), dw.as<double>(), p, normalize, ds.as<double>(),
                                      dn.as<double>(), dout.as<double>(), st);
  else
    rc = embeddings_impl<KPOP_MINKOWSKI>(dm.as<double>(), rows, n_dims, dmet.as<double>(), dw.as<double>(), p, normalize, ds.as<double>(),
                                         dn.as<double>(), dout.as<double>(), st);
  KPOP_TRY(rc);
  KPOP_HIP(hipMemcpyAsync(out, dout.p, bytes, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  return KPOP_OK;
}

extern "C" uint64_t kpop_dev_distance_workspace_bytes(uint32_t r1, uint32_t r2, uint32_t n_dims) {
  uint64_t doubles = (uint64_t)r1 + r2 + ((uint64_t)r1 + r2) * n_dims;
  if (n_dims >= kLongD) {  // slab partials of the distances and of the norms
    const uint64_t slabs = long_slabs(r1, r2, n_dims);
    doubles += slabs * r1 * r2 + slabs * std::max(r1, r2);
  }
  return doubles * sizeof(double) + 64;
}

// Base.get_normalizations, lib/Matrix.ml:42-76 (Space.compute_norm, lib/Space.ml:166-181): norms[i] = scale(sum_c m_c g(a_ic)), 0 -> 1
extern "C" int kpop_dev_row_norms(const double *d_m, uint32_t rows, uint32_t n_dims, const double *d_metric, int kind, double p,
                                  double *d_norms, void *stream) {
  KPOP_TRY(require_init());
  KPOP_TRY(check_kind(kind, p, "kpop_dev_row_norms"));
  if (rows == 0) return KPOP_OK;
  if (!d_m || !d_metric || !d_norms || n_dims == 0) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_row_norms: null argument");
  if (n_dims >= kLongD) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_dev_row_norms: rows of %u dimensions or more go through kpop_dev_distance_rowwise", kLongD);
  hipStream_t st = as_stream(stream);
  const dim3 grid(div_up(rows, kNormRows)), block(256);
  switch (kind) {
    case KPOP_EUCLIDEAN: row_norms_kernel<KPOP_EUCLIDEAN><<<grid, block, 0, st>>>(d_m, rows, n_dims, d_metric, p, d_norms, nullptr); break;
    case KPOP_COSINE: row_norms_kernel<KPOP_COSINE><<<grid, block, 0, st>>>(d_m, rows, n_dims, d_metric, p, d_norms, nullptr); break;
    default: row_norms_kernel<KPOP_MINKOWSKI><<<grid, block, 0, st>>>(d_m, rows, n_dims, d_metric, p, d_norms, nullptr); break;
  }
  KPOP_LAUNCH_CHECK();
  return KPOP_OK;
}

extern "C" int kpop_dev_distance_rowwise(const double *d_m1, uint32_t r1, const double *d_m2, uint32_t r2,
                                         uint32_t n_dims, const double *d_metric, int kind, double p, int normalize,
                                         void *d_work, double *d_out, void *stream) {
  return kpop_dev_distance_rowwise_norms(d_m1, r1, nullptr, d_m2, r2, n_dims, d_metric, kind, p, normalize, d_work, d_out, stream);
}

extern "C" int kpop_dev_distance_rowwise_norms(const double *d_m1, uint32_t r1, const double *d_norms1, const double *d_m2, uint32_t r2,
                                               uint32_t n_dims, const double *d_metric, int kind, double p, int normalize,
                                               void *d_work, double *d_out, void *stream) {
  KPOP_TRY(require_init());
  KPOP_TRY(check_kind(kind, p, "kpop_dev_distance_rowwise"));
  if (r1 == 0 || r2 == 0) return KPOP_OK;
  if (!d_m1 || !d_m2 || !d_metric || !d_out || ((normalize || n_dims >= kLongD) && !d_work))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_distance_rowwise: null argument (the workspace is needed when normalising and for rows of %u dimensions or more)", kLongD);
  if (n_dims == 0) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_distance_rowwise: n_dims must be positive");
  hipStream_t st = as_stream(stream);
  switch (kind) {
    case KPOP_EUCLIDEAN: return rowwise_impl<KPOP_EUCLIDEAN>(d_m1, r1, d_m2, r2, n_dims, d_metric, p, normalize, d_work, d_out, st, d_norms1);
    case KPOP_COSINE: return rowwise_impl<KPOP_COSINE>(d_m1, r1, d_m2, r2, n_dims, d_metric, p, normalize, d_work, d_out, st, d_norms1);
    default: return rowwise_impl<KPOP_MINKOWSKI>(d_m1, r1, d_m2, r2, n_dims, d_metric, p, normalize, d_work, d_out, st, d_norms1);
  }
}

extern "C" int kpop_dev_distance_summary(const double *d_m1, uint32_t r1, const double *d_m2, uint32_t r2,
                                         uint32_t n_dims, const double *d_metric, int kind, double p, int normalize,
                                         uint32_t keep_at_most, uint32_t max_neighbours, void *d_work,
                                         double *d_out_stats, uint32_t *d_out_n, uint32_t *d_out_idx,
                                         double *d_out_dist, double *d_out_z, void *stream) {
  KPOP_TRY(require_init());
  KPOP_TRY(check_kind(kind, p, "kpop_dev_distance_summary"));
  if (r2 == 0) return KPOP_OK;
  if (!d_m2 || !d_metric || !d_out_stats || !d_out_n || (normalize && !d_work) || (r1 && !d_m1))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_distance_summary: null argument");
  if (max_neighbours && (!d_out_idx || !d_out_dist || !d_out_z))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_distance_summary: null neighbour buffers");
  if (n_dims == 0) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_distance_summary: n_dims must be positive");
  hipStream_t st = as_stream(stream);
  if (r1 > kSummaryMaxR1) {
    switch (kind) {
      case KPOP_EUCLIDEAN:
        return summary_large_impl<KPOP_EUCLIDEAN>(d_m1, r1, d_m2, r2, n_dims, d_metric, p, normalize, keep_at_most,
                                                  max_neighbours, d_work, d_out_stats, d_out_n, d_out_idx, d_out_dist, d_out_z, st);
      case KPOP_COSINE:
        return summary_large_impl<KPOP_COSINE>(d_m1, r1, d_m2, r2, n_dims, d_metric, p, normalize, keep_at_most,
                                               max_neighbours, d_work, d_out_stats, d_out_n, d_out_idx, d_out_dist, d_out_z, st);
      default:
        return summary_large_impl<KPOP_MINKOWSKI>(d_m1, r1, d_m2, r2, n_dims, d_metric, p, normalize, keep_at_most,
                                                  max_neighbours, d_work, d_out_stats, d_out_n, d_out_idx, d_out_dist, d_out_z, st);
    }
  }
  switch (kind) {
    case KPOP_EUCLIDEAN:
      return summary_impl<KPOP_EUCLIDEAN>(d_m1, r1, d_m2, r2, n_dims, d_metric, p, normalize, keep_at_most, max_neighbours,
                                          d_work, d_out_stats, d_out_n, d_out_idx, d_out_dist, d_out_z, st);
    case KPOP_COSINE:
      return summary_impl<KPOP_COSINE>(d_m1, r1, d_m2, r2, n_dims, d_metric, p, normalize, keep_at_most, max_neighbours,
                                       d_work, d_out_stats, d_out_n, d_out_idx, d_out_dist, d_out_z, st);
    default:
      return summary_impl<KPOP_MINKOWSKI>(d_m1, r1, d_m2, r2, n_dims, d_metric, p, normalize, keep_at_most, max_neighbours,
                                          d_work, d_out_stats, d_out_n, d_out_idx, d_out_dist, d_out_z, st);
  }
}

extern "C" int kpop_dev_summarize_distances(const double *d_dist, uint32_t r2, uint32_t r1, uint32_t keep_at_most,
                                            uint32_t max_neighbours, double *d_out_stats, uint32_t *d_out_n,
                                            uint32_t *d_out_idx, double *d_out_dist, double *d_out_z, void *stream) {
  KPOP_TRY(require_init());
  if (r2 == 0) return KPOP_OK;
  if (!d_out_stats || !d_out_n || (r1 && !d_dist)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_summarize_distances: null argument");
  if (max_neighbours && (!d_out_idx || !d_out_dist || !d_out_z))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_summarize_distances: null neighbour buffers");
  if (r1 > kSummaryMaxR1) {
    // rows in chunks of up to 512, so that the two-pass path's lists stay a few hundred megabytes
    hipStream_t st = as_stream(stream);
    const uint32_t chunk = std::min<uint32_t>(r2, 512);
    void *scratch = nullptr;
    KPOP_TRY(ctx().ws_for(st).ensure(summary_large_scratch_bytes(chunk, r1), &scratch));
    for (uint32_t q0 = 0; q0 < r2; q0 += chunk)
      KPOP_TRY(launch_summary_large(d_dist + (uint64_t)q0 * r1, std::min(chunk, r2 - q0), r1, q0, keep_at_most, max_neighbours, d_out_stats,
                                    d_out_n, d_out_idx, d_out_dist, d_out_z, st, scratch));
    return KPOP_OK;
  }
  return launch_summary<KPOP_EUCLIDEAN, true>(d_dist, r1, nullptr, r2, 1, nullptr, 2.0, keep_at_most, max_neighbours,
                                              d_out_stats, d_out_n, d_out_idx, d_out_dist, d_out_z, as_stream(stream));
}

// ---------------------------------------------------------------------------
// host-buffer entry points
// ---------------------------------------------------------------------------
extern "C" int kpop_summarize_distances(const double *dist, uint32_t r2, uint32_t r1, uint32_t keep_at_most,
                                        uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx,
                                        double *out_dist, double *out_z) {
  KPOP_TRY(require_init());
  ArenaScope scratch;
  if (r2 == 0) return KPOP_OK;
  if (!out_stats || !out_n || (r1 && !dist)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_summarize_distances: null argument");
  hipStream_t st = nullptr;
  DevBuf dd, ds, dn, di, dv, dz;
  const uint64_t nn = (uint64_t)r2 * max_neighbours;
  KPOP_TRY(dd.alloc((uint64_t)r1 * r2 * 8));
  KPOP_TRY(ds.alloc((uint64_t)r2 * 4 * 8));
  KPOP_TRY(dn.alloc((uint64_t)r2 * 4));
  KPOP_TRY(di.alloc(nn * 4));
  KPOP_TRY(dv.alloc(nn * 8));
  KPOP_TRY(dz.alloc(nn * 8));
  if (r1) KPOP_HIP(hipMemcpyAsync(dd.p, dist, (uint64_t)r1 * r2 * 8, hipMemcpyHostToDevice, st));
  KPOP_TRY(kpop_dev_summarize_distances(dd.as<double>(), r2, r1, keep_at_most, max_neighbours, ds.as<double>(),
                                        dn.as<uint32_t>(), di.as<uint32_t>(), dv.as<double>(), dz.as<double>(), st));
  KPOP_HIP(hipMemcpyAsync(out_stats, ds.p, (uint64_t)r2 * 4 * 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipMemcpyAsync(out_n, dn.p, (uint64_t)r2 * 4, hipMemcpyDeviceToHost, st));
  if (nn) {
    KPOP_HIP(hipMemcpyAsync(out_idx, di.p, nn * 4, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipMemcpyAsync(out_dist, dv.p, nn * 8, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipMemcpyAsync(out_z, dz.p, nn * 8, hipMemcpyDeviceToHost, st));
  }
  KPOP_HIP(hipStreamSynchronize(st));
  LongListSource src;
  src.d_rows = dd.as<double>();
  return fill_long_lists(src, r1, r2, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z, st);
}

extern "C" int kpop_distance_rowwise(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                                     const double *metric, int kind, double p, int normalize, double *out) {
  KPOP_TRY(require_init());
  ArenaScope scratch;
  KPOP_TRY(check_kind(kind, p, "kpop_distance_rowwise"));
  if (r1 == 0 || r2 == 0) return KPOP_OK;
  if (!m1 || !m2 || !metric || !out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_distance_rowwise: null argument");
  if (n_dims == 0) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_distance_rowwise: n_dims must be positive");
  hipStream_t st = nullptr;
  DevBuf d1, d2, dm, dw, dout;
  KPOP_TRY(d1.alloc((uint64_t)r1 * n_dims * 8));
  KPOP_TRY(d2.alloc((uint64_t)r2 * n_dims * 8));
  KPOP_TRY(dm.alloc((uint64_t)n_dims * 8));
  KPOP_TRY(dw.alloc(kpop_dev_distance_workspace_bytes(r1, r2, n_dims)));
  KPOP_TRY(dout.alloc((uint64_t)r1 * r2 * 8));
  KPOP_HIP(hipMemcpyAsync(d1.p, m1, (uint64_t)r1 * n_dims * 8, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(d2.p, m2, (uint64_t)r2 * n_dims * 8, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(dm.p, metric, (uint64_t)n_dims * 8, hipMemcpyHostToDevice, st));
  KPOP_TRY(kpop_dev_distance_rowwise(d1.as<double>(), r1, d2.as<double>(), r2, n_dims, dm.as<double>(), kind, p,
                                     normalize, dw.p, dout.as<double>(), st));
  KPOP_HIP(hipMemcpyAsync(out, dout.p, (uint64_t)r1 * r2 * 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  return KPOP_OK;
}

extern "C" int kpop_distance_summary(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                                     const double *metric, int kind, double p, int normalize, uint32_t keep_at_most,
                                     uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx,
                                     double *out_dist, double *out_z) {
  KPOP_TRY(require_init());
  ArenaScope scratch;
  KPOP_TRY(check_kind(kind, p, "kpop_distance_summary"));
  if (r2 == 0) return KPOP_OK;
  if (!m2 || !metric || !out_stats || !out_n || (r1 && !m1))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_distance_summary: null argument");
  if (n_dims == 0) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_distance_summary: n_dims must be positive");
  hipStream_t st = nullptr;
  DevBuf d1, d2, dm, dw, ds, dn, di, dd, dz;
  const uint64_t nn = (uint64_t)r2 * max_neighbours;
  KPOP_TRY(d1.alloc((uint64_t)r1 * n_dims * 8));
  KPOP_TRY(d2.alloc((uint64_t)r2 * n_dims * 8));
  KPOP_TRY(dm.alloc((uint64_t)n_dims * 8));
  KPOP_TRY(dw.alloc(kpop_dev_distance_workspace_bytes(r1, r2, n_dims)));
  KPOP_TRY(ds.alloc((uint64_t)r2 * 4 * 8));
  KPOP_TRY(dn.alloc((uint64_t)r2 * 4));
  KPOP_TRY(di.alloc(nn * 4));
  KPOP_TRY(dd.alloc(nn * 8));
  KPOP_TRY(dz.alloc(nn * 8));
  if (r1) KPOP_HIP(hipMemcpyAsync(d1.p, m1, (uint64_t)r1 * n_dims * 8, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(d2.p, m2, (uint64_t)r2 * n_dims * 8, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(dm.p, metric, (uint64_t)n_dims * 8, hipMemcpyHostToDevice, st));
  KPOP_TRY(kpop_dev_distance_summary(d1.as<double>(), r1, d2.as<double>(), r2, n_dims, dm.as<double>(), kind, p,
                                     normalize, keep_at_most, max_neighbours, dw.p, ds.as<double>(), dn.as<uint32_t>(),
                                     di.as<uint32_t>(), dd.as<double>(), dz.as<double>(), st));
  KPOP_HIP(hipMemcpyAsync(out_stats, ds.p, (uint64_t)r2 * 4 * 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipMemcpyAsync(out_n, dn.p, (uint64_t)r2 * 4, hipMemcpyDeviceToHost, st));
  if (nn) {
    KPOP_HIP(hipMemcpyAsync(out_idx, di.p, nn * 4, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipMemcpyAsync(out_dist, dd.p, nn * 8, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipMemcpyAsync(out_z, dz.p, nn * 8, hipMemcpyDeviceToHost, st));
  }
  KPOP_HIP(hipStreamSynchronize(st));
  LongListSource src;
  src.d1 = d1.as<double>();
  src.d2 = d2.as<double>();
  src.dm = dm.as<double>();
  src.n_dims = n_dims;
  src.kind = kind;
  src.normalize = normalize;
  src.p = p;
  return fill_long_lists(src, r1, r2, max_neighbours, out_stats, out_n, out_idx, out_dist, out_z, st);
}

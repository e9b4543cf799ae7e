// splits.hip -- phylogenetic splits from embeddings by the "gaps" algorithm (KPopTwistDB -p; Matrix.get_splits with
// SplitsAlgorithm.Gaps, lib/Matrix.ml:524-600):
//   per dimension: the rows sorted by their coordinate (:566-568) and the differences between consecutive sorted
//   coordinates, the gaps (:571); all gaps ordered by decreasing size, then dimension, then position (:590-601); each of
//   the first max_splits gaps (dimension i, position j) is the split {the j + 1 rows with the smallest coordinate i}.
// On the device: one stable LSD radix sort of (coordinate key, row) pairs per dimension, then ONE stable sort of all
// n_dims x (rows - 1) gaps by descending size -- laid out dimension-major, position ascending, so stability yields the
// reference's tie order.  Rows with equal coordinates keep ascending row order (OCaml's Array.sort leaves that open).
// Integer keys (order-preserving maps of the f64 bits) and one f64 subtraction per gap: bit-exact against the oracle.
#include <algorithm>
#include <vector>

#include "radix_sort.h"

namespace kpop {

__device__ __forceinline__ uint64_t f64_sort_key(double x) {
  const uint64_t b = (uint64_t)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_from_sort_key(uint64_t k) {
  const uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

// keys[r] = key of E[r][dim], rows[r] = r
__global__ void splits_column_kernel(const double *__restrict__ E, uint32_t n, uint32_t d, uint32_t dim, uint64_t *__restrict__ keys,
                                     uint32_t *__restrict__ rows) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double x = E[(uint64_t)r * d + dim];
  if (x == 0.0) x = 0.0;  // -0. and +0. compare equal in OCaml's `compare`: one key for both
  keys[r] = f64_sort_key(x);
  rows[r] = r;
}

// gaps of one sorted dimension: key of the NEGATED order (largest gap first), value = position in the dimension-major table
__global__ void splits_gaps_kernel(const uint64_t *__restrict__ sorted_keys, uint32_t n, uint32_t dim, uint64_t *__restrict__ gap_keys,
                                   uint32_t *__restrict__ gap_pos, double *__restrict__ gaps) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j + 1 >= n) return;
  const double g = __dsub_rn(f64_from_sort_key(sorted_keys[j + 1]), f64_from_sort_key(sorted_keys[j]));
  const uint64_t at = (uint64_t)dim * (n - 1) + j;
  gaps[at] = g;
  gap_keys[at] = ~f64_sort_key(g);
  gap_pos[at] = (uint32_t)at;
}

}  // namespace kpop

using namespace kpop;

// Matrix.get_splits ... Gaps (lib/Matrix.ml:524-600).  embeddings: rows x n_dims row-major.  Outputs: for the s-th
// largest gap (s < *n_splits <= max_splits) its size, dimension and position; perm: n_dims x rows, the row order of
// every dimension -- split s is perm[out_dim[s]][0 .. out_idx[s]] (inclusive).
extern "C" int kpop_splits_gaps(const double *embeddings, uint32_t rows, uint32_t n_dims, uint32_t max_splits, uint32_t *n_splits,
                                double *out_gap, uint32_t *out_dim, uint32_t *out_idx, uint32_t *perm) {
  KPOP_TRY(require_init());
  ArenaScope scratch;
  if (!n_splits) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_splits_gaps: null n_splits");
  *n_splits = 0;
  if (rows == 0 || n_dims == 0) return KPOP_OK;
  if (!embeddings || !perm) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_splits_gaps: null argument");
  const uint64_t n_gaps = (uint64_t)n_dims * (rows - 1);
  if (n_gaps >= 0xFFFFFFFFull) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_splits_gaps: %llu gaps", (unsigned long long)n_gaps);
  hipStream_t st = nullptr;
  DevBuf dE, ka, kb, va, vb, scr, gk, gk2, gp, gp2, gv, dperm;
  KPOP_TRY(dE.alloc((uint64_t)rows * n_dims * 8));
  KPOP_TRY(ka.alloc((uint64_t)rows * 8));
  KPOP_TRY(kb.alloc((uint64_t)rows * 8));
  KPOP_TRY(va.alloc((uint64_t)rows * 4));
  KPOP_TRY(vb.alloc((uint64_t)rows * 4));
  KPOP_TRY(scr.alloc(radix_scratch_bytes(std::max<uint64_t>(rows, n_gaps))));
  KPOP_TRY(gk.alloc(n_gaps * 8));
  KPOP_TRY(gk2.alloc(n_gaps * 8));
  KPOP_TRY(gp.alloc(n_gaps * 4));
  KPOP_TRY(gp2.alloc(n_gaps * 4));
  KPOP_TRY(gv.alloc(n_gaps * 8));
  KPOP_TRY(dperm.alloc((uint64_t)rows * n_dims * 4));
  KPOP_HIP(hipMemcpyAsync(dE.p, embeddings, (uint64_t)rows * n_dims * 8, hipMemcpyHostToDevice, st));
  for (uint32_t dim = 0; dim < n_dims; ++dim) {
    splits_column_kernel<<<dim3(div_up(rows, 256)), dim3(256), 0, st>>>(dE.as<double>(), rows, n_dims, dim, ka.as<uint64_t>(), va.as<uint32_t>());
    KPOP_LAUNCH_CHECK();
    uint64_t *sk = nullptr;
    uint32_t *sv = nullptr;
    KPOP_TRY(radix_sort_pairs_u64(ka.as<uint64_t>(), kb.as<uint64_t>(), va.as<uint32_t>(), vb.as<uint32_t>(), rows, 64, scr.p, st, &sk, &sv));
    KPOP_HIP(hipMemcpyAsync(dperm.as<uint32_t>() + (uint64_t)dim * rows, sv, (uint64_t)rows * 4, hipMemcpyDeviceToDevice, st));
    if (rows > 1) {
      splits_gaps_kernel<<<dim3(div_up(rows - 1, 256)), dim3(256), 0, st>>>(sk, rows, dim, gk.as<uint64_t>(), gp.as<uint32_t>(), gv.as<double>());
      KPOP_LAUNCH_CHECK();
    }
  }
  KPOP_HIP(hipMemcpyAsync(perm, dperm.p, (uint64_t)rows * n_dims * 4, hipMemcpyDeviceToHost, st));
  const uint32_t take = (uint32_t)std::min<uint64_t>(max_splits, n_gaps);
  if (take) {
    uint64_t *sk = nullptr;
    uint32_t *sv = nullptr;
    KPOP_TRY(radix_sort_pairs_u64(gk.as<uint64_t>(), gk2.as<uint64_t>(), gp.as<uint32_t>(), gp2.as<uint32_t>(), n_gaps, 64, scr.p, st, &sk, &sv));
    std::vector<uint32_t> pos(take);
    KPOP_HIP(hipMemcpyAsync(pos.data(), sv, (uint64_t)take * 4, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipStreamSynchronize(st));
    std::vector<double> all;  // the few gap sizes wanted, fetched one by one would be `take` tiny copies: take the table
    all.resize(n_gaps);
    KPOP_HIP(hipMemcpy(all.data(), gv.p, n_gaps * 8, hipMemcpyDeviceToHost));
    for (uint32_t s = 0; s < take; ++s) {
      out_gap[s] = all[pos[s]];
      out_dim[s] = pos[s] / (rows - 1);
      out_idx[s] = pos[s] % (rows - 1);
    }
  }
  KPOP_HIP(hipStreamSynchronize(st));
  *n_splits = take;
  return KPOP_OK;
}

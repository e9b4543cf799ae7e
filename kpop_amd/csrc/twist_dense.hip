// twist_dense.hip -- the twist as a dense contraction on the f64 matrix cores (north_star's "LDS-tiled batched GEMM on
// MFMA"): twisted[B x D] = X[B x n_kmers] * T[n_kmers x D], X = the normalised spectra laid out
// densely, T = the twister's rows as they stand in HBM (k-mer-major, [n_rows][d_pad]).
//
// This does 2 * n_kmers * D flops per spectrum whatever the spectrum holds, against 2 * nnz * D for the reference's
// sparse mat-vec (lib/Twister.ml:183), so it only pays when spectra are dense AND many: it reads the twister once per
// batch tile instead of once per line.  DESIGN.md 5.9 has the measured crossover (tools/ab_dense_twist.py): genomes at
// k <= 8-9; never for 150 bp reads (139 of 8.39 M columns) and never at k = 12 (the dense image of a batch is too large).
// The order of additions differs from the reference's chain (blocked K, split-K slabs added in slab order): results agree
// to rounding (tests: 1e-12), and the CLI path keeps the chain unless kpop_tune("dense", 1|2) asks for this one.
#include <algorithm>

#include "gemm_f64.h"
#include "twister.h"

namespace kpop {

// acc[s] = sum of the values of the lines the twister knows (lib/Twister.ml:158); one wave per spectrum
__global__ __launch_bounds__(256) void dense_acc_kernel(TwisterView tv, const uint64_t *__restrict__ hash, const double *__restrict__ value,
                                                        const uint64_t *__restrict__ offsets, uint32_t s0, uint32_t n, double *__restrict__ acc) {
  const int lane = threadIdx.x & 63;
  const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n) return;
  const uint64_t lo = offsets[s0 + s], hi = offsets[s0 + s + 1];
  double part = 0.0;
  for (uint64_t i = lo + lane; i < hi; i += 64)
    if (lookup_col(tv, hash[i]) != kNoCol) part += value[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) acc[s] = part;
}

// X[s][row(hash)] += value / acc  (duplicate lines of a spectrum add up, :160-163)
__global__ __launch_bounds__(256) void dense_fill_kernel(TwisterView tv, const uint64_t *__restrict__ hash, const double *__restrict__ value,
                                                         const uint64_t *__restrict__ offsets, uint32_t s0, uint32_t n, int normalize,
                                                         const double *__restrict__ acc, double *__restrict__ X, uint64_t ldx) {
  const int lane = threadIdx.x & 63;
  const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n) return;
  const uint64_t lo = offsets[s0 + s], hi = offsets[s0 + s + 1];
  const double a = acc[s];
  const bool norm = normalize && a != 0.0;
  for (uint64_t i = lo + lane; i < hi; i += 64) {
    const uint32_t col = lookup_col(tv, hash[i]);
    if (col != kNoCol) atomicAdd(&X[(uint64_t)s * ldx + col], norm ? value[i] / a : value[i]);
  }
}

__global__ void dense_copy_out_kernel(const double *__restrict__ C, uint32_t n, uint32_t n_dims, uint32_t ldc, double *__restrict__ out) {
  const uint64_t total = (uint64_t)n * n_dims, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) out[e] = C[(e / n_dims) * ldc + e % n_dims];
}

constexpr uint64_t kDenseTileBytes = 2ull << 30;  // dense image of one batch tile

static uint32_t dense_tile_rows(const kpop_twister *tw, uint32_t n_spectra) {
  const uint64_t per = std::max<uint64_t>(1, tw->n_rows) * 8;
  uint64_t b = std::max<uint64_t>(128, kDenseTileBytes / per / 128 * 128);
  return (uint32_t)std::min<uint64_t>(b, std::max<uint32_t>(n_spectra, 1));
}
static uint32_t dense_splits(uint32_t rows, const kpop_twister *tw) {
  const uint32_t tiles = div_up(rows, kGT) * div_up(tw->n_dims, kGT);
  const uint32_t want = std::max(1u, (uint32_t)ctx().n_cus * 2 / std::max(1u, tiles));
  const uint64_t max_by_k = std::max<uint64_t>(1, tw->n_rows / 512);
  return (uint32_t)std::min<uint64_t>(std::min<uint64_t>(want, max_by_k), 64);
}

}  // namespace kpop

using namespace kpop;

extern "C" uint64_t kpop_dev_twist_dense_workspace_bytes(const kpop_twister *tw, uint32_t n_spectra) {
  if (!tw) return 0;
  const uint32_t rows = dense_tile_rows(tw, n_spectra);
  const uint32_t splits = dense_splits(rows, tw);
  return (uint64_t)rows * std::max<uint64_t>(1, tw->n_rows) * 8 + (uint64_t)rows * 8 + (uint64_t)(splits + 1) * rows * tw->n_dims * 8 + 4096;
}

extern "C" int kpop_dev_twist_dense(const kpop_twister *tw, const uint64_t *d_hash, const double *d_value, const uint64_t *d_offsets,
                                    uint32_t n_spectra, int normalize, void *d_work, double *d_out, void *stream) {
  KPOP_TRY(require_init());
  if (!tw || !d_offsets || !d_out || !d_work) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_twist_dense: null argument");
  if (n_spectra == 0) return KPOP_OK;
  hipStream_t st = as_stream(stream);
  const TwisterView tv = view_of(tw);
  const uint64_t K = tw->n_rows;
  const uint32_t D = tw->n_dims, tile = dense_tile_rows(tw, n_spectra), splits = dense_splits(tile, tw);
  char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(d_work) + 255) & ~(uintptr_t)255);
  double *X = reinterpret_cast<double *>(w);
  double *acc = X + (uint64_t)tile * std::max<uint64_t>(1, K);
  double *C = acc + tile;                       // [tile][D]
  double *slabs = C + (uint64_t)tile * D;       // [splits][tile][D]
  for (uint32_t s0 = 0; s0 < n_spectra; s0 += tile) {
    const uint32_t nb = std::min(tile, n_spectra - s0);
    KPOP_HIP(hipMemsetAsync(X, 0, (uint64_t)nb * std::max<uint64_t>(1, K) * 8, st));
    dense_acc_kernel<<<dim3(div_up(nb, 4)), dim3(256), 0, st>>>(tv, d_hash, d_value, d_offsets, s0, nb, acc);
    KPOP_LAUNCH_CHECK();
    dense_fill_kernel<<<dim3(div_up(nb, 4)), dim3(256), 0, st>>>(tv, d_hash, d_value, d_offsets, s0, nb, normalize, acc, X, K);
    KPOP_LAUNCH_CHECK();
    if (K == 0) {
      KPOP_HIP(hipMemsetAsync(d_out + (uint64_t)s0 * D, 0, (uint64_t)nb * D * 8, st));
      continue;
    }
    // C[nb x D] = X[nb x K] * rows[K x d_pad]
    KPOP_TRY(gemm_f64<false>(X, K, tw->d_rows, tw->d_pad, d_out + (uint64_t)s0 * D, nb, D, K, splits, slabs, 0, st));
  }
  return KPOP_OK;
}

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

// pass A (only for relative thresholds): plain sum of v^power per column slab
__global__ __launch_bounds__(kStatBlock) void col_plain_sum_kernel(const int32_t *__restrict__ storage, uint64_t ld,
                                                                  uint64_t n_rows, double power, int power_one,
                                                                  uint32_t n_slabs, double *__restrict__ partial) {
  const uint32_t col = blockIdx.y, slab = blockIdx.x;
  const int32_t *v = storage + (uint64_t)col * ld;
  const uint64_t lo = (uint64_t)slab * kRowsPerStatBlock, hi = min(n_rows, lo + kRowsPerStatBlock);
  double s = 0.;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += kStatBlock) s += pow_count((double)v[i], power, power_one);
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

// pass B: thresholded statistics per column slab (lib/KMerDB.ml:196-215)
__global__ __launch_bounds__(kStatBlock) void col_stats_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint64_t n_rows,
                                                              double power, int power_one, const double *__restrict__ thr,
                                                              uint32_t n_slabs, ColPartial *__restrict__ partial) {
  const uint32_t col = blockIdx.y, slab = blockIdx.x;
  const int32_t *v = storage + (uint64_t)col * ld;
  const uint64_t lo = (uint64_t)slab * kRowsPerStatBlock, hi = min(n_rows, lo + kRowsPerStatBlock);
  const double threshold = thr[col];
  double nz = 0., mx = 0., s = 0., sl = 0.;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += kStatBlock) {
    const double f = (double)v[i];
    if (f >= threshold) {
      nz += 1.;
      mx = fmax(mx, f);
      s += pow_count(f, power, power_one);
      sl += log(f) * power;
    }
  }
  __shared__ ColPartial sh[kStatBlock / 64];
  nz = wave_sum(nz);
  mx = wave_max(mx);
  s = wave_sum(s);
  sl = wave_sum(sl);
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

__global__ void col_stats_final_kernel(const ColPartial *__restrict__ partial, uint32_t n_slabs, uint32_t n_cols,
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
  col_stats[4 * (uint64_t)c + 3] = t.sum_log;
}

// one thread per k-mer, spectra visited in order: the reference's own summation order (lib/KMerDB.ml:182-215, Row)
__global__ __launch_bounds__(256) void row_stats_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint32_t n_cols,
                                                        uint64_t n_rows, double threshold0, double power, int power_one,
                                                        double *__restrict__ row_stats) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
    double threshold = threshold0;
    if (threshold0 < 1.) {
      double s = 0.;
      for (uint32_t c = 0; c < n_cols; ++c) s += pow_count((double)storage[(uint64_t)c * ld + r], power, power_one);
      threshold = threshold0 * s;
    }
    double nz = 0., mx = 0., s = 0., sl = 0.;
    for (uint32_t c = 0; c < n_cols; ++c) {
      const double f = (double)storage[(uint64_t)c * ld + r];
      if (f >= threshold) {
        nz += 1.;
        mx = fmax(mx, f);
        s += pow_count(f, power, power_one);
        sl += log(f) * power;
      }
    }
    double *o = row_stats + 4 * r;
    o[0] = nz;
    o[1] = mx;
    o[2] = s;
    o[3] = sl;
  }
}

// Int32.of_float on x86-64: truncate to the native int, keep the low 32 bits (lib/KMerDB.ml:716)
__device__ __forceinline__ int32_t int32_of_float(double x) {
  if (!(x > -9.2e18 && x < 9.2e18)) return 0;
  return (int32_t)(uint32_t)(uint64_t)(int64_t)x;
}

// RescaledMean: sum over the selected spectra, in the order given, of count * max_norm / norm
// (lib/KMerDB.ml:687-704).  sel/norm list only the spectra whose norm is positive (:693).
__global__ __launch_bounds__(256) void combine_mean_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint64_t n_rows,
                                                           const uint32_t *__restrict__ sel, const double *__restrict__ norm,
                                                           uint32_t m, double max_norm, int32_t *__restrict__ out,
                                                           double *__restrict__ norm_partial) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  double acc_norm = 0.;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
    double s = 0.;
    for (uint32_t j = 0; j < m; ++j) {
      const double c = (double)storage[(uint64_t)sel[j] * ld + r];
      s = __dadd_rn(s, __ddiv_rn(__dmul_rn(c, max_norm), norm[j]));
    }
    acc_norm += s;
    out[r] = int32_of_float(s);
  }
  __shared__ double sh[4];
  acc_norm = wave_sum(acc_norm);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc_norm;
  __syncthreads();
  if (threadIdx.x == 0) norm_partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// RescaledMedian (lib/KMerDB.ml:705-706): a tile of R k-mers x m spectra is staged in LDS (lanes along k-mers, so
// the loads are full lines), every k-mer's row is sorted by a bitonic network run by the whole block, and the upper
// median sorted[m/2] is multiplied by the number of selected spectra.  Row stride P+1 keeps the staging writes off
// one bank.
__global__ __launch_bounds__(256) void combine_median_kernel(const int32_t *__restrict__ storage, uint64_t ld, uint64_t n_rows,
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

__global__ void sum_partials_kernel(const double *__restrict__ partial, uint32_t n, double *__restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.;
    for (uint32_t i = 0; i < n; ++i) s += partial[i];
    *out = s;
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
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride)
    out[(uint64_t)c * n_rows + r] = transform_one(which, threshold, power, col_stats + 4 * (uint64_t)c, storage[(uint64_t)c * ld + r]);
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
      if (row0 + r < n_rows && col0 + c < n_cols) out[(row0 + r) * n_cols + col0 + c] = tile[c][r];
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
    if (n_cols > 65535) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_dev_counter_stats: more than 65535 spectra in one call");
    if (threshold < 1.) {
      col_plain_sum_kernel<<<dim3(slabs, n_cols), dim3(kStatBlock), 0, st>>>(d_storage, ld, n_rows, power, p1, slabs, plain);
      KPOP_LAUNCH_CHECK();
    }
    col_threshold_kernel<<<dim3(div_up(n_cols, 256)), dim3(256), 0, st>>>(plain, slabs, n_cols, threshold, thr);
    KPOP_LAUNCH_CHECK();
    col_stats_kernel<<<dim3(slabs, n_cols), dim3(kStatBlock), 0, st>>>(d_storage, ld, n_rows, power, p1, thr, slabs, partial);
    KPOP_LAUNCH_CHECK();
    col_stats_final_kernel<<<dim3(div_up(n_cols, 256)), dim3(256), 0, st>>>(partial, slabs, n_cols, d_col_stats);
    KPOP_LAUNCH_CHECK();
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
  hipStream_t st = as_stream(stream);
  double *partial = reinterpret_cast<double *>(d_workspace);
  uint32_t grid;
  if (criterion == KPOP_COMBINE_MEAN) {
    grid = std::min<uint32_t>(div_up(n_rows, 256), 1u << 16);
    combine_mean_kernel<<<dim3(grid), dim3(256), 0, st>>>(d_storage, ld, n_rows, d_sel, d_norm, n_valid, max_norm, d_out, partial);
  } else {
    uint32_t P = 2;
    while (P < n_valid) P <<= 1;
    const uint32_t budget = 8192 - 8;  // doubles of LDS (64 KB less the reduction scratch)
    if (P + 1 > budget) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_dev_counter_combine: median of more than 4096 spectra");
    const uint32_t R = std::max<uint32_t>(1, std::min<uint32_t>(64, budget / (P + 1)));
    grid = (uint32_t)std::min<uint64_t>((n_rows + R - 1) / R, 1u << 16);
    combine_median_kernel<<<dim3(grid), dim3(256), (size_t)R * (P + 1) * 8, st>>>(d_storage, ld, n_rows, d_sel, d_norm, n_valid, n_sel,
                                                                                   max_norm, P, R, d_out, partial);
  }
  KPOP_LAUNCH_CHECK();
  if (d_out_norm) {
    sum_partials_kernel<<<dim3(1), dim3(64), 0, st>>>(partial, grid, d_out_norm);
    KPOP_LAUNCH_CHECK();
  }
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
    if (n_cols > 65535) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_dev_counter_transform: more than 65535 spectra in one call");
    transform_kernel<<<dim3((uint32_t)std::min<uint64_t>(div_up(n_rows, 256), 1u << 16), n_cols), dim3(256), 0, st>>>(
        d_storage, ld, n_cols, n_rows, which, threshold, power, d_col_stats, d_out);
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
  KPOP_TRY(dw.alloc(kpop_dev_counter_workspace_bytes(1, n_rows)));
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

// twister.hip -- building the device-resident twister (see twister.h).
#include <algorithm>
#include <vector>

#include "kmer.h"
#include "scan.h"
#include "twister.h"

namespace kpop {

// tmp is [n_dims][chunk] (the reference's dims-major order, one chunk of
// columns); rows is the k-mer-major destination.  32x32 LDS tile transpose.
__global__ __launch_bounds__(256) void transpose_chunk_kernel(const double *__restrict__ tmp, uint64_t chunk,
                                                              uint32_t n_dims, uint32_t d_pad,
                                                              double *__restrict__ rows, uint64_t col0) {
  __shared__ double tile[32][33];
  const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const uint64_t c_base = (uint64_t)blockIdx.x * 32;
  const uint32_t d_base = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t d = d_base + ty + 8 * i;
    uint64_t c = c_base + tx;
    tile[ty + 8 * i][tx] = (d < n_dims && c < chunk) ? tmp[(uint64_t)d * chunk + c] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint64_t c = c_base + ty + 8 * i;
    uint32_t d = d_base + tx;
    if (c < chunk && d < d_pad) rows[(col0 + c) * d_pad + d] = tile[tx][ty + 8 * i];
  }
}

// Hashtbl.add shadows: the LAST column with a given name wins (lib/Twister.ml:73-76)
__global__ void lut_fill_kernel(const uint64_t *__restrict__ col_hash, uint64_t n_cols, uint32_t *lut,
                                uint64_t lut_size, int *bad) {
  uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  uint64_t h = col_hash[c];
  if (h >= lut_size) {
    *bad = 1;
    return;
  }
  // lut starts at 0 ("none"); store col+1 with atomicMax, fixed up afterwards
  atomicMax(&lut[h], (uint32_t)(c + 1));
}

__global__ void lut_fixup_kernel(uint32_t *lut, uint64_t lut_size) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < lut_size; i += stride)
    lut[i] = lut[i] - 1u;  // 0 -> 0xFFFFFFFF (kNoCol), c+1 -> c
}

// one thread per 64-hash word: presence bits from the LUT, prefix = column of the
// first present k-mer; *ok is cleared if columns do not ascend with the hash
__global__ void rank_index_kernel(const uint32_t *__restrict__ lut, uint64_t n_words, uint64_t lut_size,
                                  RankWord *__restrict__ rsel, int *ok) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  uint64_t bits = 0;
  uint32_t prefix = 0, seen = 0;
  bool good = true;
  for (uint32_t b = 0; b < 64; ++b) {
    const uint64_t h = w * 64 + b;
    const uint32_t c = (h < lut_size) ? lut[h] : kNoCol;
    if (c == kNoCol) continue;
    if (seen == 0) prefix = c;
    else if (c != prefix + seen) good = false;
    bits |= 1ull << b;
    ++seen;
  }
  rsel[w].bits = bits;
  rsel[w].prefix = prefix;
  rsel[w].pad = seen;
  if (!good) *ok = 0;
}

// consecutive non-empty words must continue each other's numbering
__global__ void rank_index_check_kernel(const RankWord *__restrict__ rsel, uint64_t n_words, int *ok) {
  // single thread walk would be slow; each thread checks its word against the previous non-empty one
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words || rsel[w].pad == 0) return;
  uint64_t p = w;
  while (p > 0) {
    --p;
    if (rsel[p].pad) {
      if (rsel[p].prefix + rsel[p].pad != rsel[w].prefix) *ok = 0;
      return;
    }
  }
  if (rsel[w].prefix != 0) *ok = 0;
}

static int build_rank_index(kpop_twister *tw, hipStream_t st) {
  if (!tw->d_lut) return 0;
  const uint64_t lut_size = 1ull << (2 * tw->k);
  const uint64_t n_words = (lut_size + 63) / 64;
  RankWord *rsel = nullptr;
  KPOP_HIP(hipMalloc((void **)&rsel, n_words * sizeof(RankWord)));
  DevBuf ok;
  if (ok.alloc(4) != 0) {
    (void)hipFree(rsel);
    return KPOP_ERR_HIP;
  }
  int one = 1;
  hipError_t e = hipMemcpyAsync(ok.p, &one, 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    rank_index_kernel<<<dim3(div_up(n_words, 256)), dim3(256), 0, st>>>(tw->d_lut, n_words, lut_size, rsel, ok.as<int>());
    rank_index_check_kernel<<<dim3(div_up(n_words, 256)), dim3(256), 0, st>>>(rsel, n_words, ok.as<int>());
    e = hipMemcpy(&one, ok.p, 4, hipMemcpyDeviceToHost);
  }
  if (e != hipSuccess) {
    (void)hipFree(rsel);
    KPOP_FAIL(KPOP_ERR_HIP, "build_rank_index: %s", hipGetErrorString(e));
  }
  if (one) {
    tw->d_rsel = rsel;
    tw->device_bytes += n_words * sizeof(RankWord);
  } else {
    (void)hipFree(rsel);  // columns not in hash order: the LUT stays the index
  }
  return 0;
}

struct CanonFlag {
  int k;
  int content;
  __device__ uint32_t operator()(uint64_t h) const {
    return (content == KPOP_DNA_SS || h <= revcomp(h, k)) ? 1u : 0u;
  }
};

struct LutOut {
  uint32_t *lut;
  __device__ void operator()(uint64_t h, uint64_t prefix, uint32_t flag) const {
    lut[h] = flag ? (uint32_t)prefix : kNoCol;
  }
};

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// coefficient(d, h) of the synthetic twister, same function as
// the oracle's kpo_synth_twister_coeff
__device__ __forceinline__ double synth_coeff(uint64_t seed, uint32_t d, uint64_t h) {
  uint64_t z = mix64(seed ^ ((uint64_t)d << 40) ^ h);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

__global__ __launch_bounds__(256) void synth_rows_kernel(uint64_t seed, const uint32_t *__restrict__ lut,
                                                         uint64_t lut_size, uint32_t n_dims, uint32_t d_pad,
                                                         double *__restrict__ rows) {
  // one wave per hash value; lanes sweep the dims so stores are coalesced.  Grid-stride: HIP caps
  // gridDim.x * blockDim.x at 2^32 and 4^15 hashes need more waves than that.
  const int lane = threadIdx.x & 63;
  for (uint64_t h = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); h < lut_size; h += (uint64_t)gridDim.x * 4) {
    const uint32_t col = lut[h];
    if (col == kNoCol) continue;
    for (uint32_t d = lane; d < d_pad; d += 64)
      rows[(uint64_t)col * d_pad + d] = (d < n_dims) ? synth_coeff(seed, d, h) : 0.0;
  }
}

static int alloc_common(kpop_twister *tw) {
  tw->d_pad = (tw->n_dims + 15u) & ~15u;
  uint64_t row_bytes = tw->n_cols * (uint64_t)tw->d_pad * sizeof(double);
  KPOP_HIP(hipMalloc((void **)&tw->d_rows, row_bytes ? row_bytes : 8));
  tw->device_bytes += row_bytes;
  return 0;
}

}  // namespace kpop

using namespace kpop;

extern "C" int kpop_twister_free(kpop_twister *tw) {
  if (!tw) return KPOP_OK;
  if (tw->d_rows) (void)hipFree(tw->d_rows);
  if (tw->d_lut) (void)hipFree(tw->d_lut);
  if (tw->d_rsel) (void)hipFree(tw->d_rsel);
  if (tw->d_sorted_hash) (void)hipFree(tw->d_sorted_hash);
  if (tw->d_sorted_col) (void)hipFree(tw->d_sorted_col);
  delete tw;
  return KPOP_OK;
}

extern "C" int kpop_twister_info(const kpop_twister *tw, uint64_t *n_cols, uint32_t *n_dims, int *k,
                                 uint64_t *device_bytes) {
  if (!tw) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_info: null twister");
  if (n_cols) *n_cols = tw->n_cols;
  if (n_dims) *n_dims = tw->n_dims;
  if (k) *k = tw->k;
  if (device_bytes) *device_bytes = tw->device_bytes;
  return KPOP_OK;
}

namespace {
struct TwGuard {  // frees a half-built twister on an error path
  kpop_twister *tw;
  ~TwGuard() {
    if (tw) kpop_twister_free(tw);
  }
};
}  // namespace

extern "C" int kpop_twister_load(const double *T_dims_major, uint64_t n_cols, uint32_t n_dims,
                                 const uint64_t *col_hash, int k, kpop_twister **out) {
  KPOP_TRY(require_init());
  if (!out || (!T_dims_major && n_cols && n_dims) || (!col_hash && n_cols))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_load: null argument");
  if (k < 1 || k > kMaxK) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_load: k=%d out of range 1..%d", k, kMaxK);
  if (n_dims == 0) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_load: n_dims must be positive");
  if (n_cols >= 0xFFFFFFFFull) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_twister_load: more than 2^32-2 columns");
  kpop_twister *tw = new kpop_twister();
  TwGuard guard{tw};
  tw->k = k;
  tw->n_dims = n_dims;
  tw->n_cols = n_cols;
  KPOP_TRY(alloc_common(tw));
  hipStream_t st = nullptr;

  // --- rows: chunked H2D of the dims-major slabs + device transpose
  if (n_cols) {
    const uint64_t budget = 256ull << 20;  // staging bytes
    uint64_t chunk = std::max<uint64_t>(32, budget / (8ull * n_dims));
    chunk = std::min<uint64_t>(chunk, n_cols);
    DevBuf tmp;
    KPOP_TRY(tmp.alloc(chunk * n_dims * 8));
    for (uint64_t c0 = 0; c0 < n_cols; c0 += chunk) {
      uint64_t cn = std::min<uint64_t>(chunk, n_cols - c0);
      for (uint32_t d = 0; d < n_dims; ++d)
        KPOP_HIP(hipMemcpyAsync(tmp.as<double>() + (uint64_t)d * cn, T_dims_major + (uint64_t)d * n_cols + c0,
                                cn * 8, hipMemcpyHostToDevice, st));
      dim3 grid(div_up(cn, 32), div_up(tw->d_pad, 32));
      transpose_chunk_kernel<<<grid, dim3(256), 0, st>>>(tmp.as<double>(), cn, n_dims, tw->d_pad, tw->d_rows, c0);
      KPOP_LAUNCH_CHECK();
      KPOP_HIP(hipStreamSynchronize(st));
    }
  }

  // --- name -> column
  if (k <= kLutMaxK) {
    const uint64_t lut_size = 1ull << (2 * k);
    KPOP_HIP(hipMalloc((void **)&tw->d_lut, lut_size * 4));
    tw->device_bytes += lut_size * 4;
    KPOP_HIP(hipMemsetAsync(tw->d_lut, 0, lut_size * 4, st));
    DevBuf dh, bad;
    KPOP_TRY(dh.alloc(n_cols * 8));
    KPOP_TRY(bad.alloc(4));
    KPOP_HIP(hipMemsetAsync(bad.p, 0, 4, st));
    if (n_cols) {
      KPOP_HIP(hipMemcpyAsync(dh.p, col_hash, n_cols * 8, hipMemcpyHostToDevice, st));
      lut_fill_kernel<<<dim3(div_up(n_cols, 256)), dim3(256), 0, st>>>(dh.as<uint64_t>(), n_cols, tw->d_lut,
                                                                        lut_size, bad.as<int>());
      KPOP_LAUNCH_CHECK();
    }
    lut_fixup_kernel<<<dim3(capped_grid(div_up(lut_size, 256))), dim3(256), 0, st>>>(tw->d_lut, lut_size);
    KPOP_LAUNCH_CHECK();
    int h_bad = 0;
    KPOP_HIP(hipMemcpy(&h_bad, bad.p, 4, hipMemcpyDeviceToHost));
    if (h_bad) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_load: a column hash does not fit k=%d", k);
    KPOP_TRY(build_rank_index(tw, st));
  } else {
    std::vector<std::pair<uint64_t, uint32_t>> hc(n_cols);
    const uint64_t lim = kmer_mask(k);
    for (uint64_t c = 0; c < n_cols; ++c) {
      if (col_hash[c] > lim) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_load: a column hash does not fit k=%d", k);
      hc[c] = {col_hash[c], (uint32_t)c};
    }
    std::sort(hc.begin(), hc.end());
    std::vector<uint64_t> sh;
    std::vector<uint32_t> sc;
    for (uint64_t i = 0; i < n_cols; ++i) {
      if (i + 1 < n_cols && hc[i + 1].first == hc[i].first) continue;  // keep the last column of a name
      sh.push_back(hc[i].first);
      sc.push_back(hc[i].second);
    }
    uint64_t m = sh.size();
    KPOP_HIP(hipMalloc((void **)&tw->d_sorted_hash, m ? m * 8 : 8));
    KPOP_HIP(hipMalloc((void **)&tw->d_sorted_col, m ? m * 4 : 4));
    tw->device_bytes += m * 12;
    if (m) {
      KPOP_HIP(hipMemcpy(tw->d_sorted_hash, sh.data(), m * 8, hipMemcpyHostToDevice));
      KPOP_HIP(hipMemcpy(tw->d_sorted_col, sc.data(), m * 4, hipMemcpyHostToDevice));
    }
    tw->n_sorted = m;  // bisection runs over the de-duplicated table; rows keep their column ids
  }
  KPOP_HIP(hipStreamSynchronize(st));
  guard.tw = nullptr;
  *out = tw;
  return KPOP_OK;
}

extern "C" int kpop_twister_synth(uint64_t seed, int k, int content, uint32_t n_dims, kpop_twister **out) {
  KPOP_TRY(require_init());
  if (!out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_synth: null out");
  if (k < 1 || k > kLutMaxK)
    KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_twister_synth: k=%d (dense synthetic twister needs k <= %d)", k, kLutMaxK);
  if (content != KPOP_DNA_DS && content != KPOP_DNA_SS) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_synth: content");
  if (n_dims == 0) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_synth: n_dims must be positive");
  const uint64_t lut_size = 1ull << (2 * k);
  uint64_t n_cols = lut_size;
  if (content == KPOP_DNA_DS) n_cols = (k % 2 == 0) ? (lut_size + (1ull << k)) / 2 : lut_size / 2;
  if (n_cols >= 0xFFFFFFFFull) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_twister_synth: too many columns");
  kpop_twister *tw = new kpop_twister();
  TwGuard guard{tw};
  tw->k = k;
  tw->n_dims = n_dims;
  tw->n_cols = n_cols;
  KPOP_TRY(alloc_common(tw));
  KPOP_HIP(hipMalloc((void **)&tw->d_lut, lut_size * 4));
  tw->device_bytes += lut_size * 4;
  hipStream_t st = nullptr;
  DevBuf sums;
  KPOP_TRY(sums.alloc((scan_blocks(lut_size) + 1) * 8));
  KPOP_TRY(exclusive_scan(CanonFlag{k, content}, LutOut{tw->d_lut}, lut_size, sums.as<uint64_t>(), st));
  uint64_t total = 0;
  KPOP_HIP(hipMemcpy(&total, sums.as<uint64_t>() + scan_blocks(lut_size), 8, hipMemcpyDeviceToHost));
  if (total != n_cols)
    KPOP_FAIL(KPOP_ERR_HIP, "kpop_twister_synth: enumerated %llu k-mers, expected %llu", (unsigned long long)total,
              (unsigned long long)n_cols);
  synth_rows_kernel<<<dim3(capped_grid(div_up(lut_size, 4))), dim3(256), 0, st>>>(seed, tw->d_lut, lut_size, n_dims, tw->d_pad,
                                                                      tw->d_rows);
  KPOP_LAUNCH_CHECK();
  KPOP_TRY(build_rank_index(tw, st));
  KPOP_HIP(hipStreamSynchronize(st));
  guard.tw = nullptr;
  *out = tw;
  return KPOP_OK;
}

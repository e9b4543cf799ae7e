// twister.hip -- building the device-resident twister (see twister.h).
#include <algorithm>
#include <vector>

#include "kmer.h"
#include "scan.h"
#include "twister.h"

namespace kpop {

// tmp is [n_dims][chunk] (the reference's dims-major order, one chunk of columns); column col0+c of the file
// goes to device row dst_row[col0+c] (kNoCol: a shadowed duplicate, dropped).  32x32 LDS tile transpose.
__global__ __launch_bounds__(256) void transpose_chunk_kernel(const double *__restrict__ tmp, uint64_t chunk,
                                                              uint32_t n_dims, uint32_t d_pad,
                                                              const uint32_t *__restrict__ dst_row,
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
    if (c < chunk && d < d_pad) {
      const uint32_t r = dst_row[col0 + c];
      if (r != kNoCol) rows[(uint64_t)r * d_pad + d] = tile[tx][ty + 8 * i];
    }
  }
}

// presence bits of the (distinct) hashes
__global__ void rank_bits_kernel(const uint64_t *__restrict__ hashes, uint64_t n, RankWord *rsel) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t h = hashes[i];
    atomicOr(reinterpret_cast<unsigned long long *>(&rsel[h >> 6].bits), 1ull << (h & 63));
  }
}

// presence bits of every canonical (DNA-ds) / every (DNA-ss) k-mer with hash in [hash_lo, hash_hi): the synthetic
// twister, or one rank's slice of it
__global__ void rank_bits_canonical_kernel(int k, int content, uint64_t n_words, uint64_t hash_lo, uint64_t hash_hi, RankWord *rsel) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
    uint64_t bits = 0;
    for (uint32_t b = 0; b < 64; ++b) {
      const uint64_t h = w * 64 + b;
      if (h >= hash_lo && h < hash_hi && (content == KPOP_DNA_SS || h <= revcomp(h, k))) bits |= 1ull << b;
    }
    rsel[w].bits = bits;
  }
}

struct WordPopcount {
  const RankWord *rsel;
  __device__ uint32_t operator()(uint64_t w) const { return (uint32_t)__popcll(rsel[w].bits); }
};
struct WordPrefixOut {
  RankWord *rsel;
  __device__ void operator()(uint64_t w, uint64_t prefix, uint32_t count) const {
    rsel[w].prefix = (uint32_t)prefix;
    rsel[w].count = count;
  }
};

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// coefficient(d, h) of the synthetic twister, same function as the oracle's kpo_synth_twister_coeff
__device__ __forceinline__ double synth_coeff(uint64_t seed, uint32_t d, uint64_t h) {
  uint64_t z = mix64(seed ^ ((uint64_t)d << 40) ^ h);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

// n_synth dimensions come from synth_coeff; dimensions n_synth .. n_dims-1 (the optional accumulator dimension of a
// k-mer-row shard) are 1 so that the twist of a spectrum carries sum(v_h over the shard's k-mers) along
__global__ __launch_bounds__(256) void synth_rows_kernel(uint64_t seed, const RankWord *__restrict__ rsel,
                                                         uint64_t n_hashes, uint32_t n_synth, uint32_t n_dims, uint32_t d_pad,
                                                         double *__restrict__ rows) {
  // one wave per hash value; lanes sweep the dims so stores are coalesced.  Grid-stride: HIP caps
  // gridDim.x * blockDim.x at 2^32 and 4^15 hashes need more waves than that.
  const int lane = threadIdx.x & 63;
  for (uint64_t h = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); h < n_hashes; h += (uint64_t)gridDim.x * 4) {
    const RankWord w = rsel[h >> 6];
    const uint32_t b = (uint32_t)h & 63u;
    if (!((w.bits >> b) & 1ull)) continue;
    const uint32_t row = w.prefix + (uint32_t)__popcll(w.bits & ((1ull << b) - 1ull));
    for (uint32_t d = lane; d < d_pad; d += 64)
      rows[(uint64_t)row * d_pad + d] = (d < n_synth) ? synth_coeff(seed, d, h) : (d < n_dims ? 1.0 : 0.0);
  }
}

static int alloc_rows(kpop_twister *tw) {
  tw->d_pad = (tw->n_dims + 15u) & ~15u;
  const uint64_t row_bytes = tw->n_rows * (uint64_t)tw->d_pad * sizeof(double);
  KPOP_HIP(hipMalloc((void **)&tw->d_rows, row_bytes ? row_bytes : 8));
  tw->device_bytes += row_bytes;
  return 0;
}

// rank-select words: bits must already be set; fills prefix/count; returns the total number of set bits
static int finish_rank_select(kpop_twister *tw, uint64_t n_words, uint64_t *total, hipStream_t st) {
  DevBuf sums;
  KPOP_TRY(sums.alloc((scan_blocks(n_words) + 1) * 8));
  RankWord *rsel = reinterpret_cast<RankWord *>(tw->d_rsel);
  KPOP_TRY(exclusive_scan(WordPopcount{rsel}, WordPrefixOut{rsel}, n_words, sums.as<uint64_t>(), st));
  KPOP_HIP(hipMemcpyAsync(total, sums.as<uint64_t>() + scan_blocks(n_words), 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  return 0;
}

// rank words -> 64-byte blocks (k >= kRankBlockMinK): block b holds the presence bits of hashes [480 b, 480 b + 480) and the rank
// of the first of them; a thread a block (it reads nine words)
__global__ __launch_bounds__(256) void rank_blocks_kernel(const RankWord *__restrict__ rsel, uint64_t n_words, uint64_t n_blocks, uint4 *__restrict__ rblk) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= n_blocks) return;
  const uint64_t h0 = b * kRankBlockBits;
  const uint64_t w0 = h0 >> 6;
  const uint32_t sh = (uint32_t)(h0 & 63);
  auto bits_of = [&](uint64_t w) -> uint64_t { return w < n_words ? rsel[w].bits : 0ull; };
  uint32_t d[16];
#pragma unroll
  for (uint32_t i = 0; i < 8; ++i) {  // eight 64-bit pieces from h0 on (the last one half used)
    const uint64_t lo = bits_of(w0 + i), hi = bits_of(w0 + i + 1);
    const uint64_t x = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    d[2 * i] = (uint32_t)x;
    d[2 * i + 1] = (uint32_t)(x >> 32);
  }
  const RankWord first = rsel[w0 < n_words ? w0 : n_words - 1];
  d[15] = first.prefix + (uint32_t)__popcll(first.bits & ((1ull << sh) - 1ull));  // (dword 15 of the pieces is not the block's: bits 480..511)
  rblk[b * 4 + 0] = make_uint4(d[0], d[1], d[2], d[3]);
  rblk[b * 4 + 1] = make_uint4(d[4], d[5], d[6], d[7]);
  rblk[b * 4 + 2] = make_uint4(d[8], d[9], d[10], d[11]);
  rblk[b * 4 + 3] = make_uint4(d[12], d[13], d[14], d[15]);
}

// from kRankBlockMinK on: the index as blocks, the words freed (after everything that reads them: the rows' placement)
static int compact_rank_index(kpop_twister *tw, hipStream_t st) {
  if (tw->k < kRankBlockMinK || !tw->d_rsel) return 0;
  const uint64_t n_words = ((1ull << (2 * tw->k)) + 63) / 64, n_blocks = rank_blocks(tw->k);
  KPOP_HIP(hipMalloc(&tw->d_rblk, n_blocks * 64));
  rank_blocks_kernel<<<dim3(div_up(n_blocks, 256)), dim3(256), 0, st>>>(reinterpret_cast<const RankWord *>(tw->d_rsel), n_words, n_blocks,
                                                                       reinterpret_cast<uint4 *>(tw->d_rblk));
  KPOP_LAUNCH_CHECK();
  KPOP_HIP(hipStreamSynchronize(st));
  KPOP_HIP(hipFree(tw->d_rsel));
  tw->d_rsel = nullptr;
  tw->device_bytes += n_blocks * 64;
  tw->device_bytes -= n_words * sizeof(RankWord);
  return 0;
}

// direct[h - lo] = the row of hash h, or the mark of a row that does not exist: a thread an element
__global__ __launch_bounds__(256) void direct_rows_kernel(TwisterView tv, uint64_t lo, uint64_t n_hashes, double *__restrict__ direct) {
  for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < n_hashes * tv.d_pad; idx += (uint64_t)gridDim.x * 256) {
    const uint64_t h = lo + idx / tv.d_pad;
    const uint32_t e = (uint32_t)(idx % tv.d_pad);
    const uint32_t col = lookup_col(tv, h);
    if (col != kNoCol) direct[idx] = tv.rows[(uint64_t)col * tv.d_pad + e];
    else if (e == 0) direct[idx] = __longlong_as_double((long long)kDirectAbsent);
  }
}

// the rows once more, at their hashes (twister.h): where that pays and fits.  [lo, hi): the hashes the twister keeps rows of -- all of
// them, or the slice of a k-mer-row-sharded twister (a rank of BASELINE config 5's multi-GPU form: a window whose k-mer is another
// rank's then costs this rank NOTHING, one of its own ONE miss; through the index either cost a line of index first)
static int build_direct_rows(kpop_twister *tw, hipStream_t st, uint64_t lo = 0, uint64_t hi = ~0ull) {
  const int mode = ctx().tune_direct;  // 2: by the rule below; 1: whenever it fits; 0: never
  if (!mode || tw->k < (mode == 1 ? 1 : kDirectMinK) || tw->k > 15 || tw->d_pad > 32 || !tw->n_rows || (!tw->d_rsel && !tw->d_rblk)) return 0;
  const uint64_t all = 1ull << (2 * tw->k);
  hi = std::min(hi, all);
  lo = std::min(lo, hi);
  const bool slice = lo != 0 || hi != all;
  const uint64_t n_hashes = hi - lo, bytes = n_hashes * tw->d_pad * 8;
  if (!n_hashes) return 0;
  // (a slice: whenever it fits -- the canonical k-mers thin out towards the high hashes, and a rank must not be the slow one for that)
  if (mode == 2 && !slice && (double)tw->n_rows < 0.45 * (double)n_hashes) return 0;
  size_t free_b = 0, total_b = 0;
  KPOP_HIP(hipMemGetInfo(&free_b, &total_b));
  if ((double)bytes > 0.8 * (double)free_b) return 0;  // (an optimisation: never the reason a twister does not load)
  if (hipMalloc((void **)&tw->d_direct, bytes) != hipSuccess) {
    (void)hipGetLastError();
    tw->d_direct = nullptr;
    return 0;
  }
  tw->direct_lo = lo;
  tw->direct_hi = hi;
  TwisterView tv = view_of(tw);
  direct_rows_kernel<<<dim3((uint32_t)std::min<uint64_t>(div_up(n_hashes * tw->d_pad, 256), 1u << 20)), dim3(256), 0, st>>>(tv, lo, n_hashes, tw->d_direct);
  KPOP_LAUNCH_CHECK();
  KPOP_HIP(hipStreamSynchronize(st));
  tw->device_bytes += bytes;
  return 0;
}

}  // namespace kpop

using namespace kpop;

extern "C" int kpop_twister_free(kpop_twister *tw) {
  if (!tw) return KPOP_OK;
  if (tw->alias) {
    delete tw;
    return KPOP_OK;
  }
  if (tw->d_rows) (void)hipFree(tw->d_rows);
  if (tw->d_rsel) (void)hipFree(tw->d_rsel);
  if (tw->d_rblk) (void)hipFree(tw->d_rblk);
  if (tw->d_sorted_hash) (void)hipFree(tw->d_sorted_hash);
  if (tw->d_direct) (void)hipFree(tw->d_direct);
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

extern "C" int kpop_twister_direct_bytes(const kpop_twister *tw, uint64_t *bytes) {
  if (!tw || !bytes) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_direct_bytes: null argument");
  *bytes = tw->d_direct ? (tw->direct_hi - tw->direct_lo) * tw->d_pad * 8 : 0;
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
  // rank of every column's k-mer; of several columns with one name the LAST wins (Hashtbl.add shadows,
  // lib/Twister.ml:73-76), the others can never be looked up and are dropped
  const uint64_t lim = kmer_mask(k);
  std::vector<std::pair<uint64_t, uint32_t>> hc(n_cols);
  for (uint64_t c = 0; c < n_cols; ++c) {
    if (col_hash[c] > lim) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_load: a column hash does not fit k=%d", k);
    hc[c] = {col_hash[c], (uint32_t)c};
  }
  if (!std::is_sorted(hc.begin(), hc.end())) std::sort(hc.begin(), hc.end());  // KPopTwist writes its columns in ascending name order
  std::vector<uint64_t> sorted_hash;
  std::vector<uint32_t> dst_row(n_cols, kNoCol);
  sorted_hash.reserve(n_cols);
  for (uint64_t i = 0; i < n_cols; ++i) {
    if (i + 1 < n_cols && hc[i + 1].first == hc[i].first) continue;
    dst_row[hc[i].second] = (uint32_t)sorted_hash.size();
    sorted_hash.push_back(hc[i].first);
  }
  kpop_twister *tw = new kpop_twister();
  TwGuard guard{tw};
  tw->slot = current_slot();
  tw->k = k;
  tw->n_dims = n_dims;
  tw->n_cols = n_cols;
  tw->n_rows = sorted_hash.size();
  KPOP_TRY(alloc_rows(tw));
  hipStream_t st = nullptr;
  DevBuf d_hash, d_dst;
  KPOP_TRY(d_hash.alloc(tw->n_rows * 8));
  KPOP_TRY(d_dst.alloc(n_cols * 4));
  if (tw->n_rows) KPOP_HIP(hipMemcpyAsync(d_hash.p, sorted_hash.data(), tw->n_rows * 8, hipMemcpyHostToDevice, st));
  if (n_cols) KPOP_HIP(hipMemcpyAsync(d_dst.p, dst_row.data(), n_cols * 4, hipMemcpyHostToDevice, st));

  // --- rows: H2D of the dims-major matrix + device transpose into rank order.  Up to 8 GB the matrix goes over in ONE
  // copy (it is contiguous on the host; n_dims x chunks small copies with a synchronisation per chunk ran at ~6 GB/s on a
  // 0.9 GB twister, a third of what one large pageable copy gets) and is transposed from there; beyond, in slabs.
  if (n_cols) {
    const uint64_t whole = (uint64_t)n_cols * n_dims * 8;
    void *full = nullptr;
    if (whole <= (8ull << 30) && hipMalloc(&full, whole) == hipSuccess) {
      hipError_t e = hipMemcpyAsync(full, T_dims_major, whole, hipMemcpyHostToDevice, st);
      if (e == hipSuccess) {
        dim3 grid(div_up(n_cols, 32), div_up(tw->d_pad, 32));
        transpose_chunk_kernel<<<grid, dim3(256), 0, st>>>(reinterpret_cast<const double *>(full), n_cols, n_dims, tw->d_pad,
                                                           d_dst.as<uint32_t>(), tw->d_rows, 0);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      (void)hipFree(full);
      if (e != hipSuccess) KPOP_FAIL(KPOP_ERR_HIP, "kpop_twister_load: %s", hipGetErrorString(e));
    } else {
      (void)hipGetLastError();  // a failed hipMalloc of the staging copy is not an error: fall back to slabs
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
        transpose_chunk_kernel<<<grid, dim3(256), 0, st>>>(tmp.as<double>(), cn, n_dims, tw->d_pad, d_dst.as<uint32_t>(),
                                                           tw->d_rows, c0);
        KPOP_LAUNCH_CHECK();
        KPOP_HIP(hipStreamSynchronize(st));
      }
    }
  }

  // --- name -> row
  if (k <= kRankMaxK) {
    const uint64_t n_words = ((1ull << (2 * k)) + 63) / 64;
    KPOP_HIP(hipMalloc(&tw->d_rsel, n_words * sizeof(RankWord)));
    tw->device_bytes += n_words * sizeof(RankWord);
    KPOP_HIP(hipMemsetAsync(tw->d_rsel, 0, n_words * sizeof(RankWord), st));
    if (tw->n_rows) {
      rank_bits_kernel<<<dim3(std::min<uint32_t>(div_up(tw->n_rows, 256), 4096)), dim3(256), 0, st>>>(
          d_hash.as<uint64_t>(), tw->n_rows, reinterpret_cast<RankWord *>(tw->d_rsel));
      KPOP_LAUNCH_CHECK();
    }
    uint64_t total = 0;
    KPOP_TRY(finish_rank_select(tw, n_words, &total, st));
    if (total != tw->n_rows) KPOP_FAIL(KPOP_ERR_HIP, "kpop_twister_load: rank index holds %llu k-mers, expected %llu",
                                       (unsigned long long)total, (unsigned long long)tw->n_rows);
    KPOP_TRY(compact_rank_index(tw, st));
    KPOP_TRY(build_direct_rows(tw, st));
  } else {
    tw->d_sorted_hash = d_hash.as<uint64_t>();
    d_hash.p = nullptr;  // ownership moves to the twister
    tw->device_bytes += tw->n_rows * 8;
  }
  KPOP_HIP(hipStreamSynchronize(st));
  guard.tw = nullptr;
  *out = tw;
  return KPOP_OK;
}

extern "C" int kpop_twister_set_count_k(kpop_twister *tw, int k) {
  if (!tw) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_set_count_k: null twister");
  if (k < 1 || k > tw->k) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_set_count_k: k=%d outside 1..%d (the k the twister was loaded with)", k, tw->k);
  tw->hk = k;
  return KPOP_OK;
}

extern "C" int kpop_twister_synth(uint64_t seed, int k, int content, uint32_t n_dims, kpop_twister **out) {
  return kpop_twister_synth_slice(seed, k, content, n_dims, 0, ~0ull, 0, out);
}

extern "C" int kpop_twister_synth_slice(uint64_t seed, int k, int content, uint32_t n_dims, uint64_t hash_lo, uint64_t hash_hi,
                                        int acc_dim, kpop_twister **out) {
  KPOP_TRY(require_init());
  if (!out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_synth: null out");
  if (k < 1 || k > kRankMaxK)
    KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_twister_synth: k=%d (dense synthetic twister needs k <= %d)", k, kRankMaxK);
  if (content != KPOP_DNA_DS && content != KPOP_DNA_SS) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_synth: content");
  if (n_dims == 0) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_synth: n_dims must be positive");
  const uint64_t n_hashes = 1ull << (2 * k);
  hash_hi = std::min(hash_hi, n_hashes);
  hash_lo = std::min(hash_lo, hash_hi);
  const bool whole = hash_lo == 0 && hash_hi == n_hashes;
  uint64_t n_cols = n_hashes;
  if (content == KPOP_DNA_DS) n_cols = (k % 2 == 0) ? (n_hashes + (1ull << k)) / 2 : n_hashes / 2;
  kpop_twister *tw = new kpop_twister();
  TwGuard guard{tw};
  tw->slot = current_slot();
  tw->k = k;
  tw->n_dims = n_dims + (acc_dim ? 1u : 0u);
  hipStream_t st = nullptr;
  const uint64_t n_words = (n_hashes + 63) / 64;
  KPOP_HIP(hipMalloc(&tw->d_rsel, n_words * sizeof(RankWord)));
  tw->device_bytes += n_words * sizeof(RankWord);
  rank_bits_canonical_kernel<<<dim3(std::min<uint32_t>(div_up(n_words, 256), 1u << 16)), dim3(256), 0, st>>>(
      k, content, n_words, hash_lo, hash_hi, reinterpret_cast<RankWord *>(tw->d_rsel));
  KPOP_LAUNCH_CHECK();
  uint64_t total = 0;
  KPOP_TRY(finish_rank_select(tw, n_words, &total, st));
  if (whole && total != n_cols)
    KPOP_FAIL(KPOP_ERR_HIP, "kpop_twister_synth: enumerated %llu k-mers, expected %llu", (unsigned long long)total,
              (unsigned long long)n_cols);
  if (total >= 0xFFFFFFFFull) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_twister_synth: too many columns");
  tw->n_cols = tw->n_rows = total;
  KPOP_TRY(alloc_rows(tw));
  synth_rows_kernel<<<dim3(capped_grid(div_up(n_hashes, 4))), dim3(256), 0, st>>>(
      seed, reinterpret_cast<const RankWord *>(tw->d_rsel), n_hashes, n_dims, tw->n_dims, tw->d_pad, tw->d_rows);
  KPOP_LAUNCH_CHECK();
  KPOP_HIP(hipStreamSynchronize(st));
  KPOP_TRY(compact_rank_index(tw, st));
  KPOP_TRY(build_direct_rows(tw, st, hash_lo, hash_hi));
  guard.tw = nullptr;
  *out = tw;
  return KPOP_OK;
}

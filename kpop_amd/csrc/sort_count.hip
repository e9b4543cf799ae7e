// sort_count.hip -- k-mer counting by device-wide sort, for what one wavefront
// cannot hold: sequences of more than 512 windows in -L mode (assembled
// genomes) and the merged -l spectrum (bin/KPopCount.ml:60).
//
//   window_keys_kernel   every window -> composite key (spectrum id << 2k | hash),
//                        invalid windows -> all-ones sentinel
//   radix_sort_u64       radix_sort.h, ceil((2k + id bits)/8) passes
//   head flags + scan    run-length collapse -> distinct keys + first positions
//   spectrum_bounds      per spectrum lower_bound into the distinct keys -> CSR offsets
//
// Integer work end to end: bit-exact against the oracle.
#include <algorithm>
#include <vector>

#include "kmer.h"
#include "radix_sort.h"
#include "scan.h"

#include "lookback.h"
#include <string.h>

#include "sort_count.h"

namespace kpop {

constexpr uint32_t kKeySeg = 16384;  // windows per block

// grid (n_reads, max_seg); woff[r] = first key slot of read r
template <typename H>
__global__ __launch_bounds__(256) void window_keys_kernel(const uint8_t *__restrict__ bases,
                                                          const uint64_t *__restrict__ offsets,
                                                          const uint64_t *__restrict__ woff, int k, int content,
                                                          int per_read, uint32_t first_id, uint64_t *__restrict__ keys,
                                                          uint32_t n_reads, uint32_t max_seg) {
  const uint64_t n_pairs = (uint64_t)n_reads * max_seg;  // (read, segment) pairs dealt round-robin, see count_twist.hip
  for (uint64_t pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
  const uint32_t r = (uint32_t)(pair / max_seg);
  const uint64_t off = offsets[r], len = offsets[r + 1] - off;
  if (len < (uint64_t)k) continue;
  const uint64_t n_win = len - k + 1;
  const uint64_t w0 = (uint64_t)(pair % max_seg) * kKeySeg;
  if (w0 >= n_win) continue;
  const uint64_t w1 = min(n_win, w0 + kKeySeg);
  const uint8_t *seq = bases + off;
  const bool protein = content == KPOP_PROTEIN;
  const int sb = symbol_bits(content), shift = sb * (k - 1);
  const uint64_t id = per_read ? ((uint64_t)(first_id + r) << (sb * k)) : 0ull;
  for (uint64_t w = w0 + threadIdx.x; w < w1; w += 256) {
    H fwd = 0, rc = 0;
    bool good = true;
    for (int j = 0; j < k; ++j) {
      if (protein) {
        const uint32_t c = protein_code(seq[w + j]);
        good = good && (c < 20u);
        fwd = (fwd << 5) | (H)(c & 31u);
      } else {
        const uint32_t c = base_code(seq[w + j]);
        good = good && (c < 4u);
        fwd = (fwd << 2) | (H)(c & 3u);
        rc = (rc >> 2) | ((H)(3u - (c & 3u)) << shift);
      }
    }
    const uint64_t h = (uint64_t)((content == KPOP_DNA_DS && rc < fwd) ? rc : fwd);
    keys[woff[r] + w] = good ? (id | h) : ~0ull;
  }
  }
}

struct HeadFlag {
  const uint64_t *keys;
  __device__ uint32_t operator()(uint64_t i) const {
    const uint64_t x = keys[i];
    return (x != ~0ull && (i == 0 || keys[i - 1] != x)) ? 1u : 0u;
  }
};
struct StoreHeads {
  const uint64_t *keys;
  uint64_t *uniq;
  uint64_t *start;
  __device__ void operator()(uint64_t i, uint64_t prefix, uint32_t flag) const {
    if (flag) {
      uniq[prefix] = keys[i];
      start[prefix] = i;
    }
  }
};
struct ValidFlag {
  const uint64_t *keys;
  __device__ uint32_t operator()(uint64_t i) const { return keys[i] != ~0ull ? 1u : 0u; }
};
struct Discard {
  __device__ void operator()(uint64_t, uint64_t, uint32_t) const {}
};

// count[u] = start[u+1] - start[u] (last: n_valid - start), hash[u] = uniq & mask
__global__ void finish_spectra_kernel(const uint64_t *__restrict__ uniq, const uint64_t *__restrict__ start,
                                      const uint64_t *__restrict__ n_unique_p, const uint64_t *__restrict__ n_valid_p,
                                      uint64_t hash_mask, uint64_t *__restrict__ out_hash,
                                      uint32_t *__restrict__ out_count) {
  const uint64_t nu = *n_unique_p, nv = *n_valid_p;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < nu; u += stride) {
    const uint64_t e = (u + 1 < nu) ? start[u + 1] : nv;
    out_hash[u] = uniq[u] & hash_mask;
    out_count[u] = (uint32_t)(e - start[u]);
  }
}

// offsets[s] = first distinct key whose spectrum id is >= first_id + s (s = 0..n_spectra)
__global__ void spectrum_bounds_kernel(const uint64_t *__restrict__ uniq, const uint64_t *__restrict__ n_unique_p,
                                       uint32_t n_spectra, uint32_t first_id, int hash_bits, uint64_t *__restrict__ offsets) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > n_spectra) return;
  const uint64_t nu = *n_unique_p;
  const uint64_t target = (uint64_t)(first_id + s) << hash_bits;
  uint64_t lo = 0, hi = nu;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (uniq[mid] < target) lo = mid + 1; else hi = mid;
  }
  offsets[s] = lo;
}

// ---------------------------------------------------------------------------
// merged spectrum (-l, bin/KPopCount.ml:60) as a histogram: when every hash fits kHistMaxBits bits the table of all
// 2^bits counters (u32; 67 MB at k = 12, 268 MB at k = 13) takes one atomic add per window, and the spectrum is its
// non-zero entries in index order = ascending hash order.  One pass over the bases, no key array, no sort passes.
// The adds execute at the memory side (device-scope atomics on a table that all eight XCDs update), so they go out
// straight from the hashing lanes: staging them through per-block LDS tables first only pays when a block sees the
// same k-mer often, which 2^24 bins against a few thousand windows per block rules out.
// ---------------------------------------------------------------------------
constexpr int kHistMaxBits = 26;

template <typename H>
__global__ __launch_bounds__(256) void window_hist_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets,
                                                          int k, int content, uint32_t *__restrict__ table, uint32_t n_reads,
                                                          uint32_t max_seg) {
  const uint64_t n_pairs = (uint64_t)n_reads * max_seg;
  for (uint64_t pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
    const uint32_t r = (uint32_t)(pair / max_seg);
    const uint64_t off = offsets[r], len = offsets[r + 1] - off;
    if (len < (uint64_t)k) continue;
    const uint64_t n_win = len - k + 1;
    const uint64_t w0 = (uint64_t)(pair % max_seg) * kKeySeg;
    if (w0 >= n_win) continue;
    const uint64_t w1 = min(n_win, w0 + kKeySeg);
    const uint8_t *seq = bases + off;
    const bool protein = content == KPOP_PROTEIN;
    const int sb = symbol_bits(content), shift = sb * (k - 1);
    for (uint64_t w = w0 + threadIdx.x; w < w1; w += 256) {
      H fwd = 0, rc = 0;
      bool good = true;
      for (int j = 0; j < k; ++j) {
        if (protein) {
          const uint32_t c = protein_code(seq[w + j]);
          good = good && (c < 20u);
          fwd = (fwd << 5) | (H)(c & 31u);
        } else {
          const uint32_t c = base_code(seq[w + j]);
          good = good && (c < 4u);
          fwd = (fwd << 2) | (H)(c & 3u);
          rc = (rc >> 2) | ((H)(3u - (c & 3u)) << shift);
        }
      }
      if (good) atomicAdd(&table[(content == KPOP_DNA_DS && rc < fwd) ? rc : fwd], 1u);
    }
  }
}

// short reads: one wavefront per read, windows hashed the way the per-read kernels hash them would cost LDS staging for
// nothing here; a read's windows are spread over the lanes of its wave instead (one read per wave keeps the offsets
// loads wave-uniform)
template <typename H>
__global__ __launch_bounds__(256) void read_hist_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets,
                                                        int k, int content, uint32_t *__restrict__ table, uint32_t n_reads) {
  const int lane = threadIdx.x & 63;
  const uint32_t waves = gridDim.x * 4;
  const bool protein = content == KPOP_PROTEIN;
  const int sb = symbol_bits(content), shift = sb * (k - 1);
  for (uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n_reads; r += waves) {
    const uint64_t off = offsets[r], len = offsets[r + 1] - off;
    if (len < (uint64_t)k) continue;
    const uint64_t n_win = len - k + 1;
    const uint8_t *seq = bases + off;
    for (uint64_t w = lane; w < n_win; w += 64) {
      H fwd = 0, rc = 0;
      bool good = true;
      for (int j = 0; j < k; ++j) {
        if (protein) {
          const uint32_t c = protein_code(seq[w + j]);
          good = good && (c < 20u);
          fwd = (fwd << 5) | (H)(c & 31u);
        } else {
          const uint32_t c = base_code(seq[w + j]);
          good = good && (c < 4u);
          fwd = (fwd << 2) | (H)(c & 3u);
          rc = (rc >> 2) | ((H)(3u - (c & 3u)) << shift);
        }
      }
      if (good) atomicAdd(&table[(content == KPOP_DNA_DS && rc < fwd) ? rc : fwd], 1u);
    }
  }
}

// (the padded LDS index and the register-blocked network pass of count_block_kernel, further down, are shared with the
// chunk sort of the merged histogram)
constexpr uint32_t kBlockSortMax = 32768;
#define KEY(i) ((i) + ((i) >> 5))  // LDS index of key i: one word of padding per 32

// one pass of the register-blocked bitonic network (count_block_kernel): 2^GB keys per (virtual) thread
template <int GB>
__device__ __forceinline__ void bitonic_group_pass(uint32_t *s_key, uint32_t NP, uint32_t sz, int h, int lo) {
  constexpr int E = 1 << GB;
  const uint32_t n_groups = NP >> GB;
  for (uint32_t vt = threadIdx.x; vt < n_groups; vt += 1024) {
    const uint32_t base = ((vt >> lo) << (h + 1)) | (vt & ((1u << lo) - 1u));
    const bool asc = (base & sz) == 0;
    uint32_t rk[E];
#pragma unroll
    for (int e = 0; e < E; ++e) rk[e] = s_key[KEY(base | ((uint32_t)e << lo))];
#pragma unroll
    for (int j = GB - 1; j >= 0; --j) {
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (!(e & (1 << j))) {
          const uint32_t x = rk[e], y = rk[e | (1 << j)];
          const uint32_t mn = min(x, y), mx = max(x, y);
          rk[e] = asc ? mn : mx;
          rk[e | (1 << j)] = asc ? mx : mn;
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) s_key[KEY(base | ((uint32_t)e << lo))] = rk[e];
  }
}


// ---------------------------------------------------------------------------
// LDS-staged forms of the merged histogram (north_star: "LDS-staged per-block counts").
//
// (a) SMALL TABLES (hash_bits <= 14: DNA k <= 7, 64 KB of counters): every block owns a private copy of the whole table in
//     LDS, adds its windows there (ds_add_u32) and hands each non-zero counter to the global table with ONE atomic at the
//     end.  Without this, 150 million windows of 5,000 genomes hammer 8,192 global counters.
// (b) ASSEMBLIES OF ONE ORGANISM (BASELINE config 3's case): a block takes the SAME stretch of kCombSeg windows of
//     kCombReads consecutive sequences -- near-identical genomes put the same k-mers there -- and counts them in an LDS
//     table of 8,192 (hash, count) slots (open addressing: ds_cmpst on the hash, ds_add on the count, eight probes, then
//     straight to the global table); the flush issues ONE global atomic per distinct k-mer of the chunk: a 32-byte sector
//     per distinct k-mer instead of one per window.  (Sorting the chunk's 32,768 hashes in LDS instead was measured:
//     2.97 ms against 7.04 direct on 5,000 mutants of one genome; the table does the same combining without the 120
//     levels of the network.)  A block whose first chunk shows no repetition (unrelated sequences: distinct > 3/4 of the
//     windows) stops staging and adds its remaining windows directly, as window_hist_kernel does.
// ---------------------------------------------------------------------------
constexpr int kHistLdsBits = 14;

template <int SB>
__global__ __launch_bounds__(1024) void window_hist_lds_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets,
                                                               int k, int content, int hash_bits_, uint32_t *__restrict__ table,
                                                               uint32_t n_reads, uint32_t max_seg) {
  extern __shared__ uint32_t s_tab[];
  const uint32_t n_bins = 1u << hash_bits_;
  for (uint32_t b = threadIdx.x; b < n_bins; b += 1024) s_tab[b] = 0;
  __syncthreads();
  const uint64_t n_pairs = (uint64_t)n_reads * max_seg;
  const int shift = SB * (k - 1);
  constexpr uint32_t kSym = (1u << SB) - 1u, kValid = SB == 2 ? 4u : 20u;
  const uint32_t mask = (uint32_t)bits_mask(SB * k);
  for (uint64_t pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
    const uint32_t r = (uint32_t)(pair / max_seg);
    const uint64_t off = offsets[r], len = offsets[r + 1] - off;
    if (len < (uint64_t)k) continue;
    const uint64_t n_win = len - k + 1;
    const uint64_t w0 = (uint64_t)(pair % max_seg) * kKeySeg;
    if (w0 >= n_win) continue;
    const uint64_t w1 = min(n_win, w0 + kKeySeg);
    const uint8_t *seq = bases + off;
    // a thread rolls the hash over a run of consecutive windows
    const uint32_t per = (uint32_t)((w1 - w0 + 1023) / 1024);
    const uint64_t a = w0 + (uint64_t)threadIdx.x * per, b = min(w1, a + per);
    if (a >= b) continue;
    uint32_t fwd = 0, rc = 0;
    int run = 0;
    for (int j = 0; j < k - 1; ++j) {
      const uint32_t c = SB == 2 ? base_code(seq[a + j]) : protein_code(seq[a + j]);
      fwd = ((fwd << SB) | (c & kSym)) & mask;
      if (SB == 2) rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
      run = c < kValid ? run + 1 : 0;
    }
    for (uint64_t w = a; w < b; ++w) {
      const uint32_t c = SB == 2 ? base_code(seq[w + k - 1]) : protein_code(seq[w + k - 1]);
      fwd = ((fwd << SB) | (c & kSym)) & mask;
      if (SB == 2) rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
      run = c < kValid ? run + 1 : 0;
      if (run >= k) atomicAdd(&s_tab[(SB == 2 && content == KPOP_DNA_DS && rc < fwd) ? rc : fwd], 1u);
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < n_bins; b += 1024) {
    const uint32_t c = s_tab[b];
    if (c) atomicAdd(&table[b], c);
  }
}

constexpr uint32_t kCombSeg = 1024, kCombReads = 64;  // a chunk: the same 1,024 windows of 64 consecutive sequences (128: 0.65 -> 2.1 ms on 5,000 mutants -- twice the lanes of a wave on one LDS word)
constexpr uint32_t kCombTpr = 1024 / kCombReads;          // threads a sequence
constexpr uint32_t kCombSlots = 8192;                  // LDS table of (hash, count): 64 KB, two blocks a CU
constexpr int kCombProbes = 8;

template <int SB>
__global__ __launch_bounds__(1024) void window_hist_combine_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets,
                                                                   int k, int content, uint32_t *__restrict__ table, uint32_t n_reads,
                                                                   uint32_t max_seg, int always_stage) {
  __shared__ uint32_t s_hash[kCombSlots], s_cnt[kCombSlots];
  __shared__ uint32_t s_distinct, s_valid;
  const uint32_t n_groups = (n_reads + kCombReads - 1) / kCombReads;
  const uint64_t n_chunks = (uint64_t)n_groups * max_seg;
  const int shift = SB * (k - 1);
  constexpr uint32_t kSym = (1u << SB) - 1u, kValid = SB == 2 ? 4u : 20u, kEmpty = 0xFFFFFFFFu;
  const uint32_t mask = (uint32_t)bits_mask(SB * k);
  constexpr uint32_t per_h = kCombSeg / kCombTpr;  // kCombTpr threads a sequence, consecutive windows a thread
  for (uint32_t q = threadIdx.x; q < kCombSlots; q += 1024) {
    s_hash[q] = kEmpty;
    s_cnt[q] = 0;
  }
  bool direct = false;
  for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    // chunks are dealt with the groups of sequences fastest: the blocks running together work on one stretch of all of them
    const uint32_t seg = (uint32_t)(chunk / n_groups), grp = (uint32_t)(chunk % n_groups);
    const uint32_t r = grp * kCombReads + threadIdx.x / kCombTpr;
    uint64_t len = 0, off = 0;
    if (r < n_reads) {
      off = offsets[r];
      len = offsets[r + 1] - off;
    }
    const uint64_t n_win = len >= (uint64_t)k ? len - k + 1 : 0;
    const uint64_t w0 = (uint64_t)seg * kCombSeg + (uint64_t)(threadIdx.x % kCombTpr) * per_h;
    const uint8_t *seq = bases + off;
    uint32_t fwd = 0, rc = 0;
    int run = 0;
    if (w0 < n_win)
      for (int j = 0; j < k - 1; ++j) {
        const uint32_t c = SB == 2 ? base_code(seq[w0 + j]) : protein_code(seq[w0 + j]);
        fwd = ((fwd << SB) | (c & kSym)) & mask;
        if (SB == 2) rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
        run = c < kValid ? run + 1 : 0;
      }
    if (!direct) {
      if (threadIdx.x == 0) {
        s_distinct = 0;
        s_valid = 0;
      }
      __syncthreads();  // (also: the table is empty -- initialised above, or emptied by the previous chunk's flush)
    }
    uint32_t n_ok = 0, n_new = 0;
    for (uint32_t i = 0; i < per_h; ++i) {
      const uint64_t w = w0 + i;
      if (w >= n_win) break;
      const uint32_t c = SB == 2 ? base_code(seq[w + k - 1]) : protein_code(seq[w + k - 1]);
      fwd = ((fwd << SB) | (c & kSym)) & mask;
      if (SB == 2) rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
      run = c < kValid ? run + 1 : 0;
      if (run < k) continue;
      const uint32_t key = (SB == 2 && content == KPOP_DNA_DS && rc < fwd) ? rc : fwd;
      if (direct) {
        atomicAdd(&table[key], 1u);
        continue;
      }
      ++n_ok;
      uint32_t slot = (key * 2654435761u) >> 19;  // 13 bits
      bool placed = false;
#pragma unroll 1
      for (int t = 0; t < kCombProbes; ++t) {
        const uint32_t prev = atomicCAS(&s_hash[slot], kEmpty, key);
        if (prev == kEmpty || prev == key) {
          atomicAdd(&s_cnt[slot], 1u);
          n_new += prev == kEmpty;
          placed = true;
          break;
        }
        slot = (slot + 1) & (kCombSlots - 1);
      }
      if (!placed) {  // the neighbourhood is taken by other k-mers: this window goes straight to the global table
        atomicAdd(&table[key], 1u);
        ++n_new;
      }
    }
    if (direct) continue;
    if (n_ok) atomicAdd(&s_valid, n_ok);
    if (n_new) atomicAdd(&s_distinct, n_new);
    __syncthreads();
    // one global atomic per distinct k-mer of the chunk, and the table is empty again
    for (uint32_t q = threadIdx.x; q < kCombSlots; q += 1024) {
      const uint32_t h = s_hash[q];
      if (h != kEmpty) {
        atomicAdd(&table[h], s_cnt[q]);
        s_hash[q] = kEmpty;
        s_cnt[q] = 0;
      }
    }
    // unrelated sequences repeat nothing inside a chunk: staging then only costs, and the block's later chunks go direct
    if (!always_stage && (uint64_t)s_distinct * 4 > (uint64_t)s_valid * 3) direct = true;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// (c) INPUTS THAT DO NOT REPEAT (a read set, unrelated genomes): PARTITION, THEN COUNT IN LDS.  Every add of the forms above
//     that reaches the global table is a scattered 4-byte memory-side atomic paying a 32-byte sector (0.03 of HBM on 100k
//     random reads, traffic 3.7x the algorithmic bytes).  Instead:
//       sizes      every window's hash -> its bucket (the top bits: a bucket = 2^LB consecutive bins, <= 32,768 = 128 KB of
//                  u32 counters), counted per block in LDS, one global add per (block, bucket);
//       offsets    exclusive scan of the (at most 2,048) bucket sizes: every bucket's stretch of the entry array, exactly;
//       partition  the hashes again, a block collecting up to 32,768 of them in LDS before it reserves room in every bucket
//                  with ONE atomic and writes each key's low LB bits (a u16) into its bucket's stretch;
//       count      one block per bucket: its entries (u16, read in order) counted into an LDS table (ds_add_u32), the
//                  bucket's range of the global table WRITTEN, dense -- no global atomic on the table, and no memset of it.
//     The sequences are cut into items of at most 512 windows (host-built list); a wavefront takes an item at a time, a lane
//     eight consecutive windows of it (rolling hash).  bin/KPopCount.ml:38,60.
// ---------------------------------------------------------------------------
struct HistItem {
  uint32_t r, seg;
};
constexpr uint32_t kPartItem = 512;    // windows an item
constexpr uint32_t kPartQuota = 2048;  // keys a wavefront collects per round (16 wavefronts: 32,768 keys, 128 KB)
constexpr int kPartMaxLB = 15;

static inline int part_bucket_bits(int hb) { return hb - 9 < kPartMaxLB ? (hb - 9 < 4 ? 4 : hb - 9) : kPartMaxLB; }  // LB: 512 buckets up to 24 hash bits, then buckets of 32,768 bins

// the item's windows of this lane (eight consecutive ones): keys[i] valid where bit i of the return value is set.  The lane's
// 8 + k - 1 symbols are loaded up front, every load in flight at once (k - 1 <= 12 here: hashes of up to 26 bits) -- as a loop of
// dependent byte loads this was 19 round trips an item, and the partition's three passes took 0.33 ms on 100k reads
template <int SB>
__device__ __forceinline__ uint32_t part_item_keys(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets, HistItem it, int k, int content,
                                                   int lane, uint32_t (&keys)[8]) {
  const uint64_t off = offsets[it.r], len = offsets[it.r + 1] - off;
  const uint64_t n_win = len >= (uint64_t)k ? len - k + 1 : 0;
  const uint64_t w0 = (uint64_t)it.seg * kPartItem + (uint64_t)lane * 8;
#pragma unroll
  for (uint32_t i = 0; i < 8; ++i) keys[i] = 0;
  if (w0 >= n_win) return 0u;
  const uint32_t nv = (uint32_t)min<uint64_t>(8, min<uint64_t>(n_win, (uint64_t)(it.seg + 1) * kPartItem) - w0);
  const uint8_t *seq = bases + off + w0;
  constexpr uint32_t kSym = (1u << SB) - 1u, kValid = SB == 2 ? 4u : 20u;
  constexpr int kMaxSyms = 8 + 12;  // (k <= 13)
  const uint32_t mask = (uint32_t)bits_mask(SB * k);
  const int shift = SB * (k - 1);
  const int ns = (int)nv + k - 1;  // symbols of this lane
  uint32_t sym[kMaxSyms];
#pragma unroll
  for (int j = 0; j < kMaxSyms; ++j) sym[j] = j < ns ? (uint32_t)seq[j] : 0u;
  uint32_t fwd = 0, rc = 0, ok = 0;
  int run = 0;
#pragma unroll
  for (int j = 0; j < kMaxSyms; ++j) {
    if (j < ns) {
      const uint32_t c = SB == 2 ? base_code(sym[j]) : protein_code(sym[j]);
      fwd = ((fwd << SB) | (c & kSym)) & mask;
      if (SB == 2) rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
      run = c < kValid ? run + 1 : 0;
      const int i = j - (k - 1);  // the window this symbol ends
      if (i >= 0) {
        const uint32_t key = (SB == 2 && content == KPOP_DNA_DS && rc < fwd) ? rc : fwd;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q == i) keys[q] = key;
        ok |= run >= k ? (1u << i) : 0u;
      }
    }
  }
  return ok;
}

// DNA items without any rolling: the wavefront packs the item's bases ONCE into LDS (two bits a base, first base in the most
// significant bits: the hash's own convention, kmer.h; one more bit a base for "not ACGT"), then a window's hash is a 2k-bit
// field of that stream, its reverse complement a bit reversal, its validity k zero bits -- ~16 vector instructions a window
// with every lane busy, against ~29 per window on the 18 lanes a 150-base read keeps busy when each lane rolls eight windows.
// Lane l gets windows l, 64 + l, ... of the item.  s_w: kPartStageWords words of the wavefront's own.
// A DNA item is at most 496 windows -- 496 + k - 1 <= 508 bases: eight bytes a lane, ONE load instruction each -- and carries
// its own address and size, so that its bytes are one dependent load behind the item list: a wavefront has FOUR items' bytes
// in flight before it packs the first (an item at a time, the list -> offsets -> bases chain was ~4 us of latency per item
// and all the kernels' time: 0.28 ms for the sizes of 5,000 genomes).
struct HistItemD {
  uint64_t off;  // the item's first base in `bases`
  uint32_t nw;   // its windows
  uint32_t pad;
};
constexpr uint32_t kPartItemD = 496;
constexpr uint32_t kPartCodeWords = 36, kPartInvWords = 20, kPartStageWords = kPartCodeWords + kPartInvWords;
__device__ __forceinline__ void part_issue(const uint8_t *__restrict__ bases, const HistItemD &it, int k, int lane, uint32_t (&raw)[8]) {
  const uint32_t nb = it.nw ? it.nw + (uint32_t)k - 1 : 0u, b0 = (uint32_t)lane * 8u;
  const uint8_t *seq = bases + it.off + b0;
#pragma unroll
  for (uint32_t i = 0; i < 8; ++i) raw[i] = b0 + i < nb ? (uint32_t)seq[i] : 0u;
}
__device__ __forceinline__ uint32_t part_finish(const uint32_t (&raw)[8], uint32_t nw, int k, int content, int lane, uint32_t (&keys)[8], uint32_t *s_w) {
  uint8_t *s_code = reinterpret_cast<uint8_t *>(s_w);
  uint8_t *s_inv = reinterpret_cast<uint8_t *>(s_w + kPartCodeWords);
  __builtin_amdgcn_wave_barrier();
  {
    // (bytes past the item's bases are zeros here: "invalid"; what lies past the 512 staged bases in LDS is stale, and no window
    // of this item reads it -- a window's 64-bit fields reach up to seven bytes further, and those bits are shifted or masked out)
    uint32_t v = 0, inv = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
      const uint32_t c = base_code(raw[i]);
      v |= (c & 3u) << (14u - 2u * i);
      inv |= (c > 3u ? 1u : 0u) << i;
    }
    s_code[2 * lane] = (uint8_t)(v >> 8);
    s_code[2 * lane + 1] = (uint8_t)v;
    s_inv[lane] = (uint8_t)inv;
  }
  __builtin_amdgcn_wave_barrier();
  const uint32_t mask = (uint32_t)bits_mask(2 * k), kmask = (1u << k) - 1u;
  uint32_t ok = 0;
#pragma unroll
  for (uint32_t t = 0; t < 8; ++t) keys[t] = 0;
#pragma unroll
  for (uint32_t t = 0; t < 8; ++t) {
    if (t * 64u >= nw) break;  // (uniform: a 150-base read has three rounds of windows, not eight)
    const uint32_t w = t * 64u + (uint32_t)lane;
    const uint32_t B = w >> 2;  // byte of the code stream the window starts in
    const uint64_t x = ((uint64_t)__builtin_bswap32(s_w[B >> 2]) << 32) | __builtin_bswap32(s_w[(B >> 2) + 1]);
    const uint32_t fwd = (uint32_t)(x >> (64u - (8u * (B & 3u) + 2u * (w & 3u)) - 2u * (uint32_t)k)) & mask;
    uint32_t r = __brev(fwd);
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    const uint32_t rc = ((~r) >> (32u - 2u * (uint32_t)k)) & mask;
    const uint64_t y = (uint64_t)s_w[kPartCodeWords + (w >> 5)] | ((uint64_t)s_w[kPartCodeWords + (w >> 5) + 1] << 32);
    const bool good = w < nw && (((uint32_t)(y >> (w & 31u))) & kmask) == 0u;
    keys[t] = (content == KPOP_DNA_DS && rc < fwd) ? rc : fwd;
    ok |= good ? (1u << t) : 0u;
  }
  return ok;
}

template <int SB>
__global__ __launch_bounds__(1024) void hist_part_sizes_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets, int k, int content,
                                                               const HistItem *__restrict__ items, uint64_t n_items, int LB, uint32_t n_buckets,
                                                               uint32_t *__restrict__ g_size) {
  extern __shared__ uint32_t s_cnt[];
  __shared__ uint32_t s_stage[16][kPartStageWords];
  for (uint32_t b = threadIdx.x; b < n_buckets; b += 1024) s_cnt[b] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (SB == 2) {
    const HistItemD *itd = reinterpret_cast<const HistItemD *>(items);
    for (uint64_t i = ((uint64_t)blockIdx.x * 16 + wv) * 4; i < n_items; i += (uint64_t)gridDim.x * 16 * 4) {
      HistItemD h[4];
      uint32_t raw[4][8];
#pragma unroll
      for (int q = 0; q < 4; ++q) h[q] = i + q < n_items ? itd[i + q] : HistItemD{0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q) part_issue(bases, h[q], k, lane, raw[q]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t keys[8];
        const uint32_t ok = part_finish(raw[q], h[q].nw, k, content, lane, keys, s_stage[wv]);
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j)
          if ((ok >> j) & 1u) atomicAdd(&s_cnt[keys[j] >> LB], 1u);
      }
    }
  } else {
    for (uint64_t i = (uint64_t)blockIdx.x * 16 + wv; i < n_items; i += (uint64_t)gridDim.x * 16) {
      uint32_t keys[8];
      const uint32_t ok = part_item_keys<SB>(bases, offsets, items[i], k, content, lane, keys);
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j)
        if ((ok >> j) & 1u) atomicAdd(&s_cnt[keys[j] >> LB], 1u);
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < n_buckets; b += 1024) {
    const uint32_t c = s_cnt[b];
    if (c) atomicAdd(&g_size[b], c);
  }
}

// exclusive scan of at most 2,048 bucket sizes (one block): off[0..n], and the cursors the partition pass advances
// scale > 0: g_size counts a SAMPLE of the items (one in `scale`); a bucket gets room for scale x 5/4 times that + margin
__global__ __launch_bounds__(1024) void hist_part_offsets_kernel(const uint32_t *__restrict__ g_size, uint32_t n_buckets, uint64_t *__restrict__ off,
                                                                 unsigned long long *__restrict__ cursor, uint32_t scale = 0, uint32_t margin = 0) {
  __shared__ uint64_t s_w[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t b0 = threadIdx.x * 2;
  auto room = [&](uint32_t c) -> uint64_t { return scale ? ((uint64_t)c * scale * 5u) / 4u + margin : (uint64_t)c; };
  const uint64_t a = b0 < n_buckets ? room(g_size[b0]) : 0, b = b0 + 1 < n_buckets ? room(g_size[b0 + 1]) : 0;
  uint64_t incl = a + b;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t up = (uint64_t)__shfl_up((unsigned long long)incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s_w[wv] = incl;
  __syncthreads();
  uint64_t before = incl - a - b;
  for (int w = 0; w < wv; ++w) before += s_w[w];
  if (b0 < n_buckets) {
    off[b0] = before;
    cursor[b0] = before;
  }
  if (b0 + 1 < n_buckets) {
    off[b0 + 1] = before + a;
    cursor[b0 + 1] = before + a;
  }
  if (b0 + 2 >= n_buckets && b0 < n_buckets) off[n_buckets] = before + a + b;  // (the thread that holds the last bucket)
}

template <int SB>
__global__ __launch_bounds__(1024) void hist_partition_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets, int k, int content,
                                                              const HistItem *__restrict__ items, uint64_t n_items, int LB, uint32_t n_buckets,
                                                              unsigned long long *__restrict__ cursor, uint16_t *__restrict__ entries, uint32_t quota,
                                                              const uint64_t *__restrict__ room_end = nullptr, uint32_t *__restrict__ overflow = nullptr,
                                                              uint64_t trash = 0) {
  // room_end (= the offsets, from bucket 1 on): the buckets' rooms were GUESSED from a sample of the items (no counting pass over
  // all of them); a bucket whose room is exceeded raises *overflow -- the caller then counts and partitions again, exactly --
  // and its keys of this round go to the block's own stretch of a trash area behind the entries (ranks stay below the round's keys)
  // quota: keys a wavefront collects per round -- 2,048 (one block a CU), or 1,024 where the buckets are few enough (<= 512:
  // up to 24 hash bits) for TWO blocks a CU: the hashing is what the kernel spends its time on, and it wants the wavefronts
  extern __shared__ uint32_t s_part[];
  uint32_t *s_keys = s_part;                           // [16][quota]
  uint32_t *s_cnt = s_part + 16 * quota;               // [n_buckets]
  unsigned long long *s_base = reinterpret_cast<unsigned long long *>(s_cnt + n_buckets + (n_buckets & 1u));  // [n_buckets]
  __shared__ unsigned long long s_next;
  __shared__ uint32_t s_used[16];
  __shared__ uint32_t s_stage[16][kPartStageWords];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // a block owns a contiguous run of items, each of its wavefronts a sixteenth of that; a wavefront fills its quota of a round with
  // whole items -- it knows an item's windows before it hashes them, so a round of 150-base reads is 14 reads a wavefront (1,946
  // keys), not the 12 that "room for a full item" allowed: two rounds a block on 100k reads instead of three
  const uint64_t i0 = n_items * blockIdx.x / gridDim.x, i1 = n_items * (blockIdx.x + 1) / gridDim.x;
  const uint64_t per_wave = (i1 - i0 + 15) / 16;
  uint64_t i = min(i1, i0 + (uint64_t)wv * per_wave);
  const uint64_t i_end = min(i1, i + per_wave);
  if (threadIdx.x == 0) s_next = 0;
  const uint32_t low = (1u << LB) - 1u;
  for (;;) {
    for (uint32_t b = threadIdx.x; b < n_buckets; b += 1024) s_cnt[b] = 0;
    if (threadIdx.x == 0) s_next = 0;  // (wavefronts that still have items after this round)
    __syncthreads();
    uint32_t used = 0;
    auto append = [&](uint32_t ok, const uint32_t (&keys)[8]) {
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        const bool v = (ok >> j) & 1u;
        const uint64_t m = __ballot(v);
        if (v) {
          s_keys[wv * quota + used + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = keys[j];
          atomicAdd(&s_cnt[keys[j] >> LB], 1u);
        }
        used += (uint32_t)__popcll(m);
      }
    };
    if (SB == 2) {
      const HistItemD *itd = reinterpret_cast<const HistItemD *>(items);
      bool full = false;
      while (i < i_end && !full) {
        HistItemD h[4];
        uint32_t raw[4][8];
        uint32_t nfit = 0, room = quota - used;
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // (uniform) the items of this batch that still fit the round, in order
          h[q] = i + q < i_end ? itd[i + q] : HistItemD{0, 0xFFFFFFFFu, 0};
          if (nfit == (uint32_t)q && h[q].nw <= room) {
            room -= h[q].nw;
            ++nfit;
          }
        }
        if (nfit < 4 && i + nfit < i_end) full = true;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if ((uint32_t)q < nfit) part_issue(bases, h[q], k, lane, raw[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if ((uint32_t)q < nfit) {
            uint32_t keys[8];
            const uint32_t ok = part_finish(raw[q], h[q].nw, k, content, lane, keys, s_stage[wv]);
            append(ok, keys);
          }
        i += nfit;
      }
    } else {
      while (i < i_end) {
        {
          const HistItem it = items[i];
          const uint64_t len = offsets[it.r + 1] - offsets[it.r], n_win = len >= (uint64_t)k ? len - k + 1 : 0;
          const uint64_t s_beg = (uint64_t)it.seg * kPartItem;
          const uint32_t nw = s_beg < n_win ? (uint32_t)min<uint64_t>(kPartItem, n_win - s_beg) : 0u;
          if (used + nw > quota) break;  // (uniform; an item never exceeds the quota by itself)
        }
        uint32_t keys[8];
        const uint32_t ok = part_item_keys<SB>(bases, offsets, items[i], k, content, lane, keys);
        ++i;
        append(ok, keys);
      }
    }
    if (lane == 0) {
      s_used[wv] = used;
      if (i < i_end) atomicAdd(&s_next, 1ull);
    }
    __syncthreads();
    const bool last = s_next == 0;  // (uniform: no wavefront has items left)
    for (uint32_t b = threadIdx.x; b < n_buckets; b += 1024) {
      const uint32_t c = s_cnt[b];
      if (c) {
        unsigned long long at = atomicAdd(&cursor[b], (unsigned long long)c);
        if (room_end && at + c > room_end[b]) {
          *overflow = 1u;
          at = trash + (unsigned long long)blockIdx.x * 16ull * quota;
        }
        s_base[b] = at;
      }
      s_cnt[b] = 0;
    }
    __syncthreads();
#pragma unroll 4
    for (uint32_t q = lane; q < used; q += 64) {
      const uint32_t key = s_keys[wv * quota + q], b = key >> LB;
      const uint32_t rank = atomicAdd(&s_cnt[b], 1u);
      entries[s_base[b] + rank] = (uint16_t)(key & low);
    }
    __syncthreads();
    if (last) break;
  }
}

// CSR = false: the bucket's bins into the dense table (a caller that wants the table).  CSR = true: the (hash, count) pairs of the
// bucket's non-zero bins straight into the spectrum -- the block counts them, takes its place by the decoupled look-back of
// lookback.h (buckets are dealt by a ticket so that a block's predecessors are always running) and writes them in bin order:
// the 4^k-counter table is never written, read twice more by two scans and compacted (73 us of a 100k-read call, 24 %).
template <bool CSR>
__global__ __launch_bounds__(1024) void hist_bucket_count_kernel(const uint16_t *__restrict__ entries, const uint64_t *__restrict__ off, int LB,
                                                                 uint32_t *__restrict__ table, uint32_t n_buckets, uint32_t *__restrict__ ticket,
                                                                 uint64_t *__restrict__ state, uint64_t *__restrict__ out_hash, uint32_t *__restrict__ out_count,
                                                                 uint64_t cap, uint64_t *__restrict__ total_out,
                                                                 const unsigned long long *__restrict__ ends = nullptr) {
  extern __shared__ uint32_t s_tab[];
  __shared__ uint32_t s_b, s_wsum[16];
  __shared__ uint64_t s_base;
  const uint32_t n_bins = 1u << LB;
  uint32_t b = blockIdx.x;
  if (CSR) {
    if (threadIdx.x == 0) s_b = atomicAdd(ticket, 1u);
    __syncthreads();
    b = s_b;
  }
  for (uint32_t j = threadIdx.x; j < n_bins; j += 1024) s_tab[j] = 0;
  __syncthreads();
  const uint64_t e0 = off[b], e1 = ends ? min((uint64_t)ends[b], off[b + 1]) : off[b + 1];  // (ends: rooms guessed from a sample, filled up to the cursors)
  {
    // eight entries a load (the run's body from its first 16-byte boundary on; the few entries before it and after the last whole
    // group one by one): a two-byte load a lane was 128 bytes a wavefront instruction
    const uint64_t a0 = min(e1, (e0 + 7u) & ~7ull), a1 = a0 + ((e1 - a0) & ~7ull);
    for (uint64_t i = e0 + threadIdx.x; i < a0; i += 1024) atomicAdd(&s_tab[entries[i]], 1u);
    const uint4 *v8 = reinterpret_cast<const uint4 *>(entries + a0);
    for (uint64_t g = threadIdx.x; g < (a1 - a0) / 8; g += 1024) {
      const uint4 v = v8[g];
      atomicAdd(&s_tab[v.x & 0xFFFFu], 1u);
      atomicAdd(&s_tab[v.x >> 16], 1u);
      atomicAdd(&s_tab[v.y & 0xFFFFu], 1u);
      atomicAdd(&s_tab[v.y >> 16], 1u);
      atomicAdd(&s_tab[v.z & 0xFFFFu], 1u);
      atomicAdd(&s_tab[v.z >> 16], 1u);
      atomicAdd(&s_tab[v.w & 0xFFFFu], 1u);
      atomicAdd(&s_tab[v.w >> 16], 1u);
    }
    for (uint64_t i = a1 + threadIdx.x; i < e1; i += 1024) atomicAdd(&s_tab[entries[i]], 1u);
  }
  __syncthreads();
  if (!CSR) {
    uint32_t *out = table + ((uint64_t)b << LB);
    for (uint32_t j = threadIdx.x; j < n_bins; j += 1024) out[j] = s_tab[j];
    return;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t per = n_bins / 16, w0 = (uint32_t)wv * per;  // (at least 16 bins a bucket: a wavefront's run of them)
  uint32_t cnt = 0;
  for (uint32_t i = 0; i < per; i += 64) cnt += (uint32_t)__popcll(__ballot(i + lane < per && s_tab[w0 + i + lane] != 0u));
  if (lane == 0) s_wsum[wv] = cnt;
  __syncthreads();
  if (wv == 0) {
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) total += s_wsum[w];
    const uint64_t base = lookback_exclusive(state, b, (uint64_t)total, lane, 1);
    if (lane == 0) {
      s_base = base;
      if (b == n_buckets - 1) *total_out = base + total;
    }
  }
  __syncthreads();
  uint64_t pos = s_base;
  for (int w = 0; w < wv; ++w) pos += s_wsum[w];
  for (uint32_t i = 0; i < per; i += 64) {
    const uint32_t bin = w0 + i + (uint32_t)lane;
    const uint32_t v = i + lane < per ? s_tab[bin] : 0u;
    const uint64_t m = __ballot(v != 0u);
    if (v) {
      const uint64_t at = pos + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
      if (at < cap) {  // (a spectrum beyond the caller's capacity is reported from the total, not written past it)
        out_hash[at] = ((uint64_t)b << LB) | bin;
        out_count[at] = v;
      }
    }
    pos += (uint64_t)__popcll(m);
  }
}

// Are the batch's sequences ONE organism?  (The merged count of assemblies then goes through window_hist_combine_kernel's
// LDS (hash, count) tables; a batch that is not, through the partition above.)  One block: the k-mers of sequence 0's first
// 4,096 windows in an LDS set, sixteen windows of each of up to 63 other sequences looked up; yes when half of them are there.
template <int SB>
__global__ __launch_bounds__(1024) void hist_related_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets, uint32_t n_reads, int k,
                                                            int content, uint32_t *__restrict__ flag) {
  constexpr uint32_t kTab = 8192, kEmpty = 0xFFFFFFFFu;
  __shared__ uint32_t s_tab[kTab];
  __shared__ uint32_t s_stat;
  for (uint32_t q = threadIdx.x; q < kTab; q += 1024) s_tab[q] = kEmpty;
  if (threadIdx.x == 0) s_stat = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t keys[8];
  {
    const uint32_t i = threadIdx.x;  // items 0..7 of sequence 0, a wavefront each: 8 x 512 windows
    const uint32_t ok = i < 512 ? part_item_keys<SB>(bases, offsets, HistItem{0u, i >> 6}, k, content, lane, keys) : 0u;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j)
      if ((ok >> j) & 1u) {
        uint32_t slot = (keys[j] * 2654435761u) >> 19;
        for (uint32_t t = 0; t < kTab; ++t) {
          const uint32_t prev = atomicCAS(&s_tab[slot], kEmpty, keys[j]);
          if (prev == kEmpty || prev == keys[j]) break;
          slot = (slot + 1) & (kTab - 1);
        }
      }
  }
  __syncthreads();
  {
    // thread t: sequence 1 + t / 16 (if there is one), two lanes' worth of windows of one item spread over the first 4,096
    const uint32_t r = 1 + threadIdx.x / 16, piece = threadIdx.x % 16;
    uint32_t st = 0;
    if (r < n_reads && r < 64) {
      const uint32_t ok = part_item_keys<SB>(bases, offsets, HistItem{r, piece / 2}, k, content, (int)((piece % 2) * 32 + 5), keys);
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j)
        if ((ok >> j) & 1u) {
          uint32_t slot = (keys[j] * 2654435761u) >> 19, key = s_tab[slot];
          for (uint32_t t = 0; t < kTab && key != keys[j] && key != kEmpty; ++t) {
            slot = (slot + 1) & (kTab - 1);
            key = s_tab[slot];
          }
          st += 1u + (key == keys[j] ? 65536u : 0u);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) st += (uint32_t)__shfl_xor((int)st, o, 64);
    if (lane == 0 && st) atomicAdd(&s_stat, st);
  }
  __syncthreads();
  if (threadIdx.x == 0) *flag = ((s_stat & 0xFFFFu) > 0 && (s_stat >> 16) * 2u >= (s_stat & 0xFFFFu)) ? 1u : 0u;
}

struct NonZero {
  const uint32_t *t;
  __device__ uint32_t operator()(uint64_t i) const { return t[i] ? 1u : 0u; }
};
struct StoreBins {
  const uint32_t *t;
  uint64_t *hash;
  uint32_t *count;
  __device__ void operator()(uint64_t i, uint64_t prefix, uint32_t flag) const {
    if (flag) {
      hash[prefix] = i;
      count[prefix] = t[i];
    }
  }
};

static int hist_count_device(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content, uint64_t cap,
                             SortedSpectra &S, hipStream_t st, bool exact_sizes = false) {
  const uint64_t base0 = offsets[0], n_bases = offsets[n_reads] - base0;
  std::vector<uint64_t> rel(n_reads + 1);
  uint64_t max_win = 0;
  for (uint32_t r = 0; r <= n_reads; ++r) rel[r] = offsets[r] - base0;
  uint64_t total_win = 0;
  for (uint32_t r = 0; r < n_reads; ++r) {
    const uint64_t len = rel[r + 1] - rel[r];
    const uint64_t w = len >= (uint64_t)k ? len - k + 1 : 0;
    max_win = std::max(max_win, w);
    total_win += w;
  }
  const int hb = hash_bits(k, content);
  const uint64_t n_bins = 1ull << hb;
  S.nu = 0;
  S.n_spectra = 1;
  KPOP_TRY(S.d_oo.alloc(16));
  KPOP_TRY(S.d_bases.alloc(n_bases));
  KPOP_TRY(S.d_off.alloc((uint64_t)(n_reads + 1) * 8));
  KPOP_TRY(S.d_ka.alloc(n_bins * 4));  // the table
  KPOP_TRY(S.d_sums.alloc((scan_blocks(n_bins) + 1) * 8));
  KPOP_HIP(hipMemcpyAsync(S.d_bases.p, bases + base0, n_bases, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(S.d_off.p, rel.data(), (uint64_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
  uint32_t *table = S.d_ka.as<uint32_t>();
  const int lds_mode = ctx().tune_histlds;  // 1 (default): LDS-staged where it applies; 0: round 2's direct atomics; 2: always combine chunks; 3: always partition
  const bool protein = content == KPOP_PROTEIN;
  // The partition path: hashes too wide for a private LDS table, enough windows to fill the chip twice over, fewer than 2^32
  // of them -- and nothing to gain from combining: reads (no sequence above 4,096 windows), few assemblies, or assemblies
  // that hist_related_kernel does not find to be one organism.
  auto use_partition = [&]() -> bool {
    if (!lds_mode || hb <= kHistLdsBits || hb - part_bucket_bits(hb) > 11 || total_win >= (1ull << 32) || k > 13) return false;
    if (lds_mode == 3 || lds_mode == 4) return true;
    if (lds_mode == 2 || total_win < (1u << 20)) return false;
    if (max_win <= 4096 || n_reads < 16) return true;
    DevBuf d_flag;
    if (d_flag.alloc(4)) return false;
    if (protein) hist_related_kernel<5><<<dim3(1), dim3(1024), 0, st>>>(S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), n_reads, k, content, d_flag.as<uint32_t>());
    else hist_related_kernel<2><<<dim3(1), dim3(1024), 0, st>>>(S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), n_reads, k, content, d_flag.as<uint32_t>());
    uint32_t related = 1;
    if (hipMemcpyAsync(&related, d_flag.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return false;
    return related == 0;
  };
  const bool partition = max_win > 0 && use_partition();
  if (!partition) KPOP_HIP(hipMemsetAsync(S.d_ka.p, 0, n_bins * 4, st));
  bool fused_csr = false;  // the partition path wrote the spectrum itself
  uint64_t fused_bound = 0;
  DevBuf d_state, d_over;
  bool over_pending = false;  // the partition's rooms were guessed and nobody has looked at its overflow flag yet
  if (max_win > 0) {
    if (lds_mode && hb <= kHistLdsBits) {
      // the whole table fits a block's LDS: private copies, one global atomic per non-zero counter and block
      const uint32_t max_seg = div_up(max_win, kKeySeg);
      const uint32_t blocks = (uint32_t)std::min<uint64_t>((uint64_t)n_reads * max_seg, (uint64_t)ctx().n_cus * 2);
      const size_t lds = (size_t)n_bins * 4;
      if (protein) window_hist_lds_kernel<5><<<dim3(blocks), dim3(1024), lds, st>>>(S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), k, content, hb, table, n_reads, max_seg);
      else window_hist_lds_kernel<2><<<dim3(blocks), dim3(1024), lds, st>>>(S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), k, content, hb, table, n_reads, max_seg);
    } else if (partition) {
      // nothing repeats (a read set, unrelated genomes): partition the hashes by their top bits, count every bucket in LDS
      const int LB = part_bucket_bits(hb);
      const uint32_t n_buckets = 1u << (hb - LB);
      static_assert(sizeof(HistItemD) == 16 && sizeof(HistItem) == 8, "two items of (read, segment) in the room of one with its address");
      std::vector<HistItem> items;  // (DNA: pairs of entries hold one HistItemD)
      if (protein) {
        items.reserve(total_win / kPartItem + n_reads);
        for (uint32_t r = 0; r < n_reads; ++r) {
          const uint64_t len = rel[r + 1] - rel[r], w = len >= (uint64_t)k ? len - k + 1 : 0;
          for (uint64_t sgm = 0; sgm * kPartItem < w; ++sgm) items.push_back(HistItem{r, (uint32_t)sgm});
        }
      } else {
        items.reserve(2 * (total_win / kPartItemD + n_reads));
        for (uint32_t r = 0; r < n_reads; ++r) {
          const uint64_t len = rel[r + 1] - rel[r], w = len >= (uint64_t)k ? len - k + 1 : 0;
          for (uint64_t w0 = 0; w0 < w; w0 += kPartItemD) {
            const HistItemD it{rel[r] + w0, (uint32_t)std::min<uint64_t>(kPartItemD, w - w0), 0u};
            HistItem two[2];
            memcpy(two, &it, sizeof it);
            items.push_back(two[0]);
            items.push_back(two[1]);
          }
        }
      }
      const uint64_t n_items = protein ? items.size() : items.size() / 2;
      const uint32_t quota = n_buckets <= 512 ? kPartQuota / 2 : kPartQuota;
      const uint32_t blocks2 = (uint32_t)std::min<uint64_t>(div_up(n_items, 64), (uint64_t)ctx().n_cus * (quota < kPartQuota ? 2 : 1));
      // The exact sizes of the buckets cost a pass that hashes every window (a quarter of the call).  DNA, enough items: the sizes of
      // one item in kPartSample instead, every bucket given 5/4 of what that predicts + a margin (the sample is a few hundred to a few
      // thousand windows a bucket: 4 % standard deviation at worst), a bucket that overflows all the same caught by the partition
      // pass -- which then runs again behind the exact count (kpop_tune("histguess", 0): always the exact count).
      constexpr uint32_t kPartSample = 32, kPartMargin = 2048;
      bool guess = !protein && ctx().tune_histguess && !exact_sizes && n_items >= 64 * kPartSample;
      // (the spectrum straight from the bucket count ends in a copy of its length anyway: the flag is looked at there, not here)
      const bool fused_next = lds_mode != 4 && (lds_mode == 3 || total_win <= 2ull * (1ull << hb));
      std::vector<HistItem> sample;
      uint64_t sample_win = 0;
      if (guess) {
        sample.reserve(items.size() / kPartSample + 2);
        for (uint64_t i = 0; i < n_items; i += kPartSample) {
          sample.push_back(items[2 * i]);
          sample.push_back(items[2 * i + 1]);
          HistItemD it;
          memcpy(&it, &items[2 * i], sizeof it);
          sample_win += it.nw;
        }
      }
      const uint64_t n_sample = sample.size() / 2;
      const uint64_t room_total = guess ? (sample_win * kPartSample * 5) / 4 + (uint64_t)n_buckets * kPartMargin : total_win;
      const uint64_t trash_at = room_total + 8, trash_len = guess ? (uint64_t)blocks2 * 16 * quota : 0;
      DevBuf d_items, d_sample, d_size, d_poff, d_cursor, d_entries;
      KPOP_TRY(d_items.alloc(items.size() * sizeof(HistItem) + 16));
      KPOP_TRY(d_sample.alloc(sample.size() * sizeof(HistItem) + 16));
      KPOP_TRY(d_size.alloc((uint64_t)n_buckets * 4));
      KPOP_TRY(d_poff.alloc((uint64_t)(n_buckets + 1) * 8));
      KPOP_TRY(d_cursor.alloc((uint64_t)n_buckets * 8));
      KPOP_TRY(d_over.alloc(64));
      KPOP_TRY(d_entries.alloc(std::max(room_total + 8 + trash_len, total_win) * 2 + 16));
      KPOP_HIP(hipMemcpyAsync(d_items.p, items.data(), items.size() * sizeof(HistItem), hipMemcpyHostToDevice, st));
      if (guess) KPOP_HIP(hipMemcpyAsync(d_sample.p, sample.data(), sample.size() * sizeof(HistItem), hipMemcpyHostToDevice, st));
      KPOP_HIP(hipMemsetAsync(d_size.p, 0, (uint64_t)n_buckets * 4, st));
      KPOP_HIP(hipMemsetAsync(d_over.p, 0, 64, st));
      const uint32_t blocks1 = (uint32_t)std::min<uint64_t>(div_up(n_items, 16), (uint64_t)ctx().n_cus * 2);
      const uint32_t blocks1s = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(div_up(n_sample, 16), 1), (uint64_t)ctx().n_cus * 2);
      const size_t lds_part = (size_t)16 * quota * 4 + (size_t)(n_buckets + (n_buckets & 1u)) * 4 + (size_t)n_buckets * 8;
      static PerSlotOnce once;
      if (!once()) {
        KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hist_partition_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16 * kPartQuota * 4 + 2048 * 12)));
        KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hist_partition_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16 * kPartQuota * 4 + 2048 * 12)));
        KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hist_bucket_count_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hist_bucket_count_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        once() = true;
      }
#define KPOP_PART(SB)                                                                                                                        \
  do {                                                                                                                                       \
    hist_part_sizes_kernel<SB><<<dim3(blocks1), dim3(1024), (size_t)n_buckets * 4, st>>>(S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), k, content, \
                                                                                        d_items.as<HistItem>(), n_items, LB, n_buckets, d_size.as<uint32_t>()); \
    hist_part_offsets_kernel<<<dim3(1), dim3(1024), 0, st>>>(d_size.as<uint32_t>(), n_buckets, d_poff.as<uint64_t>(), d_cursor.as<unsigned long long>()); \
    hist_partition_kernel<SB><<<dim3(blocks2), dim3(1024), lds_part, st>>>(S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), k, content, d_items.as<HistItem>(), \
                                                                          n_items, LB, n_buckets, d_cursor.as<unsigned long long>(), d_entries.as<uint16_t>(), quota); \
  } while (0)
      if (guess) {
        hist_part_sizes_kernel<2><<<dim3(blocks1s), dim3(1024), (size_t)n_buckets * 4, st>>>(S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), k, content,
                                                                                            d_sample.as<HistItem>(), n_sample, LB, n_buckets, d_size.as<uint32_t>());
        hist_part_offsets_kernel<<<dim3(1), dim3(1024), 0, st>>>(d_size.as<uint32_t>(), n_buckets, d_poff.as<uint64_t>(), d_cursor.as<unsigned long long>(), kPartSample,
                                                                 kPartMargin);
        hist_partition_kernel<2><<<dim3(blocks2), dim3(1024), lds_part, st>>>(S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), k, content, d_items.as<HistItem>(), n_items, LB,
                                                                             n_buckets, d_cursor.as<unsigned long long>(), d_entries.as<uint16_t>(), quota,
                                                                             d_poff.as<uint64_t>() + 1, d_over.as<uint32_t>(), trash_at);
        KPOP_LAUNCH_CHECK();
        if (fused_next) over_pending = true;
        else {
          uint32_t over = 0;
          KPOP_HIP(hipMemcpyAsync(&over, d_over.p, 4, hipMemcpyDeviceToHost, st));
          KPOP_HIP(hipStreamSynchronize(st));
          if (over) {  // (a sample that did not speak for the batch: the exact count after all)
            guess = false;
            KPOP_HIP(hipMemsetAsync(d_size.p, 0, (uint64_t)n_buckets * 4, st));
          }
        }
      }
      if (!guess) {
        if (protein) KPOP_PART(5); else KPOP_PART(2);
      }
#undef KPOP_PART
      const unsigned long long *ends = guess ? d_cursor.as<unsigned long long>() : nullptr;
      // the spectrum straight out of the buckets' LDS tables (kpop_tune("histlds", 4): always partitioned, with the dense table + compaction of round 4, for A/B)
      // the pairs straight from the bucket count where the table is sparsely hit (a read set: 13.9 M windows into 16.8 M bins,
      // 0.31 -> 0.27 ms); where every bin is hit many times over (5,000 genomes: 148 M windows) the blocks all reach the look-back
      // at once and the dense table + scans are ahead (bucket count 0.38 against 0.26 + 0.08 ms): kept there.
      // (Tried and dropped, round 5: the round's keys sorted by bucket inside LDS and every bucket's run written by one wavefront
      // with consecutive lanes -- two more barriers and two more passes over LDS a round: the partition 139 -> 148 us on reads,
      // 890 -> 950 on genomes.  The scattered two-byte stores are not what the kernel waits for; its hashing is.)
      fused_csr = fused_next;
      if (fused_csr) {
        const uint64_t bound = std::min<uint64_t>(std::min<uint64_t>(total_win, n_bins), cap);
        KPOP_TRY(S.d_oh.alloc(std::max<uint64_t>(bound, 1) * 8));
        KPOP_TRY(S.d_oc.alloc(std::max<uint64_t>(bound, 1) * 4));
        KPOP_TRY(d_state.alloc((uint64_t)n_buckets * 8 + 64));
        KPOP_HIP(hipMemsetAsync(d_state.p, 0, (uint64_t)n_buckets * 8 + 64, st));
        uint64_t *state = d_state.as<uint64_t>() + 8;  // ([0] the ticket, [1] the total, then a word a bucket)
        hist_bucket_count_kernel<true><<<dim3(n_buckets), dim3(1024), (size_t)4 << LB, st>>>(
            d_entries.as<uint16_t>(), d_poff.as<uint64_t>(), LB, nullptr, n_buckets, reinterpret_cast<uint32_t *>(d_state.p), state, S.d_oh.as<uint64_t>(),
            S.d_oc.as<uint32_t>(), bound, d_state.as<uint64_t>() + 1, ends);
        fused_bound = bound;
      } else
        hist_bucket_count_kernel<false><<<dim3(n_buckets), dim3(1024), (size_t)4 << LB, st>>>(d_entries.as<uint16_t>(), d_poff.as<uint64_t>(), LB, table, n_buckets,
                                                                                             nullptr, nullptr, nullptr, nullptr, 0, nullptr, ends);
    } else if (max_win <= 4096) {
      read_hist_kernel<uint32_t><<<dim3(std::min<uint32_t>(div_up(n_reads, 4), 1u << 16)), dim3(256), 0, st>>>(
          S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), k, content, table, n_reads);
    } else if (lds_mode && n_reads >= 16) {
      // assemblies: the same stretch of 64 sequences combined in an LDS table before it reaches the global one
      const uint32_t max_seg = div_up(max_win, kCombSeg);
      const uint64_t n_chunks = (uint64_t)div_up(n_reads, kCombReads) * max_seg;
      const uint32_t blocks = (uint32_t)std::min<uint64_t>(n_chunks, (uint64_t)ctx().n_cus * 4);
      if (protein) window_hist_combine_kernel<5><<<dim3(blocks), dim3(1024), 0, st>>>(S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), k, content, table, n_reads, max_seg, lds_mode == 2);
      else window_hist_combine_kernel<2><<<dim3(blocks), dim3(1024), 0, st>>>(S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), k, content, table, n_reads, max_seg, lds_mode == 2);
    } else {
      const uint32_t max_seg = div_up(max_win, kKeySeg);
      window_hist_kernel<uint32_t><<<dim3(capped_grid((uint64_t)n_reads * max_seg)), dim3(256), 0, st>>>(
          S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), k, content, table, n_reads, max_seg);
    }
    KPOP_LAUNCH_CHECK();
  }
  if (fused_csr) {
    uint64_t nu = 0;
    uint32_t over = 0;
    KPOP_HIP(hipMemcpyAsync(&nu, d_state.as<uint64_t>() + 1, 8, hipMemcpyDeviceToHost, st));
    if (over_pending) KPOP_HIP(hipMemcpyAsync(&over, d_over.p, 4, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipStreamSynchronize(st));
    if (over) return hist_count_device(bases, offsets, n_reads, k, content, cap, S, st, true);  // (a sample that did not speak for the batch: once more, counted)
    if (nu > cap || nu > fused_bound)
      KPOP_FAIL(KPOP_ERR_CAPACITY, "kpop_count_reads: %llu distinct k-mers, capacity %llu", (unsigned long long)nu, (unsigned long long)cap);
    S.nu = nu;
    const uint64_t two[2] = {0, nu};
    KPOP_HIP(hipMemcpyAsync(S.d_oo.p, two, 16, hipMemcpyHostToDevice, st));
    KPOP_HIP(hipStreamSynchronize(st));
    return 0;
  }
  // count the non-zero bins, then write them out in index order
  uint64_t *sums = S.d_sums.as<uint64_t>();
  const uint64_t nb = scan_blocks(n_bins);
  scan_tile_sums_kernel<NonZero><<<dim3((uint32_t)nb), dim3(kScanThreads), 0, st>>>(NonZero{table}, n_bins, sums);
  KPOP_LAUNCH_CHECK();
  scan_block_sums_kernel<0><<<dim3(1), dim3(1024), 0, st>>>(sums, nb);
  KPOP_LAUNCH_CHECK();
  uint64_t nu = 0;
  KPOP_HIP(hipMemcpyAsync(&nu, sums + nb, 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  if (nu > cap)
    KPOP_FAIL(KPOP_ERR_CAPACITY, "kpop_count_reads: %llu distinct k-mers, capacity %llu", (unsigned long long)nu, (unsigned long long)cap);
  S.nu = nu;
  KPOP_TRY(S.d_oh.alloc(nu * 8));
  KPOP_TRY(S.d_oc.alloc(nu * 4));
  scan_apply_kernel<NonZero, StoreBins><<<dim3((uint32_t)nb), dim3(kScanThreads), 0, st>>>(
      NonZero{table}, StoreBins{table, S.d_oh.as<uint64_t>(), S.d_oc.as<uint32_t>()}, n_bins, sums);
  KPOP_LAUNCH_CHECK();
  const uint64_t two[2] = {0, nu};
  KPOP_HIP(hipMemcpyAsync(S.d_oo.p, two, 16, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipStreamSynchronize(st));
  return 0;
}

// ---------------------------------------------------------------------------
// -L on assemblies of up to 32,768 windows (a SARS-CoV-2 genome is 29,892 12-mers): ONE 1024-thread block per sequence,
// everything in LDS.  The windows' hashes go into a 128 KB key array, are sorted there (bitonic network; invalid windows
// carry the all-ones sentinel and end up last), runs of equal keys become (hash, count) pairs, and the block's place in
// the CSR comes from the same ticket + look-back hand-off count_wave_kernel uses (lookback.h) -- a block works ~50 us,
// so the hand-off costs nothing here.  The sequence is read once and its spectrum written once: no key array in HBM, no
// radix passes (5 passes over 8 B per window before), no scans.  Longer sequences and hashes beyond 30 bits keep the
// device-wide sort below.
// ---------------------------------------------------------------------------
template <int SB>
__global__ __launch_bounds__(1024) void count_block_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets,
                                                           uint32_t n, int k, int content, uint32_t *__restrict__ ticket_counter,
                                                           uint64_t *__restrict__ state, uint64_t *__restrict__ out_hash,
                                                           uint32_t *__restrict__ out_count, uint64_t *__restrict__ out_offsets) {
  extern __shared__ uint32_t s_key[];  // NP keys
  __shared__ uint32_t s_ticket, s_wtot[16];
  __shared__ uint64_t s_prefix;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_ticket = atomicAdd(ticket_counter, 1u);
  __syncthreads();
  const uint32_t r = s_ticket;
  if (r >= n) return;
  const uint64_t off = offsets[r], len = offsets[r + 1] - off;
  const uint32_t n_win = len >= (uint64_t)k ? (uint32_t)(len - k + 1) : 0u;
  uint32_t NP = 64;
  while (NP < n_win) NP <<= 1;
  const uint8_t *seq = bases + off;
  const int shift = SB * (k - 1);
  constexpr uint32_t kSym = (1u << SB) - 1u, kValid = SB == 2 ? 4u : 20u;
  // a thread hashes NP / 1024 CONSECUTIVE windows, rolling the hash from one to the next (k - 1 + NP / 1024 base loads
  // instead of k per window), and drops key i of its run at s_key[i * 1024 + thread]: before the sort any place is as
  // good as any other, and this one has the lanes of a wave on consecutive banks
  {
    const uint32_t per_h = NP >= 1024 ? NP / 1024 : 1, w0 = threadIdx.x * per_h;
    const uint32_t mask = (uint32_t)bits_mask(SB * k);
    uint32_t fwd = 0, rc = 0;
    int run = 0;
    if (w0 < n_win)
      for (int j = 0; j < k - 1; ++j) {
        const uint64_t at = (uint64_t)w0 + j;
        const uint32_t c = at < len ? (SB == 2 ? base_code(seq[at]) : protein_code(seq[at])) : kValid;
        fwd = ((fwd << SB) | (c & kSym)) & mask;
        if (SB == 2) rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
        run = c < kValid ? run + 1 : 0;
      }
    for (uint32_t i = 0; i < per_h; ++i) {
      const uint32_t w = w0 + i;
      uint32_t key = 0xFFFFFFFFu;
      if (w < n_win) {
        const uint32_t c = SB == 2 ? base_code(seq[w + k - 1]) : protein_code(seq[w + k - 1]);
        fwd = ((fwd << SB) | (c & kSym)) & mask;
        if (SB == 2) rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
        run = c < kValid ? run + 1 : 0;
        if (run >= k) key = (SB == 2 && content == KPOP_DNA_DS && rc < fwd) ? rc : fwd;
      }
      const uint32_t slot = NP >= 1024 ? i * 1024u + threadIdx.x : w;
      if (slot < NP && (NP >= 1024 || w < NP)) s_key[KEY(slot)] = key;
    }
  }
  // Bitonic network over LDS, REGISTER-BLOCKED: a thread takes 2^g keys (g <= 5) whose indices differ in g consecutive
  // bits, runs the g compare-exchange levels those bits stand for in registers, and puts the keys back -- one LDS round
  // trip and one barrier per group of levels: 30 passes over the 128 KB for 32,768 keys where a level per pass takes 120
  // (the sort is bound by LDS bandwidth: 2,048 cycles per pass).  Indices are padded by one word per 32 (KEY(i)), so the
  // lowest group, where a lane's keys are 32 consecutive ones, does not put every lane of a wave on one bank.
  {
    int m = 0;
    while ((1u << m) < NP) ++m;
    const int G = min(5, max(1, m - 10));
    for (int top = 0; top < m; ++top) {            // merge phase sz = 2^(top + 1): strides 2^top .. 1
      const uint32_t sz = 2u << top;
      for (int L = top + 1; L > 0;) {
        const int g = min(G, L), h = L - 1, lo = h - g + 1;  // this pass: stride bits h .. lo
        __syncthreads();
        switch (g) {  // the group size is a compile-time constant inside: straight-line loads, no predication
          case 5: bitonic_group_pass<5>(s_key, NP, sz, h, lo); break;
          case 4: bitonic_group_pass<4>(s_key, NP, sz, h, lo); break;
          case 3: bitonic_group_pass<3>(s_key, NP, sz, h, lo); break;
          case 2: bitonic_group_pass<2>(s_key, NP, sz, h, lo); break;
          default: bitonic_group_pass<1>(s_key, NP, sz, h, lo); break;
        }
        L -= g;
      }
    }
  }
  __syncthreads();
  // heads of runs: thread-order prefix over the block, NP / 1024 consecutive positions per thread
  const uint32_t per = NP / 1024 ? NP / 1024 : 1;
  const uint32_t p0 = threadIdx.x * per;
  uint32_t mine = 0;
  for (uint32_t i = p0; i < p0 + per && i < NP; ++i) {
    const uint32_t key = s_key[KEY(i)];
    mine += (key != 0xFFFFFFFFu && (i == 0 || s_key[KEY(i - 1)] != key)) ? 1u : 0u;
  }
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s_wtot[wv] = incl;
  __syncthreads();
  uint32_t before = incl - mine, total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wv) before += s_wtot[w];
    total += s_wtot[w];
  }
  if (wv == 0) {
    const uint64_t pre = lookback_exclusive(state, r, (uint64_t)total, lane, 4);
    if (lane == 0) s_prefix = pre;
  }
  __syncthreads();
  uint64_t o = s_prefix + before;
  {
    // runs that start in this thread's stretch: a run ends where the next one starts, so only the last one needs a search
    // (for the first position beyond the stretch whose key differs: the array is sorted).  A thread's pairs are consecutive
    // in the output and go straight out: that costs ~5x the spectrum's bytes in partial-line HBM writes (PMC), and is
    // still a third faster than staging them through an LDS tile for full-line writes (2.05 vs 2.75 ms on 2,000 genomes:
    // the tiles serialise the emission over the waves).
    const uint32_t end = min(p0 + per, NP);
    uint32_t start = 0xFFFFFFFFu, cur = 0;
    uint32_t prev = (p0 > 0 && p0 < NP) ? s_key[KEY(p0 - 1)] : 0xFFFFFFFFu;
    for (uint32_t i = p0; i < end; ++i) {
      const uint32_t key = s_key[KEY(i)];
      if (key != 0xFFFFFFFFu && (i == 0 || prev != key)) {
        if (start != 0xFFFFFFFFu) {
          out_hash[o] = cur;
          out_count[o] = i - start;
          ++o;
        }
        start = i;
        cur = key;
      } else if (key == 0xFFFFFFFFu && start != 0xFFFFFFFFu) {  // the sentinels begin: the open run ends here
        out_hash[o] = cur;
        out_count[o] = i - start;
        ++o;
        start = 0xFFFFFFFFu;
      }
      prev = key;
    }
    if (start != 0xFFFFFFFFu) {
      uint32_t lo = end, hi = NP;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_key[KEY(mid)] == cur) lo = mid + 1; else hi = mid;
      }
      out_hash[o] = cur;
      out_count[o] = lo - start;
    }
  }
  if (threadIdx.x == 0) {
    out_offsets[r] = s_prefix;
    if (r == n - 1) out_offsets[n] = s_prefix + total;
  }
}

// host arrays in, CSR on the device (S.d_oh / d_oc / d_oo); false through *done when the batch is not for this kernel
static int block_count_device(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content, uint64_t cap,
                              SortedSpectra &S, hipStream_t st, bool *done) {
  *done = false;
  if (hash_bits(k, content) > 30 || !ctx().tune_blocksort) return 0;
  const uint64_t base0 = offsets[0], n_bases = offsets[n_reads] - base0;
  std::vector<uint64_t> rel(n_reads + 1);
  uint64_t max_win = 0, worst = 0;
  for (uint32_t r = 0; r <= n_reads; ++r) rel[r] = offsets[r] - base0;
  for (uint32_t r = 0; r < n_reads; ++r) {
    const uint64_t len = rel[r + 1] - rel[r], w = len >= (uint64_t)k ? len - k + 1 : 0;
    max_win = std::max(max_win, w);
    worst += w;
  }
  if (max_win > kBlockSortMax) return 0;
  uint32_t NP = 64;
  while (NP < max_win) NP <<= 1;
  S.nu = 0;
  S.n_spectra = n_reads;
  KPOP_TRY(S.d_bases.alloc(n_bases));
  KPOP_TRY(S.d_off.alloc((uint64_t)(n_reads + 1) * 8));
  KPOP_TRY(S.d_scr.alloc(64 + ((uint64_t)n_reads + 1) * 8));
  KPOP_TRY(S.d_oo.alloc((uint64_t)(n_reads + 1) * 8));
  KPOP_TRY(S.d_oh.alloc(std::max<uint64_t>(worst, 1) * 8));
  KPOP_TRY(S.d_oc.alloc(std::max<uint64_t>(worst, 1) * 4));
  KPOP_HIP(hipMemcpyAsync(S.d_bases.p, bases + base0, n_bases, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(S.d_off.p, rel.data(), (uint64_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemsetAsync(S.d_scr.p, 0, 64 + ((uint64_t)n_reads + 1) * 8, st));
  uint32_t *ticket = S.d_scr.as<uint32_t>();
  uint64_t *state = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(S.d_scr.p) + 64);
  const size_t smem = ((size_t)NP + (NP >> 5) + 1) * 4;
  static PerSlotOnce attr_once[2];
  const int which = content == KPOP_PROTEIN ? 1 : 0;
  if (!attr_once[which]()) {
    if (which) KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&count_block_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((kBlockSortMax + kBlockSortMax / 32 + 1) * 4)));
    else KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&count_block_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((kBlockSortMax + kBlockSortMax / 32 + 1) * 4)));
    attr_once[which]() = true;
  }
  if (which)
    count_block_kernel<5><<<dim3(n_reads), dim3(1024), smem, st>>>(S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), n_reads, k, content, ticket,
                                                                   state, S.d_oh.as<uint64_t>(), S.d_oc.as<uint32_t>(), S.d_oo.as<uint64_t>());
  else
    count_block_kernel<2><<<dim3(n_reads), dim3(1024), smem, st>>>(S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), n_reads, k, content, ticket,
                                                                   state, S.d_oh.as<uint64_t>(), S.d_oc.as<uint32_t>(), S.d_oo.as<uint64_t>());
  KPOP_LAUNCH_CHECK();
  uint64_t total = 0;
  KPOP_HIP(hipMemcpyAsync(&total, S.d_oo.as<uint64_t>() + n_reads, 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  if (total > cap)
    KPOP_FAIL(KPOP_ERR_CAPACITY, "kpop_count_reads: %llu distinct (spectrum,k-mer) pairs, capacity %llu", (unsigned long long)total, (unsigned long long)cap);
  S.nu = total;
  *done = true;
  return 0;
}

// One batch of reads through the sort path, host arrays in; the CSR stays on the device in S (d_oh, d_oc, and for
// per_read d_oo with n_reads + 1 offsets relative to this batch).  per_read = 0 merges everything.  The buffers come
// from the caller's ArenaScope.
int sorted_count_device(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content, int per_read,
                        uint64_t cap, SortedSpectra &S, hipStream_t st) {
  if (per_read && n_reads > 0) {  // sequences that fit a block's LDS: sorted there, one block each
    bool done = false;
    KPOP_TRY(block_count_device(bases, offsets, n_reads, k, content, cap, S, st, &done));
    if (done) return 0;
  }
  // the merged spectrum of small-enough hashes: one atomic add per window (kpop_tune("hist", 0) keeps the sort)
  if (!per_read && ctx().tune_hist && hash_bits(k, content) <= kHistMaxBits && n_reads > 0)
    return hist_count_device(bases, offsets, n_reads, k, content, cap, S, st);
  const uint64_t base0 = offsets[0], n_bases = offsets[n_reads] - base0;
  std::vector<uint64_t> rel(n_reads + 1), woff(n_reads + 1);
  uint64_t tw = 0, max_win = 0;
  for (uint32_t r = 0; r < n_reads; ++r) {
    rel[r] = offsets[r] - base0;
    woff[r] = tw;
    const uint64_t len = offsets[r + 1] - offsets[r];
    const uint64_t w = len >= (uint64_t)k ? len - k + 1 : 0;
    tw += w;
    max_win = std::max(max_win, w);
  }
  rel[n_reads] = n_bases;
  woff[n_reads] = tw;
  const uint32_t n_spectra = per_read ? n_reads : 1;
  S.nu = 0;
  S.n_spectra = n_spectra;
  KPOP_TRY(S.d_oo.alloc((uint64_t)(n_spectra + 1) * 8));
  KPOP_HIP(hipMemsetAsync(S.d_oo.p, 0, (uint64_t)(n_spectra + 1) * 8, st));
  if (tw == 0) return 0;
  int id_bits = 0;
  while (per_read && (1ull << id_bits) < (uint64_t)n_reads) ++id_bits;
  // one more bit than the valid keys use: the all-ones sentinel of invalid windows must sort after them
  const int hb = hash_bits(k, content);
  const int bits = hb + id_bits + 1;
  if (bits > 64) KPOP_FAIL(KPOP_ERR_INVALID, "sorted_count_batch: %d key bits (caller must split the batch)", bits);
  const uint32_t max_seg = div_up(max_win, kKeySeg);
  DevBuf &d_bases = S.d_bases, &d_off = S.d_off, &d_woff = S.d_woff, &d_ka = S.d_ka, &d_kb = S.d_kb, &d_scr = S.d_scr,
         &d_start = S.d_start, &d_sums = S.d_sums;
  KPOP_TRY(d_bases.alloc(n_bases));
  KPOP_TRY(d_off.alloc((uint64_t)(n_reads + 1) * 8));
  KPOP_TRY(d_woff.alloc((uint64_t)(n_reads + 1) * 8));
  KPOP_TRY(d_ka.alloc(tw * 8));
  KPOP_TRY(d_kb.alloc(tw * 8));
  KPOP_TRY(d_scr.alloc(radix_scratch_bytes(tw)));
  KPOP_TRY(d_start.alloc(tw * 8));
  KPOP_TRY(d_sums.alloc((scan_blocks(tw) + 1) * 8 * 2));
  KPOP_HIP(hipMemcpyAsync(d_bases.p, bases + base0, n_bases, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(d_off.p, rel.data(), (uint64_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(d_woff.p, woff.data(), (uint64_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
  dim3 grid(capped_grid((uint64_t)n_reads * max_seg));
  if (hb <= 30)
    window_keys_kernel<uint32_t><<<grid, dim3(256), 0, st>>>(d_bases.as<uint8_t>(), d_off.as<uint64_t>(),
                                                             d_woff.as<uint64_t>(), k, content, per_read, 0u,
                                                             d_ka.as<uint64_t>(), n_reads, max_seg);
  else
    window_keys_kernel<uint64_t><<<grid, dim3(256), 0, st>>>(d_bases.as<uint8_t>(), d_off.as<uint64_t>(),
                                                             d_woff.as<uint64_t>(), k, content, per_read, 0u,
                                                             d_ka.as<uint64_t>(), n_reads, max_seg);
  KPOP_LAUNCH_CHECK();
  uint64_t *sorted = nullptr;
  KPOP_TRY(radix_sort_u64(d_ka.as<uint64_t>(), d_kb.as<uint64_t>(), tw, bits, d_scr.p, st, &sorted));
  uint64_t *other = (sorted == d_ka.as<uint64_t>()) ? d_kb.as<uint64_t>() : d_ka.as<uint64_t>();
  // distinct keys -> `other`, first positions -> d_start; totals at the end of the sums arrays
  uint64_t *sums1 = d_sums.as<uint64_t>(), *sums2 = sums1 + scan_blocks(tw) + 1;
  KPOP_TRY(exclusive_scan(HeadFlag{sorted}, StoreHeads{sorted, other, d_start.as<uint64_t>()}, tw, sums1, st));
  {  // only the number of valid keys is wanted: tile sums and their scan, no apply pass
    const uint64_t nb = scan_blocks(tw);
    scan_tile_sums_kernel<ValidFlag><<<dim3((uint32_t)nb), dim3(kScanThreads), 0, st>>>(ValidFlag{sorted}, tw, sums2);
    KPOP_LAUNCH_CHECK();
    scan_block_sums_kernel<0><<<dim3(1), dim3(1024), 0, st>>>(sums2, nb);
    KPOP_LAUNCH_CHECK();
  }
  const uint64_t *d_nu = sums1 + scan_blocks(tw), *d_nv = sums2 + scan_blocks(tw);
  uint64_t nu = 0;
  KPOP_HIP(hipMemcpyAsync(&nu, d_nu, 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));  // the host rests its copy of `rel`/`woff` on this too
  if (nu > cap)
    KPOP_FAIL(KPOP_ERR_CAPACITY, "kpop_count_reads: %llu distinct (spectrum,k-mer) pairs, capacity %llu",
              (unsigned long long)nu, (unsigned long long)cap);
  S.nu = nu;
  KPOP_TRY(S.d_oh.alloc(nu * 8));
  KPOP_TRY(S.d_oc.alloc(nu * 4));
  if (nu) {
    finish_spectra_kernel<<<dim3(std::min<uint32_t>(div_up(nu, 256), 8192)), dim3(256), 0, st>>>(
        other, d_start.as<uint64_t>(), d_nu, d_nv, bits_mask(hb), S.d_oh.as<uint64_t>(), S.d_oc.as<uint32_t>());
    KPOP_LAUNCH_CHECK();
  }
  if (per_read) {
    spectrum_bounds_kernel<<<dim3(div_up((uint64_t)n_spectra + 1, 256)), dim3(256), 0, st>>>(other, d_nu, n_spectra, 0u,
                                                                                              hb, S.d_oo.as<uint64_t>());
    KPOP_LAUNCH_CHECK();
  } else {
    const uint64_t two[2] = {0, nu};
    KPOP_HIP(hipMemcpyAsync(S.d_oo.p, two, 16, hipMemcpyHostToDevice, st));
    KPOP_HIP(hipStreamSynchronize(st));
  }
  return 0;
}

// the same with the CSR copied out to host arrays (capacity `cap` entries)
int sorted_count_batch(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content,
                       int per_read, uint64_t *out_hash, uint32_t *out_count, uint64_t *out_offsets, uint64_t cap,
                       uint64_t *n_written) {
  hipStream_t st = nullptr;
  SortedSpectra S;
  *n_written = 0;
  KPOP_TRY(sorted_count_device(bases, offsets, n_reads, k, content, per_read, cap, S, st));
  KPOP_HIP(hipMemcpyAsync(out_offsets, S.d_oo.p, (uint64_t)(S.n_spectra + 1) * 8, hipMemcpyDeviceToHost, st));
  if (S.nu) {
    KPOP_HIP(hipMemcpyAsync(out_hash, S.d_oh.p, S.nu * 8, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipMemcpyAsync(out_count, S.d_oc.p, S.nu * 4, hipMemcpyDeviceToHost, st));
  }
  KPOP_HIP(hipStreamSynchronize(st));
  *n_written = S.nu;
  return 0;
}

}  // namespace kpop

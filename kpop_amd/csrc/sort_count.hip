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
                             SortedSpectra &S, hipStream_t st) {
  const uint64_t base0 = offsets[0], n_bases = offsets[n_reads] - base0;
  std::vector<uint64_t> rel(n_reads + 1);
  uint64_t max_win = 0;
  for (uint32_t r = 0; r <= n_reads; ++r) rel[r] = offsets[r] - base0;
  for (uint32_t r = 0; r < n_reads; ++r) {
    const uint64_t len = rel[r + 1] - rel[r];
    max_win = std::max(max_win, len >= (uint64_t)k ? len - k + 1 : 0);
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
  KPOP_HIP(hipMemsetAsync(S.d_ka.p, 0, n_bins * 4, st));
  uint32_t *table = S.d_ka.as<uint32_t>();
  if (max_win > 0) {
    if (max_win <= 4096) {
      read_hist_kernel<uint32_t><<<dim3(std::min<uint32_t>(div_up(n_reads, 4), 1u << 16)), dim3(256), 0, st>>>(
          S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), k, content, table, n_reads);
    } else {
      const uint32_t max_seg = div_up(max_win, kKeySeg);
      window_hist_kernel<uint32_t><<<dim3(capped_grid((uint64_t)n_reads * max_seg)), dim3(256), 0, st>>>(
          S.d_bases.as<uint8_t>(), S.d_off.as<uint64_t>(), k, content, table, n_reads, max_seg);
    }
    KPOP_LAUNCH_CHECK();
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

// One batch of reads through the sort path, host arrays in; the CSR stays on the device in S (d_oh, d_oc, and for
// per_read d_oo with n_reads + 1 offsets relative to this batch).  per_read = 0 merges everything.  The buffers come
// from the caller's ArenaScope.
int sorted_count_device(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content, int per_read,
                        uint64_t cap, SortedSpectra &S, hipStream_t st) {
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

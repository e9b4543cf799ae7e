// radix_sort.h -- device-wide stable LSD radix sort of u64 keys, 8 bits per pass.
//
// Used where one wavefront's registers cannot hold a sequence's k-mers: genomes
// in -L mode and the merged (-l) spectrum.  Three steps per pass, each its own
// launch (no in-launch inter-workgroup hand-off, hence no agent-scope protocol):
//   radix_count_kernel   per-tile digit histogram  -> counts[digit][tile]
//   exclusive_scan       over the flattened [256][n_tiles] table (scan.h)
//   radix_scatter_kernel stable ranks inside the tile + scanned bases -> scatter
// A tile is 4 waves x 64 lanes x kRadixItems keys; wave w owns a contiguous
// quarter of the tile, element (w, j, lane) = base + (w*kRadixItems + j)*64 + lane,
// so loads are coalesced and the in-tile order is (wave, j, lane).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "common.h"
#include "scan.h"

namespace kpop {

constexpr int kRadixItems = 8;
constexpr uint32_t kRadixTile = 4 * 64 * kRadixItems;  // 2048 keys

static inline uint64_t radix_tiles(uint64_t n) { return (n + kRadixTile - 1) / kRadixTile; }

// scratch layout: u32 counts[256*tiles] | u64 bases[256*tiles] | u64 scan sums
struct RadixScratch {
  uint32_t *counts;
  uint64_t *bases;
  uint64_t *sums;
};
static inline uint64_t radix_scratch_bytes(uint64_t n) {
  const uint64_t t = radix_tiles(n) * 256;
  return ((t * 4 + 63) & ~63ull) + t * 8 + (scan_blocks(t) + 1) * 8 + 64;
}
static inline RadixScratch radix_carve(void *p, uint64_t n) {
  const uint64_t t = radix_tiles(n) * 256;
  char *c = reinterpret_cast<char *>(p);
  RadixScratch s;
  s.counts = reinterpret_cast<uint32_t *>(c);
  s.bases = reinterpret_cast<uint64_t *>(c + ((t * 4 + 63) & ~63ull));
  s.sums = s.bases + t;
  return s;
}

template <int kDummy = 0>
__global__ __launch_bounds__(256) void radix_count_kernel(const uint64_t *__restrict__ keys, uint64_t n, int shift,
                                                          uint64_t n_tiles, uint32_t *__restrict__ counts) {
  __shared__ uint32_t s_hist[256];
  s_hist[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * kRadixTile;
  for (uint32_t e = threadIdx.x; e < kRadixTile; e += 256) {
    const uint64_t i = base + e;
    if (i < n) atomicAdd(&s_hist[(uint32_t)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  counts[(uint64_t)threadIdx.x * n_tiles + blockIdx.x] = s_hist[threadIdx.x];
}

// kVal = 1: a u32 value travels with every key (vin -> vout)
template <int kVal = 0>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                            uint64_t n, int shift, uint64_t n_tiles,
                                                            const uint64_t *__restrict__ bases,
                                                            const uint32_t *__restrict__ vin = nullptr,
                                                            uint32_t *__restrict__ vout = nullptr) {
  __shared__ uint32_t s_cnt[4][256];
  __shared__ uint64_t s_base[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < 4; ++w) s_cnt[w][threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * kRadixTile + (uint64_t)wv * kRadixItems * 64;
  uint64_t key[kRadixItems];
  uint32_t rank[kRadixItems];
  uint32_t val[kVal ? kRadixItems : 1];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int j = 0; j < kRadixItems; ++j) {
    const uint64_t i = base + (uint64_t)j * 64 + lane;
    const bool ok = i < n;
    key[j] = ok ? in[i] : ~0ull;
    if (kVal) val[j] = ok ? vin[i] : 0u;
    const uint32_t dg = (uint32_t)(key[j] >> shift) & 255u;
    // lanes of this wave holding the same digit (out-of-range lanes form a class of their own)
    uint64_t peers = __ballot(ok);
    if (!ok) peers = ~peers;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (dg >> b) & 1u;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const int leader = __ffsll((unsigned long long)peers) - 1;
    uint32_t old = 0;
    if (ok && lane == leader) {
      old = s_cnt[wv][dg];
      s_cnt[wv][dg] = old + (uint32_t)__popcll(peers);
    }
    old = (uint32_t)__shfl((int)old, leader, 64);
    rank[j] = old + (uint32_t)__popcll(peers & lt);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {  // per digit: where each wave's run starts = scanned base + the earlier waves' counts
    const uint32_t dg = threadIdx.x;
    uint64_t b = bases[(uint64_t)dg * n_tiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      s_base[w][dg] = b;
      b += s_cnt[w][dg];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kRadixItems; ++j) {
    const uint64_t i = base + (uint64_t)j * 64 + lane;
    if (i < n) {
      const uint32_t dg = (uint32_t)(key[j] >> shift) & 255u;
      out[s_base[wv][dg] + rank[j]] = key[j];
      if (kVal) vout[s_base[wv][dg] + rank[j]] = val[j];
    }
  }
}

struct LoadCounts {
  const uint32_t *p;
  __device__ uint32_t operator()(uint64_t i) const { return p[i]; }
};
struct StoreBases {
  uint64_t *out;
  __device__ void operator()(uint64_t i, uint64_t prefix, uint32_t) const { out[i] = prefix; }
};

// Sorts the low `bits` bits' worth of passes; keys ping-pong between a and b.
// Returns (through *result) the buffer holding the sorted keys.
static inline int radix_sort_u64(uint64_t *a, uint64_t *b, uint64_t n, int bits, void *scratch, hipStream_t st,
                                 uint64_t **result) {
  *result = a;
  if (n == 0) return 0;
  const uint64_t n_tiles = radix_tiles(n);
  if (n_tiles > 0x7FFFFFFFull) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "radix_sort_u64: %llu keys", (unsigned long long)n);
  RadixScratch s = radix_carve(scratch, n);
  uint64_t *src = a, *dst = b;
  for (int shift = 0; shift < bits; shift += 8) {
    radix_count_kernel<0><<<dim3((uint32_t)n_tiles), dim3(256), 0, st>>>(src, n, shift, n_tiles, s.counts);
    KPOP_LAUNCH_CHECK();
    KPOP_TRY(exclusive_scan(LoadCounts{s.counts}, StoreBases{s.bases}, n_tiles * 256, s.sums, st));
    radix_scatter_kernel<0><<<dim3((uint32_t)n_tiles), dim3(256), 0, st>>>(src, dst, n, shift, n_tiles, s.bases);
    KPOP_LAUNCH_CHECK();
    uint64_t *t = src;
    src = dst;
    dst = t;
  }
  *result = src;
  return 0;
}

// the same with a u32 value per key (va / vb ping-pong with the keys); *vresult = the values beside *result
static inline int radix_sort_pairs_u64(uint64_t *a, uint64_t *b, uint32_t *va, uint32_t *vb, uint64_t n, int bits, void *scratch,
                                       hipStream_t st, uint64_t **result, uint32_t **vresult) {
  *result = a;
  *vresult = va;
  if (n == 0) return 0;
  const uint64_t n_tiles = radix_tiles(n);
  if (n_tiles > 0x7FFFFFFFull) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "radix_sort_pairs_u64: %llu keys", (unsigned long long)n);
  RadixScratch s = radix_carve(scratch, n);
  uint64_t *src = a, *dst = b;
  uint32_t *vsrc = va, *vdst = vb;
  for (int shift = 0; shift < bits; shift += 8) {
    radix_count_kernel<0><<<dim3((uint32_t)n_tiles), dim3(256), 0, st>>>(src, n, shift, n_tiles, s.counts);
    KPOP_LAUNCH_CHECK();
    KPOP_TRY(exclusive_scan(LoadCounts{s.counts}, StoreBases{s.bases}, n_tiles * 256, s.sums, st));
    radix_scatter_kernel<1><<<dim3((uint32_t)n_tiles), dim3(256), 0, st>>>(src, dst, n, shift, n_tiles, s.bases, vsrc, vdst);
    KPOP_LAUNCH_CHECK();
    std::swap(src, dst);
    std::swap(vsrc, vdst);
  }
  *result = src;
  *vresult = vsrc;
  return 0;
}

}  // namespace kpop

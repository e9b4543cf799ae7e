// scan.h -- device-wide exclusive prefix sum (three launches, no inter-workgroup
// hand-off inside a launch, so no agent-scope protocol is needed).
//   in(i)        : functor giving the u32 addend of element i
//   out(i, pre)  : functor receiving the exclusive prefix of element i
// `d_block_sums` needs scan_blocks(n)+1 u64 entries; the grand total is left in
// d_block_sums[scan_blocks(n)].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace kpop {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;  // per thread
constexpr uint64_t kScanTile = (uint64_t)kScanThreads * kScanItems;

static inline uint64_t scan_blocks(uint64_t n) { return (n + kScanTile - 1) / kScanTile; }

__device__ __forceinline__ uint64_t block_reduce_sum_u64(uint64_t v, uint64_t *s_tmp /* >= 4 */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint64_t)__shfl_down((unsigned long long)v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) s_tmp[wave] = v;
  __syncthreads();
  uint64_t t = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_tmp[w];
  __syncthreads();
  return t;
}

// A thread owns kScanItems CONSECUTIVE elements (that is what makes the prefix a per-thread running sum), but the
// functors usually read global memory at the element's index: evaluated thread by thread that is one 128-byte line per
// lane and instruction for 8 useful bytes (measured: 22x the key bytes in HBM traffic on the head-flag scans of the
// sort path).  So the tile's elements are evaluated in lane-consecutive order into LDS, and each thread then picks its
// run out of LDS (padded: a run of 16 words per thread would put every lane of a wave on the same banks).
__device__ __forceinline__ uint32_t scan_slot(uint32_t e) { return e + (e >> 4); }
constexpr uint32_t kScanLdsWords = (uint32_t)kScanTile + ((uint32_t)kScanTile >> 4);

template <class In>
__global__ __launch_bounds__(kScanThreads) void scan_tile_sums_kernel(In in, uint64_t n, uint64_t *block_sums) {
  __shared__ uint64_t s_tmp[4];
  const uint64_t tile0 = (uint64_t)blockIdx.x * kScanTile;
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {  // a sum needs no particular assignment of elements to threads
    const uint64_t e = tile0 + (uint64_t)i * kScanThreads + threadIdx.x;
    if (e < n) s += in(e);
  }
  uint64_t t = block_reduce_sum_u64(s, s_tmp);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = t;
}

// single block: exclusive scan of block_sums[0..nb) in place, total to block_sums[nb]
template <int kDummy = 0>
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(uint64_t *block_sums, uint64_t nb) {
  __shared__ uint64_t s_wave[16];
  __shared__ uint64_t s_carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (uint64_t base = 0; base < nb; base += 1024) {
    uint64_t i = base + threadIdx.x;
    uint64_t v = (i < nb) ? block_sums[i] : 0;
    uint64_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint64_t t = (uint64_t)__shfl_up((unsigned long long)incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
    uint64_t carry = s_carry;
    if (i < nb) block_sums[i] = carry + wave_off + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + wave_off + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) block_sums[nb] = s_carry;
}

template <class In, class Out>
__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(In in, Out out, uint64_t n,
                                                                   const uint64_t *block_sums) {
  __shared__ uint64_t s_wave[4];
  __shared__ uint32_t s_v[kScanLdsWords];
  __shared__ uint32_t s_pre[kScanLdsWords];  // prefix within the tile (a tile holds < 2^32 in total: 4096 u32 addends may not, see below)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t tile0 = (uint64_t)blockIdx.x * kScanTile;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    const uint32_t e = (uint32_t)i * kScanThreads + threadIdx.x;
    s_v[scan_slot(e)] = (tile0 + e < n) ? in(tile0 + e) : 0u;
  }
  __syncthreads();
  const uint32_t first = threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    v[i] = s_v[scan_slot(first + i)];
    s += v[i];
  }
  uint64_t incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint64_t t = (uint64_t)__shfl_up((unsigned long long)incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint64_t pre = incl - s;  // within the tile
  for (int w = 0; w < wave; ++w) pre += s_wave[w];
  // prefixes inside a tile are kept as 32-bit offsets from the tile's base; the addends of every user are flags or
  // digit counts of a tile, far below 2^32 / 4096 each
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    s_pre[scan_slot(first + i)] = (uint32_t)pre;
    pre += v[i];
  }
  __syncthreads();
  const uint64_t base = block_sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    const uint32_t e = (uint32_t)i * kScanThreads + threadIdx.x;
    if (tile0 + e < n) out(tile0 + e, base + s_pre[scan_slot(e)], s_v[scan_slot(e)]);
  }
}

// Enqueue the three launches.  d_block_sums: scan_blocks(n)+1 u64.
template <class In, class Out>
int exclusive_scan(In in, Out out, uint64_t n, uint64_t *d_block_sums, hipStream_t st) {
  const uint64_t nb = scan_blocks(n);
  if (nb == 0) {
    KPOP_HIP(hipMemsetAsync(d_block_sums, 0, sizeof(uint64_t), st));
    return 0;
  }
  scan_tile_sums_kernel<In><<<dim3((uint32_t)nb), dim3(kScanThreads), 0, st>>>(in, n, d_block_sums);
  KPOP_LAUNCH_CHECK();
  scan_block_sums_kernel<0><<<dim3(1), dim3(1024), 0, st>>>(d_block_sums, nb);
  KPOP_LAUNCH_CHECK();
  scan_apply_kernel<In, Out><<<dim3((uint32_t)nb), dim3(kScanThreads), 0, st>>>(in, out, n, d_block_sums);
  KPOP_LAUNCH_CHECK();
  return 0;
}

}  // namespace kpop

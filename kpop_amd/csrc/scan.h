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

template <class In>
__global__ __launch_bounds__(kScanThreads) void scan_tile_sums_kernel(In in, uint64_t n, uint64_t *block_sums) {
  __shared__ uint64_t s_tmp[4];
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) s += in(base + i);
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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    v[i] = (base + i < n) ? in(base + i) : 0u;
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
  uint64_t pre = block_sums[blockIdx.x] + incl - s;
  for (int w = 0; w < wave; ++w) pre += s_wave[w];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) out(base + i, pre, v[i]);
    pre += v[i];
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

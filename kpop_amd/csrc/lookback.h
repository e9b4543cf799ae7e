// lookback.h -- the exclusive prefix of a per-block total over the blocks that started earlier, inside one launch
// (count_wave_kernel and count_block_kernel place their spectra in the CSR with it; see count_twist.hip for the protocol).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kpop {

constexpr uint64_t kLookbackValueMask = (1ull << 62) - 1;

// exclusive prefix of `total` over the blocks with a smaller ticket; every lane of the calling wave gets it
__device__ __forceinline__ uint64_t lookback_exclusive(uint64_t *state, uint32_t ticket, uint64_t total, int lane, int naps) {
  if (ticket == 0) {
    if (lane == 0) __hip_atomic_store(&state[0], (2ull << 62) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
  }
  if (lane == 0) __hip_atomic_store(&state[ticket], (1ull << 62) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint64_t acc = 0;
  int64_t top = (int64_t)ticket - 1;  // nearest predecessor not yet added
  // kLookWords words per lane: a step covers 64 x kLookWords predecessors, so the chain of prefixes moves that many
  // blocks per round trip to memory (with one word per lane it moved slower than the blocks finished their reads)
  constexpr int kLookWords = 8;
  for (;;) {
    uint64_t w[kLookWords];
    for (;;) {
      bool ready = true;
#pragma unroll
      for (int j = 0; j < kLookWords; ++j) {
        const int64_t idx = top - j * 64 - lane;
        w[j] = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (3ull << 62);
        ready = ready && (w[j] >> 62) != 0;
      }
      if (ready) break;
      for (int z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(8);
    }
    bool done = false;
#pragma unroll
    for (int j = 0; j < kLookWords; ++j) {
      if (done) break;  // wave-uniform
      const int64_t idx = top - j * 64 - lane;
      const uint64_t full = __ballot((w[j] >> 62) == 2);
      const int stop = full ? __ffsll((long long)full) - 1 : 64;  // nearest lane holding an inclusive prefix
      uint64_t v = (lane <= stop && idx >= 0) ? (w[j] & kLookbackValueMask) : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, o, 64);
      acc += v;
      done = full != 0;
    }
    if (done || top - 64 * kLookWords < 0) break;
    top -= 64 * kLookWords;
  }
  if (lane == 0) __hip_atomic_store(&state[ticket], (2ull << 62) | (acc + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return acc;
}

// The same prefix with the predecessors' words in two levels -- a word per block and a word per GROUP of 64 consecutive
// tickets -- so that a block polls at most 64 + 64 words, not every block in flight (a thousand resident blocks each polling
// 512 words per round was 29 % of count_wave_kernel):
//   state[t]   {1, total of block t}                       stored by block t as soon as it has counted
//   garr[g]    arrivals of group g                          the last arriver adds up the group's 64 words and stores
//   gstate[g]  {1, total of group g} or {2, prefix through the END of group g}   (the latter by the group's last ticket, once it
//              knows its own prefix; both with an atomic max, so that an aggregate never overwrites a prefix)
// A block's prefix = the totals of the earlier tickets of its group + the group words before its group, back to the nearest
// {2, ...}.  Predecessors have smaller tickets, hence are running: no wait can deadlock.  Every word is its own message (one
// 8-byte store) and every reader polls until the words it needs are there, so the atomics are RELAXED: with acquire / release
// at agent scope every poll invalidated caches and the kernel took 0.49 ms instead of 0.21.
constexpr uint32_t kLookGroup = 64;

__device__ __forceinline__ uint64_t lookback_exclusive_grouped(uint64_t *state, uint64_t *gstate, uint32_t *garr, uint32_t ticket,
                                                               uint32_t n_tickets, uint64_t total, int lane, int naps) {
  const uint32_t g = ticket / kLookGroup, p = ticket % kLookGroup;
  const uint32_t members = min(kLookGroup, n_tickets - g * kLookGroup);
  uint32_t arrived = 0;
  if (lane == 0) {
    __hip_atomic_store(&state[ticket], (1ull << 62) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrived = __hip_atomic_fetch_add(&garr[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  arrived = (uint32_t)__shfl((int)arrived, 0, 64);
  if (arrived == members - 1) {  // the group is complete: its total
    uint64_t w;
    for (;;) {
      w = (uint32_t)lane < members ? __hip_atomic_load(&state[g * kLookGroup + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (1ull << 62);
      if (__ballot((w >> 62) == 0) == 0) break;  // (all there by the arrival count; a stale read polls again)
      __builtin_amdgcn_s_sleep(2);
    }
    uint64_t v = w & kLookbackValueMask;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, o, 64);
    if (lane == 0) __hip_atomic_fetch_max(&gstate[g], (1ull << 62) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // the earlier tickets of this group
  uint64_t acc = 0;
  if (p) {
    uint64_t w;
    for (;;) {
      w = (uint32_t)lane < p ? __hip_atomic_load(&state[g * kLookGroup + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (1ull << 62);
      if (__ballot((w >> 62) == 0) == 0) break;
      for (int z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(8);
    }
    uint64_t v = (uint32_t)lane < p ? (w & kLookbackValueMask) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, o, 64);
    acc = v;
  }
  // the groups before this one, nearest first, back to a group whose word is a prefix
  int64_t top = (int64_t)g - 1;
  while (top >= 0) {
    const int64_t idx = top - lane;
    uint64_t w;
    uint64_t full;
    for (;;) {
      w = idx >= 0 ? __hip_atomic_load(&gstate[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);  // (before the first group: prefix 0)
      full = __ballot((w >> 62) == 2);
      const uint64_t missing = __ballot((w >> 62) == 0);
      const int stop = full ? __ffsll((long long)full) - 1 : 64;
      const int first_missing = missing ? __ffsll((long long)missing) - 1 : 64;
      if (first_missing > stop || !missing) break;  // everything nearer than the nearest prefix is there
      for (int z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(8);
    }
    const int stop = full ? __ffsll((long long)full) - 1 : 64;
    uint64_t v = lane <= stop ? (w & kLookbackValueMask) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, o, 64);
    acc += v;
    if (full) break;
    top -= 64;
  }
  if (p == members - 1 && lane == 0)  // the prefix through the end of this group, for the groups after it
    __hip_atomic_fetch_max(&gstate[g], (2ull << 62) | (acc + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return acc;
}

}  // namespace kpop

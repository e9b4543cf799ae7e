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

}  // namespace kpop

// wave_sort.h -- one-wavefront (64 lanes) sort + run-length collapse, in registers.
//
// A wave holds N = 64*R keys, R per lane; element index e = lane*R + r.
// Bitonic network: strides below R are in-lane register swaps, strides >= R
// are one cross-lane exchange (lane ^ (stride/R)) per register.  Invalid
// slots carry the all-ones sentinel and sort to the end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kpop {

template <typename K>
__device__ __forceinline__ K key_min(K a, K b) { return a < b ? a : b; }
template <typename K>
__device__ __forceinline__ K key_max(K a, K b) { return a < b ? b : a; }

// lane ^ m for the strides of the networks below.  ds_bpermute (what __shfl_xor is) goes through the LDS pipe and comes back ~100
// cycles later, and every stage of a network waits for the one before: 21 stages of a 64-key sort were mostly that wait.  gfx950
// can do each of the six strides in the vector pipe: 1 and 2 are quad permutations, 8 a rotation of the row of 16 (DPP modifiers of
// v_mov), 4 a half-row mirror followed by a quad reversal, 16 and 32 the row / half-wave swaps (v_permlane16_swap, v_permlane32_swap:
// with both operands the same value, one result holds the lower partner's word in both halves and the other the upper's).
template <int CTRL>
__device__ __forceinline__ uint32_t wave_dpp_mov(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_shfl_xor(uint32_t v, int m) {  // m: a constant once the networks are unrolled
  switch (m) {
    case 1: return wave_dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    case 2: return wave_dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    case 4: return wave_dpp_mov<0x1B>(wave_dpp_mov<0x141>(v));  // row_half_mirror, then quad_perm [3,2,1,0]
    case 8: return wave_dpp_mov<0x128>(v);  // row_ror:8
    case 16: {
      const auto s = __builtin_amdgcn_permlane16_swap(v, v, false, false);
      return (__lane_id() & 16) ? s[0] : s[1];
    }
    case 32: {
      const auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
      return (__lane_id() & 32) ? s[0] : s[1];
    }
    default: return (uint32_t)__shfl_xor((int)v, m, 64);
  }
}
__device__ __forceinline__ uint64_t wave_shfl_xor(uint64_t v, int m) {
  return ((uint64_t)wave_shfl_xor((uint32_t)(v >> 32), m) << 32) | wave_shfl_xor((uint32_t)v, m);
}
__device__ __forceinline__ uint32_t wave_shfl_up1(uint32_t v) { return (uint32_t)__shfl_up((int)v, 1, 64); }
__device__ __forceinline__ uint64_t wave_shfl_up1(uint64_t v) {
  return (uint64_t)__shfl_up((unsigned long long)v, 1, 64);
}

template <int CTRL>  // row_shl:n (0x100 + n: lane i reads lane i + n of its row of 16) / row_shr:n (0x110 + n: lane i - n); lanes that would read outside the row get 0
__device__ __forceinline__ uint32_t wave_row_shift(uint32_t v) { return wave_dpp_mov<CTRL>(v); }
template <int CTRL>
__device__ __forceinline__ uint64_t wave_row_shift(uint64_t v) {
  return ((uint64_t)wave_dpp_mov<CTRL>((uint32_t)(v >> 32)) << 32) | wave_dpp_mov<CTRL>((uint32_t)v);
}
// lane ^ (G - 1): the lanes of every group of G in reverse order (the first stage of a merge below)
__device__ __forceinline__ uint32_t wave_mirror(uint32_t v, int G) {  // G: a constant once the network is unrolled
  switch (G) {
    case 2: return wave_dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    case 4: return wave_dpp_mov<0x1B>(v);    // quad_perm [3,2,1,0]
    case 8: return wave_dpp_mov<0x141>(v);   // row_half_mirror
    case 16: return wave_dpp_mov<0x140>(v);  // row_mirror
    case 32: return wave_shfl_xor(wave_dpp_mov<0x140>(v), 16);
    default: return (uint32_t)__shfl((int)v, 63 - (int)__lane_id(), 64);
  }
}
__device__ __forceinline__ uint64_t wave_mirror(uint64_t v, int G) {
  return ((uint64_t)wave_mirror((uint32_t)(v >> 32), G) << 32) | wave_mirror((uint32_t)v, G);
}

// The bitonic network in its one-direction form: a merge of two ascending runs of s / 2 starts by comparing element e with
// e ^ (s - 1) -- the second run read backwards -- and goes on with the strides s / 4 ... 1; the smaller key always goes to the
// lower index.  Round 3's form sorted every other run downwards instead, so which of (min, max) a slot keeps depended on its
// lane in every stage, in-lane ones included: two selects per in-lane exchange that this form does not have (count_wave_kernel:
// ~150 of ~800 vector instructions a read).
template <int R, typename K>
__device__ __forceinline__ void wave_bitonic_sort(K (&key)[R], int lane) {
  constexpr int N = 64 * R;
#pragma unroll
  for (int s = 2; s <= N; s <<= 1) {
    if (s <= R) {  // the mirror stage inside a lane
#pragma unroll
      for (int r = 0; r < R; ++r)
        if ((r & (s >> 1)) == 0) {
          const K a = key[r], b = key[r ^ (s - 1)];
          key[r] = key_min(a, b);
          key[r ^ (s - 1)] = key_max(a, b);
        }
    } else {  // ... across the lanes of a group of G = s / R: (lane, r) against (lane ^ (G - 1), R - 1 - r)
      const int G = s / R;
      const bool lower = (lane & (G >> 1)) == 0;
      K other[R];
#pragma unroll
      for (int r = 0; r < R; ++r) other[r] = wave_mirror(key[R - 1 - r], G);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const K mn = key_min(key[r], other[r]), mx = key_max(key[r], other[r]);
        key[r] = lower ? mn : mx;
      }
    }
#pragma unroll
    for (int t = s >> 2; t > 0; t >>= 1) {
      if (t >= R) {
        const int lt = t / R;
        const bool lower = (lane & lt) == 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          K mn, mx;
          if (lt == 4) {  // a lower lane's partner is four lanes up its row, an upper lane's four down: a shift each, no xor of two moves
            mn = key_min(key[r], wave_row_shift<0x104>(key[r]));
            mx = key_max(key[r], wave_row_shift<0x114>(key[r]));
          } else {
            const K other = wave_shfl_xor(key[r], lt);
            mn = key_min(key[r], other), mx = key_max(key[r], other);
          }
          key[r] = lower ? mn : mx;
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
          if ((r & t) == 0) {
            const K a = key[r], b = key[r ^ t];
            key[r] = key_min(a, b);
            key[r ^ t] = key_max(a, b);
          }
      }
    }
  }
}

// The last phase of the network alone, on keys held STRIPED (element index e = r * 64 + lane): puts a BITONIC sequence
// (one that falls and then rises, or rises and then falls) in ascending order in log2(64 R) stages.
template <int R, typename K>
__device__ __forceinline__ void wave_bitonic_merge_striped(K (&key)[R], int lane) {
  static_assert((R & (R - 1)) == 0, "a power of two");
#pragma unroll
  for (int t = 32 * R; t > 0; t >>= 1) {
    if (t >= 64) {
      const int rt = t / 64;
#pragma unroll
      for (int r = 0; r < R; ++r)
        if ((r & rt) == 0) {
          const K a = key[r], b = key[r ^ rt];
          key[r] = key_min(a, b);
          key[r ^ rt] = key_max(a, b);
        }
    } else {
      const bool lower = (lane & t) == 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const K other = wave_shfl_xor(key[r], t);
        const K mn = key_min(key[r], other), mx = key_max(key[r], other);
        key[r] = lower ? mn : mx;
      }
    }
  }
}

// the same network on (key, value) pairs ordered by key, then value
template <int R, typename K>
__device__ __forceinline__ void wave_bitonic_sort_pairs(K (&key)[R], uint32_t (&val)[R], int lane) {
  constexpr int N = 64 * R;
  auto less = [](K ka, uint32_t va, K kb, uint32_t vb) { return ka < kb || (ka == kb && va < vb); };
#pragma unroll
  for (int s = 2; s <= N; s <<= 1) {
#pragma unroll
    for (int t = s >> 1; t > 0; t >>= 1) {
      if (t >= R) {
        const int lt = t / R;
        const bool asc = (s == N) ? true : ((lane & (s / R)) == 0);
        const bool lower = (lane & lt) == 0;
        const bool keep_min = (lower == asc);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const K ok = wave_shfl_xor(key[r], lt);
          const uint32_t ov = wave_shfl_xor(val[r], lt);
          const bool mine_less = less(key[r], val[r], ok, ov);
          const bool take_other = keep_min ? !mine_less : mine_less;
          key[r] = take_other ? ok : key[r];
          val[r] = take_other ? ov : val[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((r & t) == 0) {
            const bool asc = (s >= N) ? true : (s < R ? ((r & s) == 0) : ((lane & (s / R)) == 0));
            const K a = key[r], b = key[r ^ t];
            const uint32_t va = val[r], vb = val[r ^ t];
            const bool swap = asc ? less(b, vb, a, va) : less(a, va, b, vb);
            key[r] = swap ? b : a;
            val[r] = swap ? vb : va;
            key[r ^ t] = swap ? a : b;
            val[r ^ t] = swap ? va : vb;
          }
        }
      }
    }
  }
}

// After wave_bitonic_sort: collapse runs of equal keys.
//   s_key[u]   = u-th distinct key (ascending), u < n_unique
//   s_start[u] = index of its first occurrence; s_start[n_unique] = n_valid
// so its multiplicity is s_start[u+1]-s_start[u].  s_key/s_start are this
// wave's private LDS regions (64*R and 64*R+1 entries).  Returns n_unique.
template <int R, typename K>
__device__ __forceinline__ uint32_t wave_unique(const K (&key)[R], K sentinel, int lane, K *s_key,
                                                uint32_t *s_start, uint32_t &n_valid) {
  K prev = wave_shfl_up1(key[R - 1]);
  bool head[R];
  uint32_t below = 0, total = 0, valid = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    K p = (r == 0) ? prev : key[r - 1];
    bool first = (r == 0) && (lane == 0);
    head[r] = (key[r] != sentinel) && (first || key[r] != p);
    uint64_t m = __ballot(head[r]);
    below += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    total += (uint32_t)__popcll(m);
    valid += (uint32_t)__popcll(__ballot(key[r] != sentinel));
  }
  uint32_t pos = below;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (head[r]) {
      s_key[pos] = key[r];
      s_start[pos] = (uint32_t)(lane * R + r);
      ++pos;
    }
  }
  if (lane == 0) s_start[total] = valid;
  n_valid = valid;
  __builtin_amdgcn_wave_barrier();
  return total;
}

// the number of distinct keys only (no LDS)
template <int R, typename K>
__device__ __forceinline__ uint32_t wave_unique_count(const K (&key)[R], K sentinel, int lane) {
  K prev = wave_shfl_up1(key[R - 1]);
  uint32_t total = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    K p = (r == 0) ? prev : key[r - 1];
    const bool first = (r == 0) && (lane == 0);
    const bool head = (key[r] != sentinel) && (first || key[r] != p);
    total += (uint32_t)__popcll(__ballot(head));
  }
  return total;
}

}  // namespace kpop

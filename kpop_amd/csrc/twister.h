// twister.h -- device-resident twister.
//
// Reference layout (lib/Twister.ml:22-25, BiOCamLib Matrix.t): n_dims separate Float.Arrays of n_cols
// coefficients ("dims-major"), plus a Hashtbl from k-mer name to column (lib/Twister.ml:71-76).
//
// Layout in HBM, chosen for the gather the twist performs:
//   rows : [n_rows][d_pad] f64, k-mer-major -- one k-mer's coefficients are one contiguous, 128-byte aligned
//          row, so a wave reads a row with one fully coalesced load per 64 dims; d_pad = n_dims rounded up to
//          16.  Rows are stored in ASCENDING HASH ORDER whatever the column order of the twister file (the
//          loader permutes them; of several columns carrying the same name only the last is reachable in the
//          reference -- Hashtbl.add shadows -- and only that one is kept), so a k-mer's row index is its RANK
//          among the twister's k-mers and the kernels add rows in ascending hash order.
//   rsel : k <= 16: RankWord[4^k/64], rank-select form of the name -> row map: 64 presence bits + the number
//          of present k-mers before the word; row = prefix + popcount(bits below).  4 MB at k=12 (a u32 LUT
//          would be 67 MB and gets evicted by the row stream: +24 % HBM traffic, measured in round 1).
//   rblk : k = 15, 16: the same map in BLOCKS of one 64-byte sector -- 480 presence bits (fifteen dwords) + the number of
//          present k-mers before the block: 143 MB at k = 15 where the 16-byte words above are 268 MB, more than the whole
//          256 MB Infinity Cache.  Behind every window's look-up comes a random 128-byte row from a table of tens of GB (an HBM
//          miss by construction); with the index inside the cache the look-up in front of it no longer is one (BASELINE config 5).
//          row = prefix + popcount(the block's bits below the k-mer's).  (The tile route for assemblies reads rank WORDS: it is
//          off at these k.)
//   direct : (round 5) a NEARLY COMPLETE twister of large k and few dimensions also keeps its rows at the address the hash itself
//          names -- [4^k][d_pad], a row that does not exist marked by kDirectAbsent in its first double.  The fused kernel for
//          reads then has no name -> row look-up at all: BASELINE config 5 (k = 15, D = 16, every canonical 15-mer) moved a
//          128-byte line of index for every 128-byte row (1.96 x the algorithmic bytes, measured; two dependent misses a
//          window).  137 GB at k = 15 beside the 69 GB of rows in rank order (which every other kernel keeps using): what 288 GB
//          of HBM are for.  Built when at least 0.45 of the hashes have a row (every canonical k-mer of a double-stranded
//          twister is 0.5), d_pad <= 32, and the table fits what is free with room to spare; kpop_tune("direct", 0) opts out.
//   sorted_hash : k > 16: the ascending hashes themselves, searched by bisection; the hit's index is the row.
#pragma once
#include <stdint.h>

#include <algorithm>

#include "common.h"

struct kpop_twister {
  int k = 0;   // k of the name -> row index
  int hk = 0;  // k the fused count->twist kernels hash reads with (kpop_twister_set_count_k; <= k)
  uint32_t n_dims = 0;
  uint32_t d_pad = 0;
  uint64_t n_cols = 0;  // columns of the twister as loaded
  uint64_t n_rows = 0;  // distinct k-mers = device rows
  double *d_rows = nullptr;
  void *d_rsel = nullptr;
  void *d_rblk = nullptr;  // k >= kRankBlockMinK: 64-byte blocks instead of d_rsel (which is then freed once the rows are placed)
  uint64_t *d_sorted_hash = nullptr;
  double *d_direct = nullptr;  // [direct_hi - direct_lo][d_pad]: the rows at their hashes (see above), or nullptr
  uint64_t direct_lo = 0, direct_hi = 0;  // the hashes the table covers: all 4^k of them, or the slice of a twister that keeps a range of k-mer rows (kpop_twister_synth_slice)
  uint64_t device_bytes = 0;
  int slot = 0;        // device slot (common.h) whose memory holds the arrays
  bool alias = false;  // a second handle on another twister's arrays (kpop_twister_replicate onto the same GPU): frees nothing
};

namespace kpop {

constexpr int kRankMaxK = 16;  // 4^16 / 64 words * 16 B = 1 GiB
constexpr int kRankBlockMinK = 15;         // from this k on the index is kept as 64-byte blocks
constexpr uint32_t kRankBlockBits = 480;   // presence bits of a block: dwords 0..14; dword 15: the rank of its first bit
static inline uint64_t rank_blocks(int k) { return ((1ull << (2 * k)) + kRankBlockBits - 1) / kRankBlockBits; }
constexpr uint32_t kNoCol = 0xFFFFFFFFu;
constexpr int kDirectMinK = 13;                                 // below: the index is a few MB and sits in L2
constexpr uint64_t kDirectAbsent = 0x7FF4D1EC7AB5E27Eull;      // first double of a row of the direct table that does not exist (a NaN no twister file spells)

struct RankWord {
  uint64_t bits;
  uint32_t prefix;
  uint32_t count;
};

struct TwisterView {
  const double *rows;
  const RankWord *rsel;
  const uint4 *rblk;  // [rank_blocks(k)][4]
  const uint64_t *sorted_hash;
  uint64_t n_rows;
  uint32_t n_dims;
  uint32_t d_pad;
  int k;
  int hk;
  const double *direct;  // rows at their hashes, or nullptr
  uint32_t direct_lo, direct_hi;  // ... of hashes direct_lo <= h < direct_hi (a hash outside has no row here)
};

static inline TwisterView view_of(const kpop_twister *tw) {
  return TwisterView{tw->d_rows, reinterpret_cast<const RankWord *>(tw->d_rsel), reinterpret_cast<const uint4 *>(tw->d_rblk), tw->d_sorted_hash, tw->n_rows,
                     tw->n_dims, tw->d_pad,                                     tw->k, tw->hk ? tw->hk : tw->k, (tw->hk == 0 || tw->hk == tw->k) ? tw->d_direct : nullptr,
                     (uint32_t)tw->direct_lo, (uint32_t)std::min<uint64_t>(tw->direct_hi, 0xFFFFFFFFull)};
}

#if defined(__HIPCC__)
// hash -> twister row, kNoCol when absent (lib/Twister.ml:151 Hashtbl.find_opt)
__device__ __forceinline__ uint32_t lookup_col(const TwisterView &tv, uint64_t h) {
  if (h >> (2 * tv.k)) return kNoCol;  // not a k-mer of this twister's k (caller-supplied spectra)
  if (tv.rblk) {  // one 64-byte sector: four 16-byte loads of consecutive addresses
    const uint32_t hh = (uint32_t)h, blk = hh / kRankBlockBits, r = hh - blk * kRankBlockBits, w = r >> 5, bit = r & 31u;
    const uint4 *p = tv.rblk + (uint64_t)blk * 4;
    const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    const uint32_t d[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    uint32_t below = 0, word = 0;
#pragma unroll
    for (uint32_t i = 0; i < 15; ++i) {
      below += i < w ? (uint32_t)__popc(d[i]) : 0u;
      word = i == w ? d[i] : word;
    }
    return ((word >> bit) & 1u) ? d[15] + below + (uint32_t)__popc(word & ((1u << bit) - 1u)) : kNoCol;
  }
  if (tv.rsel) {
    const uint4 q = *reinterpret_cast<const uint4 *>(tv.rsel + (h >> 6));  // one 16-byte load
    const uint64_t bits = ((uint64_t)q.y << 32) | q.x;
    const uint32_t b = (uint32_t)h & 63u;
    const uint64_t below = bits & ((1ull << b) - 1ull);
    return ((bits >> b) & 1ull) ? q.z + (uint32_t)__popcll(below) : kNoCol;
  }
  uint64_t lo = 0, hi = tv.n_rows;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (tv.sorted_hash[mid] < h) lo = mid + 1; else hi = mid;
  }
  return (lo < tv.n_rows && tv.sorted_hash[lo] == h) ? (uint32_t)lo : kNoCol;
}
#endif

}  // namespace kpop

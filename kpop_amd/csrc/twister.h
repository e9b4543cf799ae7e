// twister.h -- device-resident twister.
//
// Reference layout (lib/Twister.ml:22-25, BiOCamLib Matrix.t): n_dims separate
// Float.Arrays of n_cols coefficients ("dims-major"), plus a Hashtbl from
// k-mer name to column (lib/Twister.ml:71-76).
//
// Layout in HBM, chosen for the gather the twist performs:
//   rows : [n_cols][d_pad] f64, k-mer-major -- one k-mer's coefficients are one
//          contiguous, 128-byte aligned row, so a wave reads a row with one
//          fully coalesced load per 64 dims;  d_pad = n_dims rounded up to 16.
//   lut  : [4^k] u32, hash -> column (0xFFFFFFFF = k-mer not in the twister),
//          for k <= kLutMaxK; replaces the name Hashtbl.
//   sorted_hash/sorted_col : for larger k, a sorted table searched by bisection.
#pragma once
#include <stdint.h>

#include "common.h"

struct kpop_twister {
  int k = 0;
  uint32_t n_dims = 0;
  uint32_t d_pad = 0;
  uint64_t n_cols = 0;
  uint64_t n_sorted = 0;  // entries of the bisection table (k > kLutMaxK)
  double *d_rows = nullptr;
  uint32_t *d_lut = nullptr;
  void *d_rsel = nullptr;  // RankWord[4^k/64], only when columns ascend with the hash
  uint64_t *d_sorted_hash = nullptr;
  uint32_t *d_sorted_col = nullptr;
  uint64_t device_bytes = 0;
};

namespace kpop {

constexpr int kLutMaxK = 16;             // 4^16 * 4 B = 16 GiB of 288 GB
constexpr uint32_t kNoCol = 0xFFFFFFFFu;

// Rank-select form of the name -> column map, usable when the twister's columns
// ascend with the k-mer hash (then column = rank of the hash among present
// k-mers): one 16-byte word per 64 consecutive hashes = presence bits + the
// number of present k-mers before the word.  4 MB at k=12 against the 67 MB
// LUT: small enough to stay cache-resident next to the streamed twister rows.
struct RankWord {
  uint64_t bits;
  uint32_t prefix;
  uint32_t pad;
};

struct TwisterView {
  const double *rows;
  const RankWord *rsel;
  const uint32_t *lut;
  const uint64_t *sorted_hash;
  const uint32_t *sorted_col;
  uint64_t n_cols;
  uint64_t n_sorted;
  uint32_t n_dims;
  uint32_t d_pad;
  int k;
};

static inline TwisterView view_of(const kpop_twister *tw) {
  return TwisterView{tw->d_rows,        reinterpret_cast<const RankWord *>(tw->d_rsel),
                     tw->d_lut,         tw->d_sorted_hash,
                     tw->d_sorted_col,  tw->n_cols,
                     tw->n_sorted,      tw->n_dims,
                     tw->d_pad,         tw->k};
}

#if defined(__HIPCC__)
// hash -> twister column, kNoCol when absent (lib/Twister.ml:151 Hashtbl.find_opt)
__device__ __forceinline__ uint32_t lookup_col(const TwisterView &tv, uint64_t h) {
  if (h >> (2 * tv.k)) return kNoCol;  // not a k-mer of this twister's k (caller-supplied spectra)
  if (tv.rsel) {
    const uint4 q = *reinterpret_cast<const uint4 *>(tv.rsel + (h >> 6));  // one 16-byte load
    const uint64_t bits = ((uint64_t)q.y << 32) | q.x;
    const uint32_t b = (uint32_t)h & 63u;
    const uint64_t below = bits & ((1ull << b) - 1ull);
    return ((bits >> b) & 1ull) ? q.z + (uint32_t)__popcll(below) : kNoCol;
  }
  if (tv.lut) return tv.lut[h];
  uint64_t lo = 0, hi = tv.n_sorted;
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (tv.sorted_hash[mid] < h) lo = mid + 1; else hi = mid;
  }
  return (lo < tv.n_sorted && tv.sorted_hash[lo] == h) ? tv.sorted_col[lo] : kNoCol;
}
#endif

}  // namespace kpop

// kmer.h -- the k-mer encoding of this implementation, in ONE place.
//
// The reference keeps it in BiOCamLib's KMers.DNAHash* (call sites
// bin/KPopCount.ml:38,46,241-245), whose source is not part of the reference
// checkout; the encoding is therefore DECLARED here and
// can be swapped by editing this header only:
//   - base code:  A/a 0, C/c 1, G/g 2, T/t 3; anything else breaks the window
//   - hash:       big-endian 2-bit packing (first base most significant)
//   - DNA-ds key: min(hash(fwd), hash(reverse complement))
//   - name:       lowercase hex, zero padded to ceil(k/2) digits
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KPOP_HD __host__ __device__ __forceinline__
#else
#define KPOP_HD inline
#endif

namespace kpop {

constexpr int kMaxK = 30;  // bin/KPopCount.ml:113

// 0..3 for ACGT in either case, 4 for everything else.
KPOP_HD uint32_t base_code(uint32_t c) {
  uint32_t u = c & 0xDFu;  // fold case
  uint32_t x = (u >> 1) & 3u;  // A0 C1 T2 G3
  uint32_t code = x ^ (x >> 1);  // A0 C1 G2 T3
  bool ok = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
  return ok ? code : 4u;
}

KPOP_HD uint64_t kmer_mask(int k) { return (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1ull); }

KPOP_HD uint64_t revcomp(uint64_t h, int k) {
  // complement = 3 - code = bitwise not on 2 bits; then reverse the 2-bit groups
  uint64_t x = ~h;
  x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
  x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
  x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
  x = (x >> 32) | (x << 32);
  return x >> (64 - 2 * k);
}

inline int hex_digits(int k) { return (k + 1) / 2; }

}  // namespace kpop

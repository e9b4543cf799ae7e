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
// Protein k-mers (KMers.ProteinHash, bin/KPopCount.ml:246-248; counting only, k <= 12):
//   - residue code: the 20 standard amino acids in alphabetical order of their one-letter codes
//                   (A0 C1 D2 E3 F4 G5 H6 I7 K8 L9 M10 N11 P12 Q13 R14 S15 T16 V17 W18 Y19), either case;
//                   anything else (B J O U X Z * ...) breaks the window
//   - hash:         big-endian 5-bit packing; no reverse complement
//   - name:         lowercase hex, zero padded to ceil(5k/4) digits
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KPOP_HD __host__ __device__ __forceinline__
#else
#define KPOP_HD inline
#endif

namespace kpop {

constexpr int kMaxK = 30;         // bin/KPopCount.ml:113
constexpr int kMaxKProtein = 12;  // same line: "<= 12 for protein"
constexpr int kContentProtein = 2;  // KPOP_PROTEIN of include/kpop_hip.h

KPOP_HD int symbol_bits(int content) { return content == kContentProtein ? 5 : 2; }
KPOP_HD int hash_bits(int k, int content) { return symbol_bits(content) * k; }
KPOP_HD uint64_t bits_mask(int bits) { return (bits >= 64) ? ~0ull : ((1ull << bits) - 1ull); }

// 0..19 for the standard amino acids in either case, 31 for everything else.  The 26-letter table
//   A0 B- C1 D2 E3 F4 G5 H6 I7 J- K8 L9 | M10 N11 O- P12 Q13 R14 S15 T16 U- V17 W18 X- | Y19 Z-
// is packed 5 bits per letter, 12 letters per word.
KPOP_HD uint32_t protein_code(uint32_t c) {
  const uint32_t u = c & 0xDFu;  // fold case
  if (u < 'A' || u > 'Z') return 31u;
  const uint32_t i = u - 'A';
  const uint64_t w = i < 12 ? 0x4a3e731483107e0ull : (i < 24 ? 0xfca3f83dcd67d6aull : 0xffffffffffffff3ull);
  return (uint32_t)(w >> (5u * (i % 12u))) & 31u;
}

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
inline int hex_digits_bits(int bits) { return (bits + 3) / 4; }

}  // namespace kpop

// read_hash.h -- one wavefront turns one read into its k-mer keys.
//
// Replaces KIH.iterc (bin/KPopCount.ml:38; KMers.DNAHash* in BiOCamLib, absent):
// slide a k-window over the read, 2-bit encode, keep min(fwd, revcomp) for
// DNA-ds, skip every window that holds a non-ACGT symbol.
//
// Lane `lane` owns the R consecutive window starts lane*R .. lane*R+R-1 and
// rolls the hash across them, so a read of up to 64*R windows costs each lane
// k-1+R base reads from the wave's LDS staging area.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kpop_hip.h"
#include "kmer.h"

namespace kpop {

// bytes of LDS staging a wave needs for reads of up to 64*R windows
template <int R>
constexpr int codes_bytes() { return 64 * R + 32; }

// Stage the read as base codes (0..3, 4 = breaks the window) into s_codes;
// everything past the read's end is 4, so windows running off the end are
// rejected by the same test as windows over an N.
// SB = bits per symbol: 2 = DNA (codes 0..3, 4 breaks), 5 = protein (codes 0..19, 31 breaks)
template <int R, int SB = 2>
__device__ __forceinline__ void wave_stage_codes(const uint8_t *__restrict__ seq, uint32_t len, int lane,
                                                 uint8_t *s_codes) {
  constexpr int NB = codes_bytes<R>();
  constexpr uint32_t kBreak = SB == 2 ? 4u : 31u;
#pragma unroll
  for (int i = 0; i < (NB + 63) / 64; ++i) {
    int p = i * 64 + lane;
    if (p < NB) {
      uint32_t c = kBreak;
      if ((uint32_t)p < len) c = SB == 2 ? base_code(seq[p]) : protein_code(seq[p]);
      s_codes[p] = (uint8_t)c;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// The same from the PACKED form of a batch (packed.hip: 16 bases a word of 2-bit codes, first base least significant; one bit a base
// that is none of ACGTacgt): base `off + p` of the batch to lane p -- a read of 150 bases is ten words of codes and five of marks
// where it was 150 bytes (BASELINE north_star: "coalesced HBM loads of packed bases")
template <int R>
__device__ __forceinline__ void wave_stage_codes_packed(const uint32_t *__restrict__ codes, const uint32_t *__restrict__ invalid, uint64_t off, uint32_t len,
                                                        int lane, uint8_t *s_codes) {
  constexpr int NB = codes_bytes<R>();
#pragma unroll
  for (int i = 0; i < (NB + 63) / 64; ++i) {
    int p = i * 64 + lane;
    if (p < NB) {
      uint32_t c = 4u;
      if ((uint32_t)p < len) {
        const uint64_t b = off + (uint64_t)p;
        const uint32_t bad = (invalid[b >> 5] >> (uint32_t)(b & 31u)) & 1u;
        c = bad ? 4u : (codes[b >> 4] >> (2u * (uint32_t)(b & 15u))) & 3u;
      }
      s_codes[p] = (uint8_t)c;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// H = uint32_t for k <= 16, uint64_t above.  key[r] = canonical hash of window
// lane*R+r, or the all-ones sentinel when the window is invalid.
template <int R, typename H, int SB = 2>
__device__ __forceinline__ void wave_hash_windows(const uint8_t *s_codes, int k, int content, int lane,
                                                  H (&key)[R]) {
  const H mask = (H)bits_mask(SB * k);
  const int shift = SB * (k - 1);
  constexpr uint32_t kSym = (1u << SB) - 1u, kValid = SB == 2 ? 4u : 20u;
  const uint8_t *p = s_codes + lane * R;
  H fwd = 0, rc = 0;
  int run = 0;
  for (int j = 0; j < k - 1; ++j) {
    uint32_t c = p[j];
    fwd = ((fwd << SB) | (H)(c & kSym)) & mask;
    if (SB == 2) rc = (rc >> 2) | ((H)(3u - (c & 3u)) << shift);
    run = (c < kValid) ? run + 1 : 0;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    uint32_t c = p[k - 1 + r];
    fwd = ((fwd << SB) | (H)(c & kSym)) & mask;
    if (SB == 2) rc = (rc >> 2) | ((H)(3u - (c & 3u)) << shift);
    run = (c < kValid) ? run + 1 : 0;
    H canon = (SB == 2 && content == KPOP_DNA_DS && rc < fwd) ? rc : fwd;
    key[r] = (run >= k) ? canon : (H)~(H)0;
  }
}

}  // namespace kpop

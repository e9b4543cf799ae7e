// packed.hip -- sequences in 2.25 bits a base at the boundary (BASELINE north_star: "coalesced HBM loads of packed bases").
//
// Every entry point of the count -> twist path took one ASCII byte a base: a host caller of BASELINE config 3 (50,000 assemblies
// of 30 kb, 1.5 GB) is then bound by the bus -- 26 ms at the 57 GB/s this link delivers where the kernels need 9.6.  A host that
// KEEPS its sequences packed (a database of assemblies) sends
//   codes   : 16 bases a 32-bit word, two bits a base (A 0, C 1, G 2, T 3, either case), base i of the batch in bits 2 (i % 16) ..
//             of word i / 16 -- first base LEAST significant, the order tile_pipe.h stages the complement strand in;
//   invalid : one bit a base that is none of ACGTacgt (N, IUPAC codes, dashes, anything else: bin/KPopCount.ml:242-245's
//             Lint.dnaize turns them all into bases that start no k-mer), 32 bases a word
// -- 3.56 times fewer bytes over the bus.  On the device kpop_dev_unpack_bases spreads them back to one byte a base (ACGT, N for an
// invalid one) at HBM's rate (1.5 GB in 0.3-0.4 ms: 1.28 bytes moved a base) and the kernels run as they are: what they hash is the
// same letters, so every result is the ASCII path's bit for bit.  A batch of READS (all of up to 512 windows: the headline's kind) skips
// even that: count_twist_wave_kernel<..., PACKED> stages a read's 2-bit codes straight from the words.  (The genome kernels still take
// bytes: the tile kernel re-codes its stretches as it stages them; the bus was the bound, not that.)
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "common.h"
#include "twister.h"

// count_twist.hip
int count_twist_wave_from_packed(const kpop_twister *tw, const uint32_t *d_codes, const uint32_t *d_invalid, const uint64_t *d_offsets, uint32_t n_reads,
                                 uint32_t max_len, int content, int normalize, double *d_out, hipStream_t st, int *done);

namespace kpop {

// sixteen output bytes a thread: bases 16 t .. 16 t + 15 of the slice that starts `cshift` bases into codes[0] and `mshift` bases
// into invalid[0] (a chunk of a batch starts anywhere)
__global__ __launch_bounds__(256) void unpack_bases_kernel(const uint32_t *__restrict__ codes, uint32_t cshift, const uint32_t *__restrict__ invalid,
                                                           uint32_t mshift, uint64_t n_bases, uint8_t *__restrict__ out) {
  const uint64_t n16 = (n_bases + 15) / 16;
  for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < n16; t += (uint64_t)gridDim.x * 256) {
    const uint64_t gc = 16 * t + cshift, gm = 16 * t + mshift;
    const uint32_t c0 = codes[gc >> 4], c1 = (gc & 15u) ? codes[(gc >> 4) + 1] : 0u;
    const uint32_t cw = (uint32_t)((((uint64_t)c1 << 32) | c0) >> (2u * (uint32_t)(gc & 15u)));
    const uint32_t m0 = invalid[gm >> 5], m1 = ((gm & 31u) > 16u) ? invalid[(gm >> 5) + 1] : 0u;
    const uint32_t mw = (uint32_t)((((uint64_t)m1 << 32) | m0) >> (uint32_t)(gm & 31u)) & 0xFFFFu;
    uint32_t w[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
      uint32_t word = 0;
#pragma unroll
      for (uint32_t b = 0; b < 4; ++b) {
        const uint32_t i = 4 * q + b, code = (cw >> (2 * i)) & 3u;
        const uint32_t letter = ((mw >> i) & 1u) ? (uint32_t)'N' : (0x54474341u >> (8 * code)) & 0xFFu;  // "ACGT"[code]
        word |= letter << (8 * b);
      }
      w[q] = word;
    }
    if (16 * t + 16 <= n_bases)
      *reinterpret_cast<uint4 *>(out + 16 * t) = make_uint4(w[0], w[1], w[2], w[3]);
    else
      for (uint64_t i = 16 * t; i < n_bases; ++i) out[i] = (uint8_t)(w[(i - 16 * t) >> 2] >> (8 * ((i - 16 * t) & 3)));
  }
}

int launch_unpack_bases(const uint32_t *d_codes, uint32_t cshift, const uint32_t *d_invalid, uint32_t mshift, uint64_t n_bases, uint8_t *d_out, hipStream_t st) {
  if (!n_bases) return 0;
  unpack_bases_kernel<<<dim3((uint32_t)std::min<uint64_t>(div_up((n_bases + 15) / 16, 256), 1u << 16)), dim3(256), 0, st>>>(d_codes, cshift, d_invalid, mshift,
                                                                                                                              n_bases, d_out);
  KPOP_LAUNCH_CHECK();
  return 0;
}

// the host's packer: eight bytes at a time (the letters' bits 1 and 2 ARE a 2-bit code up to a swap of G and T; a byte is a base
// when folding its case and spelling the code back gives the byte), a thread a slice of whole words
static void pack_slice(const uint8_t *bases, uint64_t lo, uint64_t hi, uint64_t n_bases, uint32_t *codes, uint32_t *invalid) {
  for (uint64_t b = lo; b < hi; b += 32) {  // lo is a multiple of 32: one invalid word, two code words
    uint32_t cw[2] = {0u, 0u}, mw = 0u;
    const uint64_t e = std::min(n_bases, b + 32);
    uint64_t i = b;
    for (; i + 8 <= e; i += 8) {
      uint64_t x;
      memcpy(&x, bases + i, 8);
      const uint64_t u = x & 0xDFDFDFDFDFDFDFDFull;                    // fold case
      const uint64_t t = (u >> 1) & 0x0303030303030303ull;            // A0 C1 T2 G3
      const uint64_t c = t ^ ((t >> 1) & 0x0101010101010101ull);      // A0 C1 G2 T3
      // "ACGT"[code], a byte each: 0x41 + code-dependent offsets {0, 2, 6, 19}
      const uint64_t lo1 = c & 0x0101010101010101ull, hi1 = (c >> 1) & 0x0101010101010101ull;
      const uint64_t back = 0x4141414141414141ull + lo1 * 2 + hi1 * 6 + (lo1 & hi1) * 11;  // A 0x41, C 0x43, G 0x47, T 0x54
      const uint64_t d = back ^ u;
      const uint64_t bad = (((d & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | d) & 0x8080808080808080ull;
      uint64_t g = c | (c >> 6);  // the eight 2-bit codes pushed together: pairs, then fours, then the two halves
      g |= g >> 12;
      const uint32_t c8 = (uint32_t)(g & 0xFFu) | ((uint32_t)(g >> 32) & 0xFFu) << 8;
      const uint32_t m8 = (uint32_t)((((bad >> 7) * 0x0102040810204080ull)) >> 56);  // bit 7 of byte k -> bit k
      const uint32_t at = (uint32_t)(i - b);
      cw[at >> 4] |= c8 << (2 * (at & 15u));
      mw |= m8 << at;
    }
    for (; i < e; ++i) {
      const uint8_t u = bases[i] & 0xDFu;
      const uint32_t code = u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : 3u;
      const uint32_t at = (uint32_t)(i - b);
      cw[at >> 4] |= code << (2 * (at & 15u));
      if (!(u == 'A' || u == 'C' || u == 'G' || u == 'T')) mw |= 1u << at;
    }
    if (e < b + 32) mw |= e - b >= 32 ? 0u : ~0u << (uint32_t)(e - b);  // (past the batch's end: no bases)
    codes[b >> 4] = cw[0];
    if (b + 16 < ((n_bases + 15) & ~15ull)) codes[(b >> 4) + 1] = cw[1];
    invalid[b >> 5] = mw;
  }
}

}  // namespace kpop

using namespace kpop;

extern "C" uint64_t kpop_packed_code_words(uint64_t n_bases) { return (n_bases + 15) / 16; }
extern "C" uint64_t kpop_packed_mask_words(uint64_t n_bases) { return (n_bases + 31) / 32; }

extern "C" int kpop_pack_bases(const uint8_t *bases, uint64_t n_bases, uint32_t *codes, uint32_t *invalid, int threads) {
  if (n_bases && (!bases || !codes || !invalid)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pack_bases: null argument");
  if (!n_bases) return KPOP_OK;
  unsigned nt = threads > 0 ? (unsigned)threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  const uint64_t per = ((n_bases + nt - 1) / nt + 4095) & ~4095ull;  // slices of whole words
  if (per >= n_bases) nt = 1;
  if (nt == 1) {
    pack_slice(bases, 0, n_bases, n_bases, codes, invalid);
    return KPOP_OK;
  }
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nt && (uint64_t)t * per < n_bases; ++t)
    pool.emplace_back(pack_slice, bases, (uint64_t)t * per, std::min(n_bases, (uint64_t)(t + 1) * per), n_bases, codes, invalid);
  for (std::thread &th : pool) th.join();
  return KPOP_OK;
}

extern "C" int kpop_dev_unpack_bases(const uint32_t *d_codes, const uint32_t *d_invalid, uint64_t n_bases, uint8_t *d_bases, void *stream) {
  KPOP_TRY(require_init());
  if (n_bases && (!d_codes || !d_invalid || !d_bases)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_unpack_bases: null argument");
  return launch_unpack_bases(d_codes, 0, d_invalid, 0, n_bases, d_bases, as_stream(stream));
}

extern "C" int kpop_dev_count_twist_packed(const kpop_twister *tw, const uint32_t *d_codes, const uint32_t *d_invalid, const uint64_t *d_offsets, uint32_t n_reads,
                                           uint64_t n_bases, uint32_t max_len, int content, int normalize, double *d_out, void *stream) {
  KPOP_TRY(require_init());
  if (!tw || !d_offsets || !d_out || (n_bases && (!d_codes || !d_invalid))) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_count_twist_packed: null argument");
  if (n_reads == 0) return KPOP_OK;
  hipStream_t st = as_stream(stream);
  // reads that all fit the one-wavefront-per-read kernel (the headline's kind of batch) are twisted straight from the words: a read of 150
  // bases is fifteen words where it was 150 bytes, and no bytes are made at all
  if (content != KPOP_DNA_DS && content != KPOP_DNA_SS)
    KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_dev_count_twist_packed: content %d (the packed form is DNA)", content);
  int done = 0;
  KPOP_TRY(count_twist_wave_from_packed(tw, d_codes, d_invalid, d_offsets, n_reads, max_len, content, normalize, d_out, st, &done));
  if (done) return KPOP_OK;
  void *bytes = nullptr;
  KPOP_TRY(ctx().ws2_for(st).ensure(n_bases + 64, &bytes));  // (the library's SECOND block of the stream: kpop_dev_count_twist takes the first)
  KPOP_TRY(launch_unpack_bases(d_codes, 0, d_invalid, 0, n_bases, reinterpret_cast<uint8_t *>(bytes), st));
  return kpop_dev_count_twist(tw, reinterpret_cast<const uint8_t *>(bytes), d_offsets, n_reads, n_bases, max_len, content, normalize, d_out, stream);
}

extern "C" int kpop_count_twist_packed(const kpop_twister *tw, const uint32_t *codes, const uint32_t *invalid, const uint64_t *offsets, uint32_t n_reads,
                                       int content, int normalize, double *out) {
  KPOP_TRY(require_init());
  ArenaScope scratch;
  if (!tw || !offsets || (!out && n_reads)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_count_twist_packed: null argument");
  if (n_reads == 0) return KPOP_OK;
  uint64_t max_len = 0;
  for (uint32_t r = 0; r < n_reads; ++r) {
    if (offsets[r + 1] < offsets[r]) KPOP_FAIL(KPOP_ERR_INVALID, "offsets are not non-decreasing at read %u", r);
    max_len = std::max(max_len, offsets[r + 1] - offsets[r]);
  }
  if (max_len > 0xFFFFFFFFull) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_count_twist_packed: sequence longer than 2^32 bases");
  const uint64_t n_bases = offsets[n_reads];  // (the packed arrays hold the batch from its base 0: offsets[0] may be past it)
  if (n_bases && (!codes || !invalid)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_count_twist_packed: null argument");
  hipStream_t st = nullptr;
  DevBuf d_codes, d_mask, d_off, d_out;
  const uint64_t cw = kpop_packed_code_words(n_bases), mw = kpop_packed_mask_words(n_bases);
  KPOP_TRY(d_codes.alloc((cw + 2) * 4));
  KPOP_TRY(d_mask.alloc((mw + 2) * 4));
  KPOP_TRY(d_off.alloc((uint64_t)(n_reads + 1) * 8));
  KPOP_TRY(d_out.alloc((uint64_t)n_reads * tw->n_dims * 8));
  if (n_bases) {
    KPOP_HIP(hipMemcpyAsync(d_codes.p, codes, cw * 4, hipMemcpyHostToDevice, st));
    KPOP_HIP(hipMemcpyAsync(d_mask.p, invalid, mw * 4, hipMemcpyHostToDevice, st));
  }
  KPOP_HIP(hipMemcpyAsync(d_off.p, offsets, (uint64_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
  KPOP_TRY(kpop_dev_count_twist_packed(tw, d_codes.as<uint32_t>(), d_mask.as<uint32_t>(), d_off.as<uint64_t>(), n_reads, n_bases, (uint32_t)max_len, content,
                                       normalize, d_out.as<double>(), st));
  KPOP_HIP(hipMemcpyAsync(out, d_out.p, (uint64_t)n_reads * tw->n_dims * 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  return KPOP_OK;
}

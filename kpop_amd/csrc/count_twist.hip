// count_twist.hip -- the count and twist stages, and their fusion.
//
//   count_wave_kernel        bin/KPopCount.ml:36-50 (-L: one spectrum per read)
//   twist_csr_kernel         lib/Twister.ml:146-188 on CSR spectra
//   count_twist_wave_kernel  both, fused: the README.md:606 pipeline without
//                            the text spectra in between
//
// Work decomposition: ONE WAVEFRONT PER READ.  Reads are independent through
// count and twist, a 150 bp read has 139 windows (<= 64*R with
// R = 4 keys per lane), and the twist of one read is a gather of <= 139 twister
// rows of n_dims f64 -- with lane d owning dimension d, every row is one
// fully coalesced 512-byte load (n_dims = 64) and no cross-lane reduction is
// ever needed.  The kernel is HBM-bound on that gather.
#include <algorithm>
#include <vector>

#include "kmer.h"
#include "lookback.h"
#include "read_hash.h"
#include "scan.h"
#include "sort_count.h"
#include "twister.h"
#include "wave_sort.h"

namespace kpop {

constexpr int kWavesPerBlock = 4;
constexpr int kGatherUnroll = 8;   // row loads in flight per wave in the streaming kernels
constexpr int kGatherPad = 16;     // LDS padding: the deepest gather unroll
constexpr uint64_t kCsrSegLines = 8192;  // a spectrum's sums are formed per stretch of this many lines and the stretches added in order, whichever kernel
constexpr uint32_t kFewSpectra = 8192;  // up to this many spectra (a wave each: one round of the chip) twist_csr_kernel keeps 32 row loads in flight per wave; with more, eight and more resident waves do better (20,000 x 139 lines: 0.194 -> 0.161 ms, 32,768 x 139: 0.311 -> 0.247 -- the threshold was 32,768)

// ---------------------------------------------------------------------------
// per-wave LDS
// ---------------------------------------------------------------------------
template <int R, typename K>
struct WaveLds {
  K key[64 * R + kGatherPad];             // distinct keys (twister columns / hashes), ascending
  uint32_t start[64 * R + 4];             // first index of each run; [n_unique] = n_valid
  double x[64 * R + kGatherPad];          // normalised multiplicities
  uint8_t codes[(codes_bytes<R>() + 15) & ~15];
};

// ---------------------------------------------------------------------------
// count: one wave per read -> CSR of (hash, count), in ONE launch.
//
// A read's place in the CSR is the number of distinct k-mers of all reads before it, which no block knows by itself.
// Blocks take a ticket (the order in which they start running) and own the four reads of that ticket, so "before" means
// "smaller ticket", and hand their totals on through one 8-byte word per block -- {status, value} in a single store,
// so the word IS the message and no fence orders anything (MI355X_MICROARCH.md, hand-off granules):
//   status 1: value = distinct k-mers of this block's reads          (published as soon as they are counted)
//   status 2: value = distinct k-mers of this and all earlier blocks (published once the block knows its prefix)
// A block looks back over its predecessors 64 at a time (one wave, one word per lane) until it meets a status-2 word.
// A predecessor has a smaller ticket, hence is already running: waiting for it cannot deadlock, whatever the dispatch
// order.  Round 1 wrote fixed-stride scratch rows, scanned the per-read counts in three launches and compacted in a
// fifth: 12 B x 256 slots per read written and read back.
// ---------------------------------------------------------------------------
// reads per wave: the sorted keys of a wave's reads wait in registers (R per lane and read) for the block's prefix
template <int R>
constexpr int reads_per_wave() { return R <= 2 ? 8 : (R == 4 ? 8 : 2); }

template <int R, typename H, int SB = 2>
__global__ __launch_bounds__(64 * kWavesPerBlock, sizeof(H) == 4 ? 4 : 1) void count_wave_kernel(
    const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets, uint32_t n, int k, int content,
    uint32_t *__restrict__ ticket_counter, uint64_t *__restrict__ state, uint64_t *__restrict__ gstate, uint32_t *__restrict__ garr,
    uint64_t *__restrict__ out_hash, uint32_t *__restrict__ out_count, uint64_t *__restrict__ out_offsets, int dbg = 0) {
  constexpr int RW = reads_per_wave<R>(), RPB = RW * kWavesPerBlock;
  __shared__ WaveLds<R, H> lds[kWavesPerBlock];
  __shared__ uint32_t s_ticket;
  __shared__ uint32_t s_nu[RPB];
  __shared__ uint64_t s_prefix;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // one ticket per RPB reads: a single counter hands out ~90 tickets per microsecond on this part, which at four reads a
  // ticket was slower than the counting itself
  if (threadIdx.x == 0) s_ticket = (dbg & 2) ? blockIdx.x : atomicAdd(ticket_counter, 1u);
  __syncthreads();
  const uint32_t ticket = s_ticket;
  const uint32_t r0 = ticket * RPB + wv * RW;
  WaveLds<R, H> &L = lds[wv];
  H keys[RW][R];
  uint32_t nus[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const uint32_t r = r0 + i;
    nus[i] = 0;
#pragma unroll
    for (int q = 0; q < R; ++q) keys[i][q] = (H)~(H)0;
    if (r < n) {
      const uint64_t off = offsets[r];
      const uint32_t len = (uint32_t)(offsets[r + 1] - off);
      if (!(len >= (uint32_t)k && len - (uint32_t)k + 1 > 64u * R)) {  // longer sequences are the sort path's business
        wave_stage_codes<R, SB>(bases + off, len, lane, L.codes);
        wave_hash_windows<R, H, SB>(L.codes, k, content, lane, keys[i]);
        __builtin_amdgcn_wave_barrier();  // the staging area is reused by the next read
        wave_bitonic_sort<R, H>(keys[i], lane);
        nus[i] = wave_unique_count<R, H>(keys[i], (H)~(H)0, lane);
      }
    }
    if (lane == 0) s_nu[wv * RW + i] = nus[i];
  }
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (int i = 0; i < RPB; ++i) {
    if (i < wv * RW) before += s_nu[i];
    total += s_nu[i];
  }
  if (wv == 0) {
    const int naps = (dbg >> 4) & 15 ? (dbg >> 4) & 15 : 1;
    const uint64_t pre = (dbg & 1)     ? (uint64_t)ticket * RPB * 64 * R
                         : (dbg & 8) ? lookback_exclusive(state, ticket, (uint64_t)total, lane, 16)  // (8: one level, round 2's)
                                     : lookback_exclusive_grouped(state, gstate, garr, ticket, gridDim.x, (uint64_t)total, lane, naps);
    if (lane == 0) s_prefix = pre;
  }
  __syncthreads();
  uint64_t o = s_prefix + before;
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const uint32_t r = r0 + i;
    if (r < n) {
      if (nus[i]) {
        uint32_t n_valid;
        wave_unique<R, H>(keys[i], (H)~(H)0, lane, L.key, L.start, n_valid);
        for (uint32_t u = lane; u < nus[i]; u += 64) {
          out_hash[o + u] = (uint64_t)L.key[u];
          out_count[o + u] = L.start[u + 1] - L.start[u];
        }
        __builtin_amdgcn_wave_barrier();  // L.key / L.start are rewritten by the next read
      }
      if (lane == 0) {
        out_offsets[r] = o;
        if (r == n - 1) out_offsets[n] = o + nus[i];
      }
      o += nus[i];
    }
  }
}

struct LoadU32 {
  const uint32_t *p;
  __device__ uint32_t operator()(uint64_t i) const { return p[i]; }
};
struct StoreOffsets {
  uint64_t *out;
  __device__ void operator()(uint64_t i, uint64_t prefix, uint32_t) const { out[i] = prefix; }
};

// ---------------------------------------------------------------------------
// the gather: t_d = sum_u rows[col_u][d] * x_u, u ascending (lib/Twister.ml:183)
// lane owns dims lane, lane+64, ...; products and sums are NOT fused (the
// reference is OCaml: one rounding per multiply and per add).
// ---------------------------------------------------------------------------
// The LAST block of dimensions when it holds at most 32 of them (n_dims = 65, 100 - 96 ...): a pass with one load
// instruction per row would cost what a full block costs (D = 65: 1.66 ms against 1.16 at 64), so one instruction fetches
// G = 64 / P rows (P = 8, 16 or 32 lanes a row) -- and every dimension's sum stays the ONE ascending chain of the full
// blocks: the products are handed down the lanes row by row (a shuffle per row) and added in row order by every lane
// group alike.  Same multiplications, same additions, same order: the same bits as a pass of its own.
template <int U, bool NT>
__device__ __forceinline__ void wave_gather_rows_tail(const TwisterView &tv, const uint32_t *s_col, const double *s_x, uint32_t nu,
                                                      int lane, uint32_t d0, double *__restrict__ out_row) {
  const uint32_t rem = tv.n_dims - d0;  // 1..32
  const uint32_t P = rem <= 8 ? 8u : (rem <= 16 ? 16u : 32u), G = 64u / P;
  const uint32_t g = (uint32_t)lane / P, dd = (uint32_t)lane % P;
  const double *base = tv.rows + d0 + min(dd, rem - 1);  // (lanes past the last dimension: a column that exists)
  double acc = 0.0;
  constexpr int UT = 4;  // instructions in flight: G * UT rows
  for (uint32_t u0 = 0; u0 < nu; u0 += G * UT) {
    double pr[UT];
#pragma unroll
    for (int j = 0; j < UT; ++j) {
      const uint32_t u = u0 + (uint32_t)j * G + g;
      const uint32_t uu = min(u, nu);  // [nu] is zero padding: row 0 against x = 0
      const double *p = base + (uint64_t)s_col[uu] * tv.d_pad;
      const double v = NT ? __builtin_nontemporal_load(p) : *p;
      pr[j] = u < nu ? __dmul_rn(v, s_x[uu]) : 0.0;
    }
#pragma unroll
    for (int j = 0; j < UT; ++j)
      for (uint32_t gg = 0; gg < G; ++gg) {  // rows u0 + j G + gg in order
        const double t = __shfl(pr[j], (int)(gg * P + dd), 64);
        if (u0 + (uint32_t)j * G + gg < nu) acc = __dadd_rn(acc, t);  // (uniform)
      }
  }
  if (g == 0 && dd < rem) out_row[d0 + dd] = acc;
}

template <int U, bool NT>
__device__ __forceinline__ void wave_gather_rows(const TwisterView &tv, const uint32_t *s_col, const double *s_x,
                                                 uint32_t nu, int lane, double *__restrict__ out_row) {
  for (uint32_t d0 = 0; d0 < tv.n_dims; d0 += 64) {
    if (d0 > 0 && tv.n_dims - d0 <= 32) {
      wave_gather_rows_tail<U, NT>(tv, s_col, s_x, nu, lane, d0, out_row);
      break;
    }
    const uint32_t d = d0 + lane;
    const bool active = d < tv.n_dims;
    const double *base = tv.rows + d;
    double acc = 0.0;
    for (uint32_t u0 = 0; u0 < nu; u0 += U) {
      double v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const uint32_t col = s_col[u0 + j];  // padded with 0 past nu
        const double *p = base + (uint64_t)col * tv.d_pad;
        // (kept conditional: with every load unconditional -- no branch, no exec juggling, the eight loads issued together --
        // the headline launch took 1.218 ms against 1.197 on the same box: the bound is HBM, and the staggered issue suits it)
        v[j] = (active && u0 + j < nu) ? (NT ? __builtin_nontemporal_load(p) : *p) : 0.0;
      }
#pragma unroll
      for (int j = 0; j < U; ++j) acc = __dadd_rn(acc, __dmul_rn(v[j], s_x[u0 + j]));  // x padded with 0
    }
    if (active) out_row[d] = acc;
  }
}

// n_dims <= 32: a row is at most 256 bytes, so one wave instruction fetches G = 64/P rows (P = 8, 16 or 32
// lanes per row).  Group g sums the rows u = g, g+G, ... in ascending order; the G partial sums are then
// combined by a fixed xor-butterfly (same value in every lane).  The order of adds differs from the
// reference's single ascending chain by that last tree only: deterministic, equal up to rounding.
template <int U, bool NT>
__device__ __forceinline__ void wave_gather_rows_packed(const TwisterView &tv, const uint32_t *s_col, const double *s_x,
                                                        uint32_t nu, int lane, double *__restrict__ out_row) {
  const uint32_t P = (tv.n_dims <= 8) ? 8u : (tv.n_dims <= 16 ? 16u : 32u);
  const uint32_t G = 64u / P, g = (uint32_t)lane / P, d = (uint32_t)lane % P;
  const bool active = d < tv.n_dims;
  const double *base = tv.rows + d;
  double acc = 0.0;
  for (uint32_t u0 = 0; u0 < nu; u0 += G * U) {
    double v[U], x[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t u = u0 + (uint32_t)j * G + g;
      const uint32_t uu = min(u, nu);  // [nu] is zero padding
      const uint32_t col = s_col[uu];
      x[j] = s_x[uu];
      const double *p = base + (uint64_t)col * tv.d_pad;
      v[j] = (active && u < nu) ? (NT ? __builtin_nontemporal_load(p) : *p) : 0.0;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) acc = __dadd_rn(acc, __dmul_rn(v[j], x[j]));
  }
  for (uint32_t off = P; off < 64; off <<= 1) acc = __dadd_rn(acc, __shfl_xor(acc, (int)off, 64));
  if (active && g == 0) out_row[d] = acc;
}

// The same gather from a twister that keeps its rows at their hashes (twister.h, `direct`): s_col holds the k-mers' HASHES, in
// ascending order as the ranks would be, and a row that does not exist says so in its first double -- it then adds nothing, and
// the windows that counted it are reported back (the caller takes them out of the normaliser and, if there were any, comes again).
// With every k-mer present (the twisters this layout is built for) the loads, the products and the order of the additions are
// those of wave_gather_rows_packed: the same bits.
template <int U, bool NT>
__device__ __forceinline__ uint32_t wave_gather_rows_direct(const TwisterView &tv, const uint32_t *s_hash, const double *s_x, const uint32_t *s_start,
                                                            uint32_t nu, int lane, double *__restrict__ out_row) {
  const uint32_t P = (tv.n_dims <= 8) ? 8u : (tv.n_dims <= 16 ? 16u : 32u);
  const uint32_t G = 64u / P, g = (uint32_t)lane / P, d = (uint32_t)lane % P;
  const bool active = d < tv.n_dims;
  const double *base = tv.direct + d;
  double acc = 0.0;
  uint32_t gone = 0;  // windows of k-mers without a row (kept by the first lane of a group)
  for (uint32_t u0 = 0; u0 < nu; u0 += G * U) {
    double v[U], x[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t u = u0 + (uint32_t)j * G + g;
      const uint32_t uu = min(u, nu);  // [nu] is zero padding
      const uint32_t h = s_hash[uu];
      x[j] = s_x[uu];
      // (a twister that keeps a range of the k-mer rows: a hash outside it is not looked at -- marked as having no row)
      const bool mine = h >= tv.direct_lo && h < tv.direct_hi;
      const double *p = base + (uint64_t)(mine ? h - tv.direct_lo : 0u) * tv.d_pad;
      v[j] = (active && u < nu) ? (mine ? (NT ? __builtin_nontemporal_load(p) : *p) : __longlong_as_double((long long)kDirectAbsent)) : 0.0;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint64_t none = __ballot(d == 0 && (uint64_t)__double_as_longlong(v[j]) == kDirectAbsent);
      if (none) {  // (uniform; never taken for a complete twister)
        if ((none >> (g * P)) & 1ull) {
          v[j] = 0.0;
          const uint32_t u = u0 + (uint32_t)j * G + g;
          if (d == 0) gone += s_start[u + 1] - s_start[u];
        }
      }
      acc = __dadd_rn(acc, __dmul_rn(v[j], x[j]));
    }
  }
  for (uint32_t off = P; off < 64; off <<= 1) acc = __dadd_rn(acc, __shfl_xor(acc, (int)off, 64));
  if (active && g == 0) out_row[d] = acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gone += (uint32_t)__shfl_xor((int)gone, o, 64);
  return gone;
}

// x_u = count_u / acc when normalising and acc <> 0 (lib/Twister.ml:177-178)
template <int R>
__device__ __forceinline__ void wave_fill_x(WaveLds<R, uint32_t> &L, uint32_t nu, double acc, int normalize, int lane) {
  const bool norm = normalize && acc != 0.0;
  for (uint32_t u = lane; u < nu + kGatherPad; u += 64) {
    if (u < nu) {
      double c = (double)(L.start[u + 1] - L.start[u]);
      L.x[u] = norm ? c / acc : c;
    } else {
      L.x[u] = 0.0;
      L.key[u] = 0u;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------
// fused count -> twist: one wave per read
// ---------------------------------------------------------------------------
// (PACKED: `bases` is the batch's words of 2-bit codes and `pinvalid` its marks of bases that are none -- kpop_dev_count_twist_packed)
// (SB: bits a symbol -- 2 DNA, 5 protein, as count_wave_kernel's)
template <int R, typename H, int U, bool NT, bool PACKED = false, int SB = 2>
__global__ __launch_bounds__(64 * kWavesPerBlock) void count_twist_wave_kernel(
    TwisterView tv, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets,
    const uint32_t *__restrict__ read_ids, uint32_t n, int content, int normalize, double *__restrict__ out,
    const uint32_t *__restrict__ pinvalid = nullptr) {
  __shared__ WaveLds<R, uint32_t> lds[kWavesPerBlock];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t w = blockIdx.x * kWavesPerBlock + wv;
  if (w >= n) return;
  const uint32_t r = read_ids ? read_ids[w] : w;
  const uint64_t off = offsets[r];
  const uint32_t len = (uint32_t)(offsets[r + 1] - off);
  WaveLds<R, uint32_t> &L = lds[wv];
  if (len >= (uint32_t)tv.hk && len - (uint32_t)tv.hk + 1 > 64u * R) {
    // count_twist_stream_kernel's business -- if the host scheduled it (bit 1 of `normalize`).  It does so from the
    // caller's max_len; a max_len that understates the batch must not leave a row of stale memory behind: NaNs say so.
    if (!(normalize & 2))
      for (uint32_t d = lane; d < tv.n_dims; d += 64) out[(uint64_t)r * tv.n_dims + d] = __longlong_as_double(0x7FF8000000000000ll);
    return;
  }
  normalize &= 1;

  if constexpr (PACKED)
    wave_stage_codes_packed<R>(reinterpret_cast<const uint32_t *>(bases), pinvalid, off, len, lane, L.codes);
  else
    wave_stage_codes<R, SB>(bases + off, len, lane, L.codes);
  H hkey[R];
  wave_hash_windows<R, H, SB>(L.codes, tv.hk, content, lane, hkey);
  // name -> column (lib/Twister.ml:151); k-mers the twister does not know are
  // dropped here, hence also from the normaliser (:158,:167-169)
  const bool direct = tv.direct != nullptr && tv.n_dims <= 32;  // rows at their hashes: no look-up (twister.h)
  uint32_t key[R];
#pragma unroll
  for (int i = 0; i < R; ++i) key[i] = (hkey[i] != (H)~(H)0) ? (direct ? (uint32_t)hkey[i] : lookup_col(tv, (uint64_t)hkey[i])) : kNoCol;
  wave_bitonic_sort<R, uint32_t>(key, lane);
  uint32_t n_valid;
  const uint32_t nu = wave_unique<R, uint32_t>(key, kNoCol, lane, L.key, L.start, n_valid);
  // counts are integers, so acc (:158) is exact whatever the order of the adds
  wave_fill_x<R>(L, nu, (double)n_valid, normalize, lane);
  if (direct) {
    const uint32_t gone = wave_gather_rows_direct<U, NT>(tv, L.key, L.x, L.start, nu, lane, out + (uint64_t)r * tv.n_dims);
    if (gone && normalize) {  // k-mers the twister does not know are not part of the normaliser (:158,:167-169): once more without them
      __builtin_amdgcn_wave_barrier();
      wave_fill_x<R>(L, nu, (double)(n_valid - gone), normalize, lane);
      (void)wave_gather_rows_direct<U, NT>(tv, L.key, L.x, L.start, nu, lane, out + (uint64_t)r * tv.n_dims);
    }
  } else if (tv.n_dims <= 32)
    wave_gather_rows_packed<U, NT>(tv, L.key, L.x, nu, lane, out + (uint64_t)r * tv.n_dims);
  else
    wave_gather_rows<U, NT>(tv, L.key, L.x, nu, lane, out + (uint64_t)r * tv.n_dims);
}

// ---------------------------------------------------------------------------
// twist of CSR spectra (hash, value): one wave per spectrum, any length.
// Lines are taken in file order; duplicates are not merged first (the sum is
// the same up to rounding) -- see DESIGN.md "twist_csr".  acc is a wave
// reduction, exact for integer counts.
// ---------------------------------------------------------------------------
// U = row loads in flight per wave: 8 when there is a wave per SIMD slot anyway (read spectra by the hundred thousand), 32
// when a few thousand long spectra leave two waves per SIMD and the latency of each batch of loads shows (genomes).
// KEEP > 0: every spectrum has at most 64 * KEEP lines, so the columns and values found by pass 1 stay in registers (KEEP
// per lane) and pass 2 neither reads the hashes nor walks the name -> row index again -- round 2 did both twice: 1.33 x
// the algorithmic traffic on 100,000 read spectra.  Same lines in the same order, same arithmetic: identical results.
// SEG: a few very long spectra (class spectra of millions of lines, genomes by the hundred) are cut into n_seg segments of
// seg_lines lines, a wave per (spectrum, segment): acc comes from twist_csr_seg_acc_kernel's per-segment sums (added in
// segment order), the segment's sum goes to `out` as a partial row [spectrum * n_seg + segment] and
// twist_csr_seg_combine_kernel adds the partial rows in segment order.  One wave per spectrum left 200 spectra of 300,000
// lines on a fifth of the chip (2.4 G lines/s against 15).
// NDB: blocks of 64 dimensions a lane sums per pass over the lines (1 to 4): with more than 64 dimensions the lines are
// looked up and staged ONCE per 64 NDB dimensions (every dimension's sum is the same chain either way: the same bits).
template <typename V, int U = kGatherUnroll, int KEEP = 0, bool SEG = false, int NDB = 1>  // V double: spectra as parsed from text; uint32_t: counts straight from the counting kernels
__global__ __launch_bounds__(64 * kWavesPerBlock) void twist_csr_kernel(
    TwisterView tv, const uint64_t *__restrict__ hash, const V *__restrict__ value,
    const uint64_t *__restrict__ offsets, uint32_t n, int normalize, double *__restrict__ out, uint32_t n_seg = 1,
    uint64_t seg_lines = 0, const double *__restrict__ acc_part = nullptr) {
  __shared__ uint32_t s_col[kWavesPerBlock][64 + kGatherUnroll];
  __shared__ double s_x[kWavesPerBlock][64 + kGatherUnroll];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t s = blockIdx.x * kWavesPerBlock + wv;
  if (s >= n * (SEG ? n_seg : 1u)) return;
  const uint32_t spec = SEG ? s / n_seg : s;
  uint64_t lo = offsets[spec], hi = offsets[spec + 1];
  if (SEG) {
    if (s % n_seg == n_seg - 1 && hi - lo > (uint64_t)n_seg * seg_lines) {
      // the caller's max_lines understated this spectrum and its last stretches have no segment: a row of NaNs in the last
      // segment's partial (the combine adds it in) says so -- never a silently shortened sum
      for (uint32_t d = lane; d < tv.n_dims; d += 64) out[(uint64_t)s * tv.n_dims + d] = __longlong_as_double(0x7FF8000000000000ll);
      return;
    }
    lo = min(hi, lo + (uint64_t)(s % n_seg) * seg_lines);
    hi = min(hi, lo + seg_lines);
  }
  if (KEEP > 0 && hi - lo > 64ull * KEEP) {
    // the caller's max_lines understated this spectrum: a row of NaNs says so (never a silently shortened sum)
    for (uint32_t d = lane; d < tv.n_dims; d += 64) out[(uint64_t)s * tv.n_dims + d] = __longlong_as_double(0x7FF8000000000000ll);
    return;
  }
  // pass 1: acc over lines whose k-mer the twister knows
  constexpr int NK = KEEP > 0 ? KEEP : 1;
  uint32_t kcol[NK];
  double kval[NK];
  double part = 0.0;
  if (KEEP > 0) {
#pragma unroll
    for (int q = 0; q < NK; ++q) {
      const uint64_t i = lo + (uint64_t)q * 64 + lane;
      kcol[q] = kNoCol;
      kval[q] = 0.0;
      if (i < hi) {
        kcol[q] = lookup_col(tv, hash[i]);
        kval[q] = (double)value[i];
        if (kcol[q] != kNoCol) part += kval[q];
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  double acc = part;
  // A spectrum's sums -- acc here, the twisted row below -- are formed per STRETCH of kCsrSegLines lines and the stretches
  // added in order, in this kernel (one wave walks the stretches) as in its SEG form (a wave per stretch, added up by
  // twist_csr_seg_combine_kernel): the same additions in the same order, so the same bits whichever launch a caller's
  // batch gets.  Spectra of up to kCsrSegLines lines are one stretch: nothing changes for them.
  if (KEEP == 0 && !SEG) {
    bool first = true;
    for (uint64_t s0 = lo; s0 < hi; s0 += kCsrSegLines) {
      const uint64_t s1 = min(hi, s0 + kCsrSegLines);
      double ps = 0.0;
      for (uint64_t i = s0 + lane; i < s1; i += 64)
        if (lookup_col(tv, hash[i]) != kNoCol) ps += (double)value[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ps += __shfl_xor(ps, o, 64);
      acc = first ? ps : __dadd_rn(acc, ps);
      first = false;
    }
  }
  if (SEG) {  // the whole spectrum's acc: its stretches' sums in order
    acc = acc_part[(uint64_t)spec * n_seg];
    for (uint32_t g = 1; g < n_seg; ++g) acc = __dadd_rn(acc, acc_part[(uint64_t)spec * n_seg + g]);
  }
  const bool norm = normalize && acc != 0.0;
  // pass 2: 64 lines at a time through LDS, then the shared gather
  const uint32_t n_dims = tv.n_dims;
  constexpr int UE = (U >= 16 && NDB > 1) ? (U / NDB < 8 ? 8 : U / NDB) : U;  // rows in flight (x NDB loads each)
  for (uint32_t d0 = 0; d0 < n_dims; d0 += 64 * NDB) {
    const double *base[NDB];
    bool active[NDB];
    double t[NDB], t_done[NDB];  // t: the current stretch; t_done: the stretches before it
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      const uint32_t d = d0 + 64u * b + lane;
      active[b] = d < n_dims;
      base[b] = tv.rows + (active[b] ? d : n_dims - 1);  // (lanes past the last dimension: a column that exists)
      t[b] = 0.0;
      t_done[b] = 0.0;
    }
    bool any_done = false;
    int q = 0;
    for (uint64_t i0 = lo; i0 < hi; i0 += 64, ++q) {
      if (KEEP == 0 && !SEG && i0 != lo && (i0 - lo) % kCsrSegLines == 0) {  // (uniform) a stretch ends
#pragma unroll
        for (int b = 0; b < NDB; ++b) {
          t_done[b] = any_done ? __dadd_rn(t_done[b], t[b]) : t[b];
          t[b] = 0.0;
        }
        any_done = true;
      }
      const uint64_t i = i0 + lane;
      uint32_t col = kNoCol;
      double x = 0.0;
      if (KEEP > 0) {
        // (a constant-index walk of the register arrays: q is uniform over the wave)
#pragma unroll
        for (int qq = 0; qq < NK; ++qq)
          if (qq == q) {
            col = kcol[qq];
            x = norm ? kval[qq] / acc : kval[qq];
          }
      } else if (i < hi) {
        col = lookup_col(tv, hash[i]);
        x = norm ? (double)value[i] / acc : (double)value[i];
      }
      if (col == kNoCol) {
        col = 0;
        x = 0.0;
      }
      __builtin_amdgcn_wave_barrier();
      s_col[wv][lane] = col;
      s_x[wv][lane] = x;
      __builtin_amdgcn_wave_barrier();
      const uint32_t cnt = (uint32_t)min((uint64_t)64, hi - i0);
      for (uint32_t u0 = 0; u0 < cnt; u0 += UE) {
        double v[UE][NDB];
#pragma unroll
        for (int j = 0; j < UE; ++j) {
          const uint32_t uu = min(u0 + j, 63u);
          const uint32_t c = s_col[wv][uu];
          const double xx = s_x[wv][uu];
          // Few long spectra (U = 32: genomes, two wavefronts a SIMD): every load unconditional -- a line without a row
          // brings row 0 in against x = 0, an exact zero added -- where the conditional form compiles to a branch and an exec
          // save / restore per line: 5,000 x 30 kb spectra 4.72 -> 3.19 ms.  Read spectra by the hundred thousand (U = 8)
          // are HBM-bound and 2.5 % FASTER with the conditional form's staggered issue (1.31 against 1.34 ms): it stays.
#pragma unroll
          for (int b = 0; b < NDB; ++b) {
            if constexpr (U >= 16) v[j][b] = base[b][(uint64_t)c * tv.d_pad];
            else v[j][b] = (active[b] && u0 + j < cnt && xx != 0.0) ? base[b][(uint64_t)c * tv.d_pad] : 0.0;
          }
        }
#pragma unroll
        for (int j = 0; j < UE; ++j) {
          const uint32_t uu = min(u0 + j, 63u);
          const double xx = (u0 + j < cnt) ? s_x[wv][uu] : 0.0;
#pragma unroll
          for (int b = 0; b < NDB; ++b) t[b] = __dadd_rn(t[b], __dmul_rn(v[j][b], xx));
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      if (any_done) t[b] = __dadd_rn(t_done[b], t[b]);
      if (active[b]) out[(uint64_t)s * n_dims + d0 + 64u * b + lane] = t[b];
    }
  }
}

// acc_part[spectrum * n_seg + segment] = the values of the segment's lines the twister knows, added up (a wave per pair)
template <typename V>
__global__ __launch_bounds__(64 * kWavesPerBlock) void twist_csr_seg_acc_kernel(TwisterView tv, const uint64_t *__restrict__ hash,
                                                                                const V *__restrict__ value, const uint64_t *__restrict__ offsets,
                                                                                uint32_t n, uint32_t n_seg, uint64_t seg_lines,
                                                                                double *__restrict__ acc_part) {
  const int lane = threadIdx.x & 63;
  const uint32_t s = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (s >= n * n_seg) return;
  const uint32_t spec = s / n_seg;
  uint64_t lo = offsets[spec], hi = offsets[spec + 1];
  lo = min(hi, lo + (uint64_t)(s % n_seg) * seg_lines);
  hi = min(hi, lo + seg_lines);
  double part = 0.0;
  for (uint64_t i = lo + lane; i < hi; i += 64)
    if (lookup_col(tv, hash[i]) != kNoCol) part += (double)value[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) acc_part[s] = part;
}

// out[spectrum] = its segments' partial rows added in segment order
__global__ __launch_bounds__(256) void twist_csr_seg_combine_kernel(const double *__restrict__ partial, uint32_t n_seg, uint32_t n_dims,
                                                                    double *__restrict__ out) {
  const uint32_t spec = blockIdx.x;
  for (uint32_t d = threadIdx.x; d < n_dims; d += blockDim.x) {
    double t = partial[(uint64_t)spec * n_seg * n_dims + d];
    for (uint32_t g = 1; g < n_seg; ++g) t = __dadd_rn(t, partial[((uint64_t)spec * n_seg + g) * n_dims + d]);
    out[(uint64_t)spec * n_dims + d] = t;
  }
}

// launch by the longest spectrum of the batch: <= 512 lines keeps pass 1's columns in registers
template <typename V>
static int launch_twist_csr(const TwisterView &tv, const uint64_t *hash, const V *value, const uint64_t *offsets, uint32_t n,
                            uint64_t max_lines, int normalize, double *out, hipStream_t st) {
  const dim3 grid(div_up(n, kWavesPerBlock)), block(64 * kWavesPerBlock);
  const bool few = n <= kFewSpectra;
  // A few very long spectra: segments, so that the whole chip works on them (see the kernel).  An empty segment adds +0.0.
  if (max_lines >= 16384 && n <= 2048 && !(ctx().tune_dbg & (1 << 28))) {
    const uint32_t n_seg = (uint32_t)div_up(max_lines, kCsrSegLines);  // (stretches of a fixed length: see the kernel)
    if (n_seg >= 2 && (uint64_t)n * n_seg <= (1u << 22)) {
      const uint64_t seg_lines = kCsrSegLines;
      const uint64_t bytes_acc = ((uint64_t)n * n_seg * 8 + 255) & ~255ull;
      void *ws = nullptr;
      KPOP_TRY(ctx().ws_for(st).ensure(bytes_acc + (uint64_t)n * n_seg * tv.n_dims * 8, &ws));
      double *acc_part = reinterpret_cast<double *>(ws);
      double *partial = reinterpret_cast<double *>(reinterpret_cast<char *>(ws) + bytes_acc);
      const dim3 sgrid(div_up(n * n_seg, kWavesPerBlock));
      twist_csr_seg_acc_kernel<V><<<sgrid, block, 0, st>>>(tv, hash, value, offsets, n, n_seg, seg_lines, acc_part);
      KPOP_LAUNCH_CHECK();
      if (tv.n_dims <= 64) twist_csr_kernel<V, 32, 0, true, 1><<<sgrid, block, 0, st>>>(tv, hash, value, offsets, n, normalize, partial, n_seg, seg_lines, acc_part);
      else if (tv.n_dims <= 128) twist_csr_kernel<V, 32, 0, true, 2><<<sgrid, block, 0, st>>>(tv, hash, value, offsets, n, normalize, partial, n_seg, seg_lines, acc_part);
      else if (tv.n_dims <= 192) twist_csr_kernel<V, 32, 0, true, 3><<<sgrid, block, 0, st>>>(tv, hash, value, offsets, n, normalize, partial, n_seg, seg_lines, acc_part);
      else twist_csr_kernel<V, 32, 0, true, 4><<<sgrid, block, 0, st>>>(tv, hash, value, offsets, n, normalize, partial, n_seg, seg_lines, acc_part);
      KPOP_LAUNCH_CHECK();
      twist_csr_seg_combine_kernel<<<dim3(n), dim3(256), 0, st>>>(partial, n_seg, tv.n_dims, out);
      KPOP_LAUNCH_CHECK();
      return 0;
    }
  }
  // (UU loads in flight, KK lines kept in registers; the blocks of 64 dimensions a lane sums per pass by the twister's width)
#define KPOP_CSR(UU, KK)                                                                                                          \
  do {                                                                                                                            \
    if (tv.n_dims <= 64) twist_csr_kernel<V, UU, KK, false, 1><<<grid, block, 0, st>>>(tv, hash, value, offsets, n, normalize, out);        \
    else if (tv.n_dims <= 128) twist_csr_kernel<V, UU, KK, false, 2><<<grid, block, 0, st>>>(tv, hash, value, offsets, n, normalize, out);  \
    else if (tv.n_dims <= 192) twist_csr_kernel<V, UU, KK, false, 3><<<grid, block, 0, st>>>(tv, hash, value, offsets, n, normalize, out);  \
    else twist_csr_kernel<V, UU, KK, false, 4><<<grid, block, 0, st>>>(tv, hash, value, offsets, n, normalize, out);                        \
  } while (0)
  if (max_lines == 0 || max_lines > 512) {
    if (few) KPOP_CSR(32, 0); else KPOP_CSR(kGatherUnroll, 0);
  } else if (max_lines <= 64) {
    if (few) KPOP_CSR(32, 1); else KPOP_CSR(kGatherUnroll, 1);
  } else if (max_lines <= 128) {
    if (few) KPOP_CSR(32, 2); else KPOP_CSR(kGatherUnroll, 2);
  } else if (max_lines <= 256) {
    if (few) KPOP_CSR(32, 4); else KPOP_CSR(kGatherUnroll, 4);
  } else {
    if (few) KPOP_CSR(32, 8); else KPOP_CSR(kGatherUnroll, 8);
  }
#undef KPOP_CSR
  KPOP_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// long sequences (assembled genomes: 30 kb .. Mb): streaming fused count->twist.
// A sequence is cut into segments of kSegWindows windows, one 256-thread block
// per segment.  Each wave takes 64 windows at a time: every lane hashes one
// window and looks its column up, then the wave walks the 64 columns with
// v_readlane (wave-uniform row address) and all lanes load that row -- lane d
// owns dimension d exactly as in the per-read kernel.  No sort: the sum runs
// over k-mer INSTANCES in sequence order and is divided by acc once at the end,
//   t_d = (sum_windows T[col(w)][d]) / acc
// which equals the reference's sum_h T[h][d]*(c_h/acc) (lib/Twister.ml:177-183)
// up to rounding.  Segment partials are combined in segment order by a second
// kernel, so results are bitwise reproducible.
// ---------------------------------------------------------------------------
constexpr uint32_t kWaveMaxWindows = 512;  // 64 lanes x R=8: the per-read kernels' limit
constexpr uint32_t kSegWindows = 16384;    // upper bound of a segment
// Row-load policy, kpop_tune("nt", 2) = automatic (measured: profiles/r02_b_realistic_inputs.txt).
// Non-temporal loads keep once-read rows from displacing the name -> row index, which is worth 0-6 % when reads are
// uniformly random over a table many times the 256 MB Infinity Cache -- and costs 6-35 % whenever rows ARE re-read:
// tables up to ~1 GB even under uniform access (the cache holds a useful share of them), reads drawn from a few
// genomes, assemblies of one organism.  So: reads kernel non-temporal only above 2 GiB of rows; genome kernel never
// (a batch of assemblies is one species more often than not, and there plain loads are 1.5x faster).
constexpr uint64_t kStreamingRowBytes = 2ull << 30;

// nseg[r] = the segments of sequence r (0: the one-wavefront-per-read kernel's), and the list of the sequences that have
// any (wave-aggregated append: the order of the list is not the batch's, and nothing depends on it).  The streaming kernel
// and the combine walk the LIST: one genome among 100,000 reads cost 0.66 ms of looking at pairs that had no segment.
// (lpos, grel, segw: with the tile route on, a sequence of a group of ONE organism -- tile_group_probe_kernel -- is cut into the
// tile kernel's stretches of 512 windows, any other into seg_windows as ever; segw[r] = which)
__global__ void segment_count_kernel(const uint64_t *__restrict__ offsets, uint32_t n, int k, uint32_t seg_windows,
                                     uint32_t *__restrict__ nseg, uint32_t *__restrict__ long_ids, uint32_t *__restrict__ n_long,
                                     const uint32_t *__restrict__ lpos, const uint32_t *__restrict__ grel, uint32_t *__restrict__ segw) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t ns = 0;
  if (r < n) {
    const uint64_t len = offsets[r + 1] - offsets[r];
    const uint64_t w = (len >= (uint64_t)k) ? len - k + 1 : 0;
    uint32_t sw = seg_windows;
    if (lpos && w > kWaveMaxWindows && grel[lpos[r] / 64u]) sw = 512u;
    ns = (w > kWaveMaxWindows) ? (uint32_t)((w + sw - 1) / sw) : 0u;
    nseg[r] = ns;
    if (segw) segw[r] = sw;
  }
  const uint64_t m = __ballot(ns != 0);
  if (m) {
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(n_long, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (ns) long_ids[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = r;
  }
}

struct StoreU64 {
  uint64_t *out;
  __device__ void operator()(uint64_t i, uint64_t prefix, uint32_t) const { out[i] = prefix; }
};

// NDB: blocks of 64 dimensions a lane sums per pass over the windows (1, 2 or 4): with more than 64 dimensions the windows
// are hashed and looked up ONCE per 64 NDB dimensions, not once per 64 (every dimension's sum is the same chain either way).
template <typename H, bool NT, int NDB = 1, int SB = 2>
__global__ __launch_bounds__(256) void count_twist_stream_kernel(
    TwisterView tv, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets, int content,
    const uint32_t *__restrict__ nseg, const uint64_t *__restrict__ seg_off, double *__restrict__ partial,
    uint32_t *__restrict__ partial_cnt, const uint32_t *__restrict__ long_ids, const uint32_t *__restrict__ n_long_ptr, uint32_t max_seg,
    uint32_t seg_windows_all, const uint2 *__restrict__ todo, const uint32_t *__restrict__ segw, uint32_t *__restrict__ next_pair,
    uint32_t d_begin, uint32_t d_end) {
  __shared__ double s_part[4][64 * NDB];
  __shared__ uint32_t s_cnt[4], s_pair;
  // todo: the (sequence, segment) pairs count_twist_tile_kernel left (tile_todo_kernel's list; n_long_ptr then counts those)
  const uint32_t n_reads = *n_long_ptr;  // the sequences that have segments (long_ids), not the batch
  if (n_reads == 0) return;
  // (segment, read) pairs are dealt to blocks round-robin, READS FASTEST: the blocks in flight at any moment work on the
  // same stretch of many sequences.  Assemblies of one organism are near-identical (BASELINE config 3), so those blocks
  // gather the same twister rows -- a segment's rows (seg_windows x d_pad x 8 B, sized to sit in an XCD's 4 MB L2) come
  // from HBM once per XCD and are L2 hits for every other sequence.  Unrelated sequences lose nothing by this order.
  // A ragged batch (one genome among a million reads) has far more pairs than HIP allows blocks, and almost all of them
  // are empty, hence the grid-stride loop.
  // With a to-do list every pair has work and the blocks are as many as the chip holds at once: each takes the next pair
  // off a counter when it is done with one (dealt round-robin, 50,000 pairs over 2,048 blocks are 25 for some and 24 for
  // others: the launch took 25 pairs' time, 12.99 against 12.60 ms).
  const uint64_t n_pairs = todo ? (uint64_t)n_reads : (uint64_t)n_reads * max_seg;
  for (uint64_t pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
  uint32_t seg, r;
  if (todo) {
    if (threadIdx.x == 0) s_pair = atomicAdd(next_pair, 1u);
    __syncthreads();
    const uint32_t mine = s_pair;
    if (mine >= n_pairs) break;  // (uniform)
    pair = blockIdx.x;           // (the loop's own counter plays no part)
    const uint2 t = todo[mine];
    r = t.x;
    seg = t.y;
  } else {
    seg = (uint32_t)(pair / n_reads);
    r = long_ids[pair % n_reads];
    if (seg >= nseg[r]) continue;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t off = offsets[r];
  const uint64_t len = offsets[r + 1] - off;
  const int k = tv.hk;
  const uint64_t n_win = len - k + 1;  // nseg > 0 implies len >= k
  const uint32_t seg_windows = segw ? segw[r] : seg_windows_all;
  const uint64_t w0 = (uint64_t)seg * seg_windows;
  const uint64_t w1 = min(n_win, w0 + seg_windows);
  const uint8_t *seq = bases + off;
  const int shift = 2 * (k - 1);  // (DNA: where a base enters the reverse complement)
  const uint64_t slot = seg_off[r] + seg;
  for (uint32_t d0 = d_begin; d0 < d_end; d0 += 64 * NDB) {  // (the launch's share of the dimensions: see the host side)
    // (lanes past the last dimension load a column that exists and keep a sum nobody reads; a window without a row loads
    // row 0 and adds 0.0: no branch and no exec juggling per window -- the compiler's version of the conditional load was
    // thirteen instructions and a branch a window, 151 registers, three wavefronts a SIMD; 5,000 x 30 kb on one box:
    // 6.39 -> 5.74 ms for mutants of one genome, 13.09 -> 12.66 ms for unrelated ones)
    const double *base[NDB];
    double acc[NDB];
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      const uint32_t d = d0 + 64u * b + lane;
      base[b] = tv.rows + (d < tv.n_dims ? d : tv.n_dims - 1);
      acc[b] = 0.0;
    }
    uint32_t cnt = 0;
    for (uint64_t cb = w0 + (uint64_t)wv * 64; cb < w1; cb += 4 * 64) {
      const uint64_t win = cb + lane;
      uint32_t col = kNoCol;
      if (win < w1) {
        H fwd = 0, rc = 0;
        bool good = true;
        if constexpr (SB == 2) {
          for (int j = 0; j < k; ++j) {
            const uint32_t c = base_code(seq[win + j]);
            good = good && (c < 4u);
            fwd = (fwd << 2) | (H)(c & 3u);
            rc = (rc >> 2) | ((H)(3u - (c & 3u)) << shift);
          }
        } else {
          for (int j = 0; j < k; ++j) {
            const uint32_t c = protein_code(seq[win + j]);
            good = good && (c < 20u);
            fwd = (fwd << 5) | (H)(c & 31u);
          }
          rc = fwd;
        }
        if (good) col = lookup_col(tv, (uint64_t)((SB == 2 && content == KPOP_DNA_DS && rc < fwd) ? rc : fwd));
      }
      cnt += (uint32_t)__popcll(__ballot(col != kNoCol));
      constexpr int GU = kGatherUnroll / (NDB > 2 ? 2 : 1);  // rows in flight (x NDB loads each)
#pragma unroll
      for (int j0 = 0; j0 < 64; j0 += GU) {
        double v[GU][NDB];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
          const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)col, j0 + u);
          const uint32_t cs = cj != kNoCol ? cj : 0u;  // (scalar)
#pragma unroll
          for (int b = 0; b < NDB; ++b) {
            const double x = NT ? __builtin_nontemporal_load(base[b] + (uint64_t)cs * tv.d_pad) : base[b][(uint64_t)cs * tv.d_pad];
            v[u][b] = cj != kNoCol ? x : 0.0;
          }
        }
#pragma unroll
        for (int u = 0; u < GU; ++u)
#pragma unroll
          for (int b = 0; b < NDB; ++b) acc[b] = __dadd_rn(acc[b], v[u][b]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NDB; ++b) s_part[wv][64 * b + lane] = acc[b];
    if (lane == 0) s_cnt[wv] = cnt;
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int b = 0; b < NDB; ++b) {
        const uint32_t d = d0 + 64u * b + lane;
        const double t = __dadd_rn(__dadd_rn(__dadd_rn(s_part[0][64 * b + lane], s_part[1][64 * b + lane]), s_part[2][64 * b + lane]), s_part[3][64 * b + lane]);
        if (d < tv.n_dims) partial[slot * tv.n_dims + d] = t;
      }
      if (d0 == 0 && lane == 0) partial_cnt[slot] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
    __syncthreads();  // (s_part is the next pass's)
  }
  __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Assemblies of ONE organism (BASELINE config 3's kind of batch): CONSENSUS on the matrix cores + a RESIDUAL gather.
// The same stretch of 512 windows of 64 consecutive sequences holds little more than 512 DISTINCT k-mers, so the 64 x 512
// row gathers of the streaming kernel (33 MB through L2 per chunk) are 64 times the rows that differ.  A block instead
//   1. builds the chunk's CONSENSUS: the distinct twister rows of four SEED sequences (0, 16, 32, 48 of the group) in an LDS
//      hash set -- seed 0 first, whole (at most 512 rows), the other three up to kTileSetCap rows in all,
//   2. numbers them; every sequence looks its windows up in the set: a hit counts into an LDS matrix X[64][U] of u16, a miss
//      (a k-mer around a substitution: private to the sequence, 12 of them per substitution) goes on the sequence's
//      RESIDUAL list in HBM, in window order,
//   3. multiplies: partial[64 x D] = X[64 x U] * T_U[U x D] on the f64 matrix cores, the U rows of the twister gathered
//      ONCE per chunk,
// and tile_residual_kernel adds every sequence's listed rows to its partial (one wavefront per (sequence, segment), the
// gather of the streaming kernel without its hashing: those rows are the ones HBM has to deliver whatever the scheme).
// Nothing overflows on divergence: a chunk whose sequences share little with the seeds (unrelated genomes; found from every
// thread's first two windows, a sixteenth of the chunk) is left to the streaming kernel as a whole (slot_done stays 0), and
// a block that meets four of those in a row skips ahead.  The sums run over the set's rows in set order, then the residual
// rows in window order: equal to the streaming kernel's up to rounding (<= 2e-15 relative measured), like the rest of the
// "dense" routes.  lib/Twister.ml:146-188.
// ---------------------------------------------------------------------------
using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr uint32_t kTileProbeG = 64, kTileS = 512, kTileH = 2048;  // sequences of a probed group; windows of a stretch; slots of the set's table
constexpr uint32_t tile_set_rows(int g) { return g == 64 ? 1024u : 896u; }  // rows the numbering has room for (X rows are padded by 2: 16 rows on 16 banks)
constexpr uint32_t kTileSetCap = 768;    // rows in the set beyond which the seeds after the first add no more
constexpr uint32_t kTileStageW = 136;    // dwords of a sequence's staged stretch: 3 + 512 + 14 bytes and the thirteenth dword of the last thread
constexpr uint32_t kTileMinSeqs = 16;    // sequences with segments in a batch below which the tile kernel does not try

// the sequences of more than kWaveMaxWindows windows, in batch order (an exclusive scan's compaction), each one's place in that list
struct LoadLongW {
  const uint64_t *offsets;
  int k;
  __device__ uint32_t operator()(uint64_t i) const {
    const uint64_t len = offsets[i + 1] - offsets[i];
    return (len >= (uint64_t)k && len - k + 1 > kWaveMaxWindows) ? 1u : 0u;
  }
};
struct StoreLong {
  uint32_t *olong, *lpos;
  __device__ void operator()(uint64_t i, uint64_t prefix, uint32_t v) const {
    if (v) {
      olong[prefix] = (uint32_t)i;
      lpos[i] = (uint32_t)prefix;
    }
  }
};

// Is a group of 64 of them ONE organism?  Asked before anything is laid out, at three stretches (a quarter, a half, three
// quarters of the way along the group's longest sequence): the k-mers of the four seeds' 512 windows go into an LDS set (hashes:
// no twister needed), eight windows of each of the 64 sequences are looked up, and half of them or more found at any of the three
// stretches says yes.  A group that says no is never touched by the tile kernel and keeps the streaming kernel's own (longer)
// segments: a batch of unrelated genomes pays this probe and nothing else (it paid 8-10 % for the tile kernel's attempts and for
// segments of 512 windows before).  One block per group; also the group's longest sequence in stretches of 512 windows (gmax).
__global__ __launch_bounds__(256) void tile_group_probe_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets, int k, int content,
                                                               const uint32_t *__restrict__ olong, const uint64_t *__restrict__ n_long_ptr,
                                                               uint32_t *__restrict__ gmax, uint32_t *__restrict__ grel) {
  constexpr uint32_t kTab = 4096;
  __shared__ uint32_t tab[kTab];
  __shared__ uint32_t s_maxw, s_stat;
  const uint32_t n_long = (uint32_t)*n_long_ptr, g = blockIdx.x;
  if ((uint64_t)g * kTileProbeG >= n_long) return;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    const uint32_t li = g * kTileProbeG + threadIdx.x;
    uint32_t w = 0;
    if (li < n_long) {
      const uint32_t r = olong[li];
      const uint64_t len = offsets[r + 1] - offsets[r];
      w = (uint32_t)min<uint64_t>(len - k + 1, 0xFFFFFFFFull);  // (of the list: more than 512 windows)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w = max(w, (uint32_t)__shfl_xor((int)w, o, 64));
    if (lane == 0) s_maxw = w;
  }
  __syncthreads();
  const uint32_t ns = (s_maxw + kTileS - 1) / kTileS;
  const int shift = 2 * (k - 1);
  const uint32_t mask = (uint32_t)bits_mask(2 * k);
  // eight consecutive windows of sequence `li` of the list from window w on: their hashes, kNoCol where there is none
  auto hashes = [&](uint32_t li, uint64_t w, uint32_t (&h)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = kNoCol;
    if (li >= n_long) return;
    const uint32_t r = olong[li];
    const uint64_t off = offsets[r], len = offsets[r + 1] - off;
    const uint64_t n_win = len - k + 1;
    if (w >= n_win) return;
    const uint8_t *seq = bases + off + w;
    const uint32_t nb = (uint32_t)min<uint64_t>(n_win - w, 8) + (uint32_t)k - 1;  // bases to read
    uint32_t fwd = 0, rc = 0;
    int run = 0;
    for (uint32_t j = 0; j < nb; ++j) {
      const uint32_t c = base_code(seq[j]);
      fwd = ((fwd << 2) | (c & 3u)) & mask;
      rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
      run = c < 4u ? run + 1 : 0;
      if (j + 1 >= (uint32_t)k && run >= k) {
        const uint32_t v = (content == KPOP_DNA_DS && rc < fwd) ? rc : fwd;
        const uint32_t i = j + 1 - (uint32_t)k;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if ((uint32_t)q == i) h[q] = v;
      }
    }
  };
  uint32_t rel = 0;
  for (uint32_t sample = 1; sample <= 3 && !rel; ++sample) {  // (rel is uniform)
    const uint64_t s_beg = (uint64_t)(ns * sample / 4) * kTileS;
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < kTab; q += 256) tab[q] = kNoCol;
    if (threadIdx.x == 0) s_stat = 0;
    __syncthreads();
    uint32_t h[8];
    hashes(g * kTileProbeG + (threadIdx.x >> 6) * 16u, s_beg + (uint64_t)(threadIdx.x & 63) * 8, h);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (h[i] != kNoCol) {
        uint32_t slot = (h[i] * 2654435761u) >> 20;  // 12 bits
        for (uint32_t t = 0; t < kTab; ++t) {        // (at most 2,048 of 4,096 slots are ever taken)
          const uint32_t prev = atomicCAS(&tab[slot], kNoCol, h[i]);
          if (prev == kNoCol || prev == h[i]) break;
          slot = (slot + 1) & (kTab - 1);
        }
      }
    __syncthreads();
    hashes(g * kTileProbeG + (threadIdx.x >> 2), s_beg + (uint64_t)(threadIdx.x & 3) * 128, h);
    uint32_t st = 0;  // windows | hits << 16
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (h[i] != kNoCol) {
        uint32_t slot = (h[i] * 2654435761u) >> 20;
        uint32_t key = tab[slot];
        for (uint32_t t = 0; t < kTab && key != h[i] && key != kNoCol; ++t) {
          slot = (slot + 1) & (kTab - 1);
          key = tab[slot];
        }
        st += 1u + (key == h[i] ? 65536u : 0u);
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) st += (uint32_t)__shfl_xor((int)st, o, 64);
    if (lane == 0 && st) atomicAdd(&s_stat, st);
    __syncthreads();
    const uint32_t tot = s_stat;
    rel = ((tot & 0xFFFFu) > 0 && (tot >> 16) * 2u >= (tot & 0xFFFFu)) ? 1u : 0u;
  }
  if (threadIdx.x == 0) {
    gmax[g] = ns;
    // (fewer than kTileMinSeqs long sequences: the tile kernels do not try, and segment_count_kernel must not cut them into
    // stretches of 512 windows for nothing -- 10 related genomes among 100k reads kept the streaming kernel's own segments)
    grel[g] = n_long < kTileMinSeqs ? 0u : rel;
  }
}

// phase clocks of count_twist_tile_kernel (kpop_tune("dbg", 16 << 24) only: s_memtime ticks of every block's thread 0, summed
// over blocks and chunks; read and cleared by kpop_debug_counters)
__device__ unsigned long long g_tile_stamps[16];

}  // namespace kpop
#include "tile_pipe.h"  // count_twist_tile_pipe_kernel: the same scheme as a producer / consumer pipeline (up to 64 dimensions)
namespace kpop {

// G sequences a chunk: 64 (one block of 1,024 threads a CU, 152 KB of LDS) or 32 (512 threads, 78 KB: TWO blocks a CU, one in
// its matrix phase while the other waits for bases and index words)
template <typename H, int G>
__global__ __launch_bounds__(16 * G) void count_twist_tile_kernel(
    TwisterView tv, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets, int content,
    const uint32_t *__restrict__ nseg, const uint64_t *__restrict__ seg_off, double *__restrict__ partial,
    uint32_t *__restrict__ partial_cnt, const uint32_t *__restrict__ olong, const uint64_t *__restrict__ n_long_ptr,
    const uint32_t *__restrict__ gmax, const uint32_t *__restrict__ grel, uint32_t max_seg, uint32_t *__restrict__ slot_done,
    uint32_t *__restrict__ res_rows, uint32_t *__restrict__ wave_lists, int dbg) {
  constexpr uint32_t THREADS = 16 * G, WAVES = G / 4, TU = tile_set_rows(G), XS = TU + 2, SEEDEVERY = G / 4, MT = G / 16, KQ = WAVES / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_lds[];
  uint2 *ht = reinterpret_cast<uint2 *>(tile_lds);                  // [kTileH] {twister row (kNoCol = empty), its number in the set}
  uint32_t *ucol = reinterpret_cast<uint32_t *>(ht + kTileH);       // [TU] row of number u
  uint32_t *Xw = ucol + TU;                                     // [G][XS / 2] pairs of u16 counts
  uint32_t *stage = Xw;                                             // before X is needed: [G][kTileStageW] the stretch's bases,
  uint32_t *seedcols = Xw + G * kTileStageW;                   //   and [4][kTileS] the seeds' rows
  __shared__ uint32_t s_n, s_new, s_samp, s_add[G / SEEDEVERY], s_wbase[WAVES];
  __shared__ uint64_t s_slot[G];  // the group's (sequence, segment) slots, ~0: the sequence has no such segment
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // groups of 64 are cut from the sequences that HAVE segments, in batch order (olong: the scan's list, the same every run): a
  // few assemblies among a million reads make a few groups, not sixteen thousand empty ones
  const uint32_t n_long = (uint32_t)*n_long_ptr;
  if (n_long < kTileMinSeqs) return;  // (too few sequences to share anything: the streaming kernel's)
  const uint32_t n_groups = (n_long + G - 1) / G;
  const uint64_t n_chunks = (uint64_t)n_groups * max_seg;
  const int k = tv.hk;
  const int shift = 2 * (k - 1);
  const H mask = (H)bits_mask(2 * k);
  // thread t works on sequence t / 16 of the group and a sixteenth of the segment's windows (32), rolling the hash along
  constexpr uint32_t kPer = kTileS / 16;
  const uint32_t tg = threadIdx.x >> 4, tq = threadIdx.x & 15u;
  const bool seed = (tg % SEEDEVERY) == 0;
  // chunks in a row that shared too little with their seeds: after four the block skips the next `backoff` of its chunks
  // (they are the streaming kernel's), twice as many every time until a chunk is taken again
  int misses = 0;
  uint32_t skip = 0, backoff = 8;
  auto missed = [&]() {
    if (++misses >= 4) {
      misses = 0;
      skip = backoff;
      backoff = min(backoff * 2u, 1u << 20);
    }
  };
  unsigned long long t_last = 0;
  const bool stamps = (dbg & 16) && threadIdx.x == 0;
  auto stamp = [&](int phase) {
    if (stamps) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      atomicAdd(&g_tile_stamps[phase], now - t_last);
      t_last = now;
    }
  };
  auto set_slot = [](uint32_t col) { return (col * 2654435761u) >> 21; };  // 11 bits
  // ORDERED linear probing: where two rows meet the smaller one keeps the slot and the other moves on, so the table -- hence the
  // numbering of the rows, hence the order of the additions on the matrix cores -- is the same whatever order the threads
  // arrive in (a first version, first come first served, gave sums that differed in the last bits from run to run).
  // True: an empty slot was taken (one more row in the set).
  auto insert = [&](uint32_t col) -> bool {
    uint32_t slot = set_slot(col);
#pragma unroll 1
    for (uint32_t t = 0; t < 2 * kTileH; ++t) {
      const uint32_t prev = atomicMin(&ht[slot].x, col);
      if (prev == col) return false;    // (already there)
      if (prev == kNoCol) return true;  // (an empty slot: kNoCol is the largest value)
      if (prev > col) col = prev;       // (took prev's slot: prev is carried on)
      slot = (slot + 1) & (kTileH - 1);
    }
    return false;
  };
  auto find = [&](uint32_t col) -> uint2 {  // the row's entry, or an empty one
    uint32_t slot = set_slot(col);
    uint2 e = ht[slot];
#pragma unroll 1
    for (uint32_t t = 0; t < kTileH && e.x != col && e.x != kNoCol; ++t) {
      slot = (slot + 1) & (kTileH - 1);
      e = ht[slot];
    }
    return e;
  };
  for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    // chunks are dealt with the groups fastest: blocks running together work on one stretch of all sequences
    const uint32_t seg = (uint32_t)(chunk / n_groups), grp = (uint32_t)(chunk % n_groups);
    const uint32_t pg = (uint32_t)(((uint64_t)grp * G) / kTileProbeG);  // (the probe's groups are of 64)
    if (!grel[pg] || seg >= gmax[pg]) continue;  // (not one organism: tile_group_probe_kernel; none of the group's sequences is this long)
    if (skip) {  // (uniform over the block)
      --skip;
      continue;
    }
    const uint32_t li = grp * G + tg;
    uint32_t r = 0;
    uint64_t off = 0, len = 0;
    bool mine = false;
    if (li < n_long) {
      r = olong[li];
      off = offsets[r];
      len = offsets[r + 1] - off;
      mine = seg < nseg[r];
    }
    const uint64_t n_win = len >= (uint64_t)k ? len - k + 1 : 0;
    const uint64_t s_beg = (uint64_t)seg * kTileS;  // the stretch's first base (and window) in the sequence
    const uint64_t w0 = s_beg + (uint64_t)tq * kPer, w1 = min(n_win, w0 + kPer);
    const uint32_t nv = (mine && w1 > w0) ? (uint32_t)(w1 - w0) : 0u;  // windows of this thread that exist
    if (stamps) t_last = __builtin_amdgcn_s_memtime();
    __syncthreads();
    // ---- 0. the stretch's bases into LDS: dword d of a sequence's row holds the four bytes at (stretch - a) + 4 d, a = the
    // stretch's address modulo 4, so that the loads are aligned dwords, 64 bytes per 16 lanes (bytes past either end of the
    // sequence are never read: the edge dwords are put together from the bytes that exist).  The byte loads this replaces
    // were a chain of 43 dependent round trips per thread, a third of the kernel.
    const uint8_t *ga = bases + off + s_beg;
    const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(ga) & 3u);
    {
      const int avail = mine ? (int)min<uint64_t>(len - s_beg, (uint64_t)(kTileS + k - 1)) : 0;
#pragma unroll
      for (uint32_t j = 0; j < (kTileStageW + 15) / 16; ++j) {
        const uint32_t d = tq + 16u * j;
        if (d < kTileStageW) {
          const int b0 = 4 * (int)d - (int)a;
          uint32_t v = 0;
          if (b0 >= 0 && b0 + 4 <= avail)
            v = *reinterpret_cast<const uint32_t *>(ga + b0);
          else
            for (int q = 0; q < 4; ++q)
              if (b0 + q >= 0 && b0 + q < avail) v |= (uint32_t)ga[b0 + q] << (8 * q);
          stage[tg * kTileStageW + d] = v;
        }
      }
    }
    if (tq == 0) s_slot[tg] = mine ? seg_off[r] + seg : ~0ull;
    for (uint32_t q = threadIdx.x; q < kTileH; q += THREADS) ht[q] = make_uint2(kNoCol, 0u);
    if (threadIdx.x == 0) {
      s_n = 0;
      s_new = 0;
      s_samp = 0;
    }
    if (threadIdx.x < G / SEEDEVERY) s_add[threadIdx.x] = 0;
    __syncthreads();
    stamp(0);  // the bases staged, the set cleared
    // ---- 1. the thread's 32 windows: hashes rolled out of LDS, then their twister rows, eight look-ups in flight at a time
    uint32_t cols[kPer];
    {
      const uint32_t *sg = stage + tg * kTileStageW + tq * 8u;
      uint32_t pw[4];  // the k - 1 bases before the first window's last base
#pragma unroll
      for (int i = 0; i < 4; ++i) pw[i] = __builtin_amdgcn_alignbyte(sg[i + 1], sg[i], a);
      const uint64_t plo = (uint64_t)pw[0] | ((uint64_t)pw[1] << 32), phi = (uint64_t)pw[2] | ((uint64_t)pw[3] << 32);
      H fwd = 0, rc = 0;
      int run = 0;
      for (int j = 0; j < k - 1; ++j) {
        const uint32_t c = base_code((uint32_t)((j < 8 ? plo >> (8 * j) : phi >> (8 * (j - 8))) & 0xFFull));
        fwd = ((fwd << 2) | (H)(c & 3u)) & mask;
        rc = (rc >> 2) | ((H)(3u - (c & 3u)) << shift);
        run = c < 4u ? run + 1 : 0;
      }
      const uint32_t sh = a + (uint32_t)(k - 1);
      const uint32_t *sm = sg + (sh >> 2);
      uint32_t mw[kPer / 4];  // the windows' last bases
#pragma unroll
      for (uint32_t i = 0; i < kPer / 4; ++i) mw[i] = __builtin_amdgcn_alignbyte(sm[i + 1], sm[i], sh & 3u);
      uint32_t vmask = 0;
#pragma unroll
      for (uint32_t i = 0; i < kPer; ++i) {
        const uint32_t c = base_code((mw[i >> 2] >> (8u * (i & 3u))) & 0xFFu);
        fwd = ((fwd << 2) | (H)(c & 3u)) & mask;
        rc = (rc >> 2) | ((H)(3u - (c & 3u)) << shift);
        run = c < 4u ? run + 1 : 0;
        cols[i] = (uint32_t)((content == KPOP_DNA_DS && rc < fwd) ? rc : fwd);  // (the hash, for now)
        vmask |= (run >= k && i < nv) ? (1u << i) : 0u;
      }
#pragma unroll
      for (uint32_t i0 = 0; i0 < kPer; i0 += 8) {
        uint4 q[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) q[u] = *reinterpret_cast<const uint4 *>(tv.rsel + (cols[i0 + u] >> 6));  // (a hash of k bases: inside the index)
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
          const uint64_t bits = ((uint64_t)q[u].y << 32) | q[u].x;
          const uint32_t b = cols[i0 + u] & 63u;
          const bool there = ((vmask >> (i0 + u)) & 1u) && ((bits >> b) & 1ull);
          cols[i0 + u] = there ? q[u].z + (uint32_t)__popcll(bits & ((1ull << b) - 1ull)) : kNoCol;
        }
      }
    }
    if (seed) {
#pragma unroll
      for (uint32_t i = 0; i < kPer; ++i) seedcols[(tg / SEEDEVERY) * kTileS + tq * kPer + i] = cols[i];
    }
    __syncthreads();
    stamp(1);  // everybody's rows found
    // ---- 2. the consensus set: a PRIMARY seed's rows whole, a thread each; then the other seeds, admitted in order while the set
    // stays within kTileSetCap rows, by what each would add at most (its rows not of the primary): a rule that does not depend on
    // who runs when.  The primary is seed 0 -- unless every other seed finds fewer than half of its rows there (sequence 0 of the
    // group is the odd one out: a contaminant, another lineage): then the set is started again from seed 1.
    constexpr uint32_t NS = G / SEEDEVERY;  // seeds (four)
    constexpr uint32_t EJ = ((NS - 1) * kTileS + THREADS - 1) / THREADS;
#pragma unroll 1
    for (uint32_t primary = 0; primary < 2; ++primary) {
      {
        bool took = false;
        if (threadIdx.x < kTileS) {
          const uint32_t col = seedcols[primary * kTileS + threadIdx.x];
          if (col != kNoCol) took = insert(col);
        }
        const uint32_t n = (uint32_t)__popcll(__ballot(took));
        if (lane == 0 && n) atomicAdd(&s_new, n);
      }
      __syncthreads();
      uint32_t ecol[EJ];
#pragma unroll
      for (uint32_t j = 0; j < EJ; ++j) {
        const uint32_t e = threadIdx.x + THREADS * j;  // (a wavefront's 64 entries are one seed's: 512 is a multiple of 64)
        ecol[j] = kNoCol;
        if (e < (NS - 1) * kTileS) {
          const uint32_t pos = 1u + e / kTileS, q = (primary + pos) % NS;  // pos: the seed's place in the order of admission
          const uint32_t col = seedcols[q * kTileS + e % kTileS];
          if (col != kNoCol && find(col).x == kNoCol) ecol[j] = col;
          const uint32_t n = (uint32_t)__popcll(__ballot(ecol[j] != kNoCol)), nv = (uint32_t)__popcll(__ballot(col != kNoCol));
          if (lane == 0 && nv) atomicAdd(&s_add[pos], n | (nv << 16));  // rows it would add | rows it has
        }
      }
      __syncthreads();
      uint32_t total = s_new, in = 1u, strangers = 0, others = 0;  // in, bit pos: that seed is admitted
#pragma unroll
      for (uint32_t pos = 1; pos < NS; ++pos) {
        const uint32_t add = s_add[pos] & 0xFFFFu, has = s_add[pos] >> 16;
        total += add;
        in |= (((in >> (pos - 1)) & 1u) && total <= kTileSetCap) ? (1u << pos) : 0u;
        others += has ? 1u : 0u;
        strangers += (has && add * 2u > has) ? 1u : 0u;
      }
      if (primary == 0 && others >= 2 && strangers == others) {  // (uniform) start again from seed 1
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < kTileH; q += THREADS) ht[q] = make_uint2(kNoCol, 0u);
        if (threadIdx.x == 0) s_new = 0;
        if (threadIdx.x < NS) s_add[threadIdx.x] = 0;
        __syncthreads();
        continue;
      }
#pragma unroll
      for (uint32_t j = 0; j < EJ; ++j) {
        const uint32_t e = threadIdx.x + THREADS * j;
        if (e < (NS - 1) * kTileS && ecol[j] != kNoCol && ((in >> (1u + e / kTileS)) & 1u)) (void)insert(ecol[j]);
      }
      break;
    }
    __syncthreads();
    stamp(2);  // the set built
    {
      // X cleared (the bases and the seeds' rows have been used); the occupied slots numbered in table order (2 slots a
      // thread, a block-wide scan)
      uint4 *X4 = reinterpret_cast<uint4 *>(Xw);
      for (uint32_t q = threadIdx.x; q < G * XS / 8; q += THREADS) X4[q] = make_uint4(0u, 0u, 0u, 0u);
      constexpr uint32_t SL = kTileH / THREADS;  // slots a thread
      const uint32_t q0 = threadIdx.x * SL;
      uint32_t kk[SL], occ = 0;
#pragma unroll
      for (uint32_t i = 0; i < SL; ++i) {
        kk[i] = ht[q0 + i].x;
        occ += kk[i] != kNoCol;
      }
      uint32_t incl = occ;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
      }
      if (lane == 63) s_wbase[wv] = incl;
      __syncthreads();
      uint32_t before = incl - occ;
      for (int w = 0; w < wv; ++w) before += s_wbase[w];
      if (threadIdx.x == THREADS - 1) s_n = before + occ;  // (<= kTileSetCap < TU)
#pragma unroll
      for (uint32_t i = 0; i < SL; ++i)
        if (kk[i] != kNoCol) {
          ht[q0 + i].y = before;
          ucol[before] = kk[i];
          ++before;
        }
      __syncthreads();
    }
    stamp(3);  // X cleared, the set numbered
    // ---- 3. every window against the set: a hit becomes the row's number (bit 31 set), a miss stays the row
    {
      uint32_t fh = 0;  // windows with a row | hits << 16
#pragma unroll
      for (uint32_t i = 0; i < kPer; ++i)
        if (cols[i] != kNoCol) {
          const uint2 e = find(cols[i]);
          fh += 1u;
          if (e.x != kNoCol) {
            cols[i] = 0x80000000u | e.y;
            fh += 65536u;
          }
        }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) fh += (uint32_t)__shfl_xor((int)fh, o, 64);
      if (lane == 0 && fh) atomicAdd(&s_samp, fh);
    }
    __syncthreads();
    stamp(4);  // the windows looked up in the set
    {
      // sequences that share little with the seeds (unrelated genomes, a divergent stretch): fewer than half the windows are
      // of the consensus, and the chunk is left to the streaming kernel
      const uint32_t fh = s_samp;  // (uniform: read after the barrier; at most 32,768 windows)
      if ((fh >> 16) * 2u < (fh & 0xFFFFu)) {
        missed();
        continue;
      }
    }
    misses = 0;
    backoff = 8;
    // ---- 4. X[sequence][number of the row] += 1, or the row onto the sequence's residual list
    const bool fused = tv.n_dims <= 64;  // (uniform)
    uint32_t wcnt = 0;                   // residual rows of this wavefront's four sequences
    {
      uint32_t found = 0, resm = 0;
#pragma unroll
      for (uint32_t i = 0; i < kPer; ++i) {
        const uint32_t c = cols[i];
        if (c != kNoCol) {
          ++found;
          if (c & 0x80000000u) {
            const uint32_t u = c & 0x7FFFFFFFu;
            if (!(dbg & 2)) atomicAdd(&Xw[tg * (XS / 2) + (u >> 1)], 1u << (16 * (u & 1u)));
          } else
            resm |= 1u << i;
        }
      }
      // the sequence's sixteen threads: where each one's residual rows go (window order), and the sequence's totals
      const uint32_t rcnt = (uint32_t)__popc(resm);
      uint32_t incl = rcnt;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 16);
        if (tq >= (uint32_t)o) incl += up;
      }
      const uint32_t rtot = (uint32_t)__shfl((int)incl, 15, 16);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) found += (uint32_t)__shfl_xor((int)found, o, 16);
      if (fused) {
        // up to 64 dimensions the wavefront gathers its four sequences' residual rows itself, under the MFMAs below: ONE list
        // per wavefront (its sequence in an entry's top two bits -- a row of k <= 15 needs 30 --, sequences in order, windows in
        // order), in a scratch of the block's own that it reads back past the L1 (the previous chunk's list may still sit there)
        uint32_t incl64 = rcnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t up = (uint32_t)__shfl_up((int)incl64, o, 64);
          if (lane >= o) incl64 += up;
        }
        wcnt = (uint32_t)__builtin_amdgcn_readlane((int)incl64, 63);
        uint32_t *wl = wave_lists + ((uint64_t)blockIdx.x * WAVES + wv) * (4 * kTileS) + (incl64 - rcnt);
        uint32_t pos = 0;
        if (!(dbg & 4)) {
#pragma unroll
          for (uint32_t i = 0; i < kPer; ++i)
            if ((resm >> i) & 1u) wl[pos++] = cols[i] | ((uint32_t)(lane >> 4) << 30);
        }
      }
      if (mine) {
        const uint64_t slot = s_slot[tg];
        if (!fused && !(dbg & 4)) {
          uint32_t *list = res_rows + slot * kTileS + (incl - rcnt);
          uint32_t pos = 0;
#pragma unroll
          for (uint32_t i = 0; i < kPer; ++i)
            if ((resm >> i) & 1u) list[pos++] = cols[i];
        }
        if (tq == 0) {
          slot_done[slot] = fused ? 1u : 1u + rtot;  // (> 1: rows listed for tile_residual_kernel)
          partial_cnt[slot] = found;
        }
      }
    }
    // (the columns are padded with row 0 of the twister against zero counts: the loops below have no branches, so that the
    // loads of the steps ahead stay in flight under the MFMAs)
    const uint32_t U = s_n;
    const bool ksplit = fused;
    const uint32_t UP = ksplit ? ((U + 16 * KQ - 1) & ~(16 * KQ - 1)) : ((U + 31) & ~31u);
    for (uint32_t u = U + threadIdx.x; u < UP; u += THREADS) ucol[u] = 0;
    if ((dbg & 32) && threadIdx.x == 0) {  // (bench.py's count of the matrix cores' work: chunks taken, rows of their sets as multiplied)
      atomicAdd(&g_tile_stamps[14], 1ull);
      atomicAdd(&g_tile_stamps[15], (unsigned long long)UP);
    }
    __syncthreads();
    stamp(5);  // the windows counted into X or listed
    const uint16_t *X16 = reinterpret_cast<const uint16_t *>(Xw);
    if (ksplit && !(dbg & 1)) {
      // ---- 3a. partial[64 x D] = X[64 x U] * T_U for D <= 64: wave wv owns the 16 dims of slice wv & 3 and a share (a quarter, or a half when G = 32) of the
      // columns, for all G sequences (G / 16 accumulator tiles) -- every row of T is loaded once per block, not once per M tile
      // (the rows' latency, not the MFMAs, was this phase: 0.86 of the kernel's 2.04 ms).  The quarters are added in order.
      const int ni = wv & 3, kh = wv >> 2;
      const uint32_t dc = 16u * ni + (lane & 15);
      const double *trow = tv.rows + min(dc, tv.d_pad - 1);  // (columns past the twister's are not written below)
      const uint32_t Uq = UP / KQ, c0 = (uint32_t)kh * Uq;   // Uq a multiple of 16: a multiple of four steps
      f64x4 acc[MT];
#pragma unroll
      for (uint32_t mi = 0; mi < MT; ++mi) acc[mi] = f64x4{0.0, 0.0, 0.0, 0.0};
      constexpr int PF = 4;  // (eight, with the columns padded to 128: no faster -- the phase is the MFMA pipe's now)
      double bb[PF];
      if (Uq) {
#pragma unroll
        for (int q = 0; q < PF; ++q) bb[q] = trow[(uint64_t)ucol[c0 + 4 * q + (lane >> 4)] * tv.d_pad];
      }
      const uint16_t *xrow = X16 + (lane & 15) * XS + c0 + (lane >> 4);
      // the residual gather, lane = dimension: 16 rows loaded before an iteration's 16 MFMAs (an iteration is ~4,000 cycles of a
      // matrix pipe four wavefronts share: the rows' latency), added after them, in list order -- sequence by sequence
      const uint32_t *wl = wave_lists + ((uint64_t)blockIdx.x * WAVES + wv) * (4 * kTileS);
      const double *grow = tv.rows + min((uint32_t)lane, tv.n_dims - 1);
      constexpr int GR = 8;
      double rsum[4] = {0.0, 0.0, 0.0, 0.0}, cur = 0.0;
      uint32_t cur_j = 0, gpos = 0;
      uint32_t colv = lane < (int)wcnt ? __hip_atomic_load(wl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kNoCol;
      double gv[GR];
      uint32_t gj[GR];  // (scalars)
      auto stash = [&]() {
        rsum[0] = cur_j == 0 ? cur : rsum[0];
        rsum[1] = cur_j == 1 ? cur : rsum[1];
        rsum[2] = cur_j == 2 ? cur : rsum[2];
        rsum[3] = cur_j == 3 ? cur : rsum[3];
      };
      auto gather_issue = [&]() {  // the next GR rows of the list (gpos is a multiple of GR, GR divides 64)
#pragma unroll
        for (int u = 0; u < GR; ++u) {
          const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)colv, (int)(gpos & 63u) + u);  // (scalar; kNoCol past the end)
          gj[u] = cj;
          gv[u] = grow[(uint64_t)(cj != kNoCol ? (cj & 0x3FFFFFFFu) : 0u) * tv.d_pad];
        }
      };
      auto gather_add = [&]() {
#pragma unroll
        for (int u = 0; u < GR; ++u) {
          if (gj[u] != kNoCol) {  // (scalar: uniform branches)
            const uint32_t j = gj[u] >> 30;
            if (j != cur_j) {  // (the list goes sequence by sequence: at most three changes)
              stash();
              cur = 0.0;
              cur_j = j;
            }
            cur = __dadd_rn(cur, gv[u]);
          }
        }
        gpos += GR;
        if ((gpos & 63u) == 0 && gpos < wcnt)
          colv = gpos + lane < wcnt ? __hip_atomic_load(wl + gpos + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kNoCol;
      };
      for (uint32_t q0 = 0; q0 < Uq / 4; q0 += PF) {
        const bool g = gpos < wcnt;  // (uniform)
        if (g) gather_issue();
#pragma unroll
        for (int qq = 0; qq < PF; ++qq) {
          const uint32_t q = q0 + qq;
          const double b = bb[qq];
          bb[qq] = trow[(uint64_t)ucol[c0 + min(4 * (q + PF) + (lane >> 4), Uq - 1)] * tv.d_pad];
#pragma unroll
          for (uint32_t mi = 0; mi < MT; ++mi)
            acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xrow[mi * 16 * XS + 4 * q], b, acc[mi], 0, 0, 0);
        }
        if (g) gather_add();
      }
      __syncthreads();  // every wave is done with X: its room takes the quarters' sums, [quarter][sequence][dim]
      stamp(6);  // the matrix cores
      double *P = reinterpret_cast<double *>(Xw);
#pragma unroll
      for (uint32_t mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)  // lane l holds rows (l >> 4) + 4 r of an M tile, column l & 15 of the wave's 16 dims
          P[((uint32_t)kh * G + 16 * mi + (lane >> 4) + 4 * rr) * 64 + dc] = acc[mi][rr];
      __syncthreads();
      // what is left of the wavefront's list (a divergent stretch: more rows than the MFMAs hid), then its four sequences' sums:
      // the quarters in order, then the residual rows' sum
      while (gpos < wcnt) {
        gather_issue();
        gather_add();
      }
      stash();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t g = 4u * wv + j, e = g * 64u + lane;
        const uint64_t sl = s_slot[g];
        double v = P[e];
#pragma unroll
        for (uint32_t kq = 1; kq < KQ; ++kq) v = __dadd_rn(v, P[kq * G * 64 + e]);
        v = __dadd_rn(v, rsum[j]);
        if (sl != ~0ull && (uint32_t)lane < tv.n_dims) partial[sl * tv.n_dims + lane] = v;
      }
    }
    // ---- 3b. the same for D > 64: wave wv owns M tile wv & 3 (16 sequences) and the 16 dims of slice wv >> 2 of every 64
    const int mi = wv % (int)MT, ni = wv / (int)MT;
    for (uint32_t d0 = 0; d0 < tv.n_dims && !ksplit && !(dbg & 1); d0 += 64) {
      const uint32_t U32 = UP;
      const uint32_t dc = d0 + 16u * ni + (lane & 15);
      const double *trow = tv.rows + min(dc, tv.d_pad - 1);  // (columns past the twister's are not written below)
      f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
      constexpr int PF = 8;
      double bb[PF];
      if (U32) {
#pragma unroll
        for (int q = 0; q < PF; ++q) bb[q] = trow[(uint64_t)ucol[4 * q + (lane >> 4)] * tv.d_pad];
      }
      const uint16_t *xrow = X16 + (16 * mi + (lane & 15)) * XS + (lane >> 4);
      for (uint32_t q0 = 0; q0 < U32 / 4; q0 += PF) {
#pragma unroll
        for (int qq = 0; qq < PF; ++qq) {
          const uint32_t q = q0 + qq;
          const double b = bb[qq];
          bb[qq] = trow[(uint64_t)ucol[min(4 * (q + PF) + (lane >> 4), U32 - 1)] * tv.d_pad];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xrow[4 * q], b, acc, 0, 0, 0);
        }
      }
      // lane l holds rows (l >> 4) + 4 r of the M tile, column l & 15 of the wave's 16 dims
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const uint64_t sl = s_slot[16 * mi + (lane >> 4) + 4 * rr];
        if (sl != ~0ull && dc < tv.n_dims) partial[sl * tv.n_dims + dc] = acc[rr];
      }
    }
    stamp(7);  // the sums written
  }
}

// The residual rows of the (sequence, segment) slots count_twist_tile_kernel took: one wavefront per slot, lane = dimension,
// eight row loads in flight, added in list (= window) order on top of the consensus sum.  These rows are private to their
// sequence: they come from HBM whatever the scheme (512 B a row at 64 dimensions), which is what bounds this kernel.
template <bool NT>
__global__ __launch_bounds__(256) void tile_residual_kernel(TwisterView tv, const uint32_t *__restrict__ slot_done,
                                                            const uint32_t *__restrict__ res_rows, double *__restrict__ partial,
                                                            const uint64_t *__restrict__ n_slots_ptr) {
  const uint64_t n_slots = *n_slots_ptr;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (uint64_t slot = (uint64_t)blockIdx.x * 4 + wv; slot < n_slots; slot += (uint64_t)gridDim.x * 4) {
    const uint32_t c = slot_done[slot];
    if (c <= 1u) continue;  // (0: the streaming kernel's; 1: no residual rows)
    const uint32_t cnt = c - 1u;
    const uint32_t *list = res_rows + slot * kTileS;
    for (uint32_t d0 = 0; d0 < tv.n_dims; d0 += 64) {
      const uint32_t d = d0 + lane;
      const double *base = tv.rows + (d < tv.n_dims ? d : tv.n_dims - 1);
      double acc = 0.0;
      for (uint32_t j0 = 0; j0 < cnt; j0 += 64) {
        const uint32_t col = (j0 + lane < cnt) ? list[j0 + lane] : kNoCol;
        const uint32_t m = min(64u, cnt - j0);
        constexpr int GU = kGatherUnroll;
#pragma unroll
        for (int j = 0; j < 64; j += GU) {
          if ((uint32_t)j >= m) break;  // (uniform)
          double v[GU];
#pragma unroll
          for (int u = 0; u < GU; ++u) {
            const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)col, j + u);
            const uint32_t cs = cj != kNoCol ? cj : 0u;  // (scalar)
            const double x = NT ? __builtin_nontemporal_load(base + (uint64_t)cs * tv.d_pad) : base[(uint64_t)cs * tv.d_pad];
            v[u] = cj != kNoCol ? x : 0.0;
          }
#pragma unroll
          for (int u = 0; u < GU; ++u) acc = __dadd_rn(acc, v[u]);
        }
      }
      if (d < tv.n_dims) partial[slot * tv.n_dims + d] = __dadd_rn(partial[slot * tv.n_dims + d], acc);
    }
  }
}

// What count_twist_tile_kernel left: the (sequence, segment) pairs whose slot it did not take, for the streaming kernel (one
// wavefront per sequence with segments; the list's order is not the batch's, and no sum depends on it).
__global__ __launch_bounds__(256) void tile_todo_kernel(const uint32_t *__restrict__ nseg, const uint64_t *__restrict__ seg_off,
                                                        const uint32_t *__restrict__ olong, const uint64_t *__restrict__ n_long_ptr,
                                                        const uint32_t *__restrict__ slot_done, uint2 *__restrict__ todo, uint32_t *__restrict__ n_todo) {
  // a thread per sequence counts its slots left, the block takes room for all of them with ONE atomic (a wavefront per sequence
  // and an atomic each: 5,000 atomics on one word, 59 us), then every sequence's pairs are written by its thread
  __shared__ uint32_t s_wsum[4], s_base;
  const uint32_t n_long = (uint32_t)*n_long_ptr;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (uint32_t l0 = blockIdx.x * 256u; l0 < n_long; l0 += gridDim.x * 256u) {  // (uniform)
    const uint32_t li = l0 + threadIdx.x;
    uint32_t r = 0, ns = 0, left = 0;
    uint64_t s0 = 0;
    if (li < n_long) {
      r = olong[li];
      ns = nseg[r];
      s0 = seg_off[r];
      for (uint32_t seg = 0; seg < ns; ++seg) left += slot_done[s0 + seg] == 0u;
    }
    uint32_t incl = left;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
      if (lane >= o) incl += up;
    }
    __syncthreads();
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t tot = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
      s_base = tot ? atomicAdd(n_todo, tot) : 0u;
    }
    __syncthreads();
    uint32_t pos = s_base + incl - left;
    for (int w = 0; w < wv; ++w) pos += s_wsum[w];
    for (uint32_t seg = 0; seg < ns && left; ++seg)
      if (slot_done[s0 + seg] == 0u) todo[pos++] = make_uint2(r, seg);
  }
}

__global__ __launch_bounds__(256) void combine_partials_kernel(const uint32_t *__restrict__ nseg,
                                                               const uint64_t *__restrict__ seg_off,
                                                               const double *__restrict__ partial,
                                                               const uint32_t *__restrict__ partial_cnt,
                                                               uint32_t n_dims, int normalize,
                                                               double *__restrict__ out, const uint32_t *__restrict__ long_ids,
                                                               const uint32_t *__restrict__ n_long_ptr) {
  const uint32_t n_long = *n_long_ptr;
  for (uint32_t b = blockIdx.x; b < n_long; b += gridDim.x) {  // (the grid is sized from the caller's n_bases: a stride, should that understate)
    const uint32_t r = long_ids[b];
    const uint32_t ns = nseg[r];
    if (ns == 0) continue;
    const uint64_t s0 = seg_off[r];
    uint64_t found = 0;
    for (uint32_t s = 0; s < ns; ++s) found += partial_cnt[s0 + s];
    const double acc = (double)found;  // lib/Twister.ml:158, exact: integer counts
    const bool norm = normalize && acc != 0.0;
    for (uint32_t d = threadIdx.x; d < n_dims; d += blockDim.x) {
      double t = 0.0;
      uint32_t s = 0;
      for (; s + 8 <= ns; s += 8) {  // (eight loads in flight, added in segment order: 59 segments of 512 windows were 59 round trips)
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(s0 + s + u) * n_dims + d];
#pragma unroll
        for (int u = 0; u < 8; ++u) t = __dadd_rn(t, v[u]);
      }
      for (; s < ns; ++s) t = __dadd_rn(t, partial[(s0 + s) * n_dims + d]);
      out[(uint64_t)r * n_dims + d] = norm ? t / acc : t;
    }
  }
}

// ---------------------------------------------------------------------------
// synthetic reads: base i of read r from SplitMix64 stream
// element (first_read + r) * read_len + i
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64d(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void synth_reads_kernel(uint64_t seed, uint64_t n_reads, uint32_t read_len, uint64_t first_read,
                                   uint8_t *__restrict__ bases, uint64_t *__restrict__ offsets) {
  const uint64_t total = n_reads * read_len;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint64_t g = first_read * read_len + i;
    uint64_t z = mix64d(seed + (g + 1) * 0x9E3779B97F4A7C15ull);
    bases[i] = (uint8_t)("ACGT"[z >> 62]);
  }
  if (offsets)
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_reads; r += stride)
      offsets[r] = r * read_len;
}

// ---------------------------------------------------------------------------
// dispatch helpers
// ---------------------------------------------------------------------------
static int pick_R(uint32_t max_windows) {
  if (max_windows <= 64) return 1;
  if (max_windows <= 128) return 2;
  if (max_windows <= 256) return 4;
  if (max_windows <= 512) return 8;
  return 0;
}

// bytes of device scratch count_wave needs: the ticket counter and one look-back word per block
// ... and, for the two-level look-back, a word and an arrival counter per group of 64 blocks (lookback.h)
static inline uint64_t count_wave_max_blocks(uint32_t n) { return (uint64_t)div_up(n, 2 * kWavesPerBlock) + 1; }
static inline uint64_t count_wave_scratch_bytes(uint32_t n) {
  const uint64_t blocks = count_wave_max_blocks(n), groups = blocks / kLookGroup + 2;
  return 64 + blocks * 8 + groups * 8 + ((groups * 4 + 63) & ~63ull);
}

template <typename H, int SB>
static int launch_count_wave_sb(int R, const uint8_t *bases, const uint64_t *offsets, uint32_t n, int k, int content, void *scratch,
                                uint64_t *oh, uint32_t *oc, uint64_t *oo, hipStream_t st) {
  dim3 block(64 * kWavesPerBlock);
  const auto grid = [&](int rw) { return dim3(div_up(n, (uint32_t)rw * kWavesPerBlock)); };
  uint32_t *ticket = reinterpret_cast<uint32_t *>(scratch);
  uint64_t *state = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(scratch) + 64);
  uint64_t *gstate = state + count_wave_max_blocks(n);
  uint32_t *garr = reinterpret_cast<uint32_t *>(gstate + count_wave_max_blocks(n) / kLookGroup + 2);
  KPOP_HIP(hipMemsetAsync(scratch, 0, count_wave_scratch_bytes(n), st));
  switch (R) {
    case 1: count_wave_kernel<1, H, SB><<<grid(reads_per_wave<1>()), block, 0, st>>>(bases, offsets, n, k, content, ticket, state, gstate, garr, oh, oc, oo, ctx().tune_dbg); break;
    case 2: count_wave_kernel<2, H, SB><<<grid(reads_per_wave<2>()), block, 0, st>>>(bases, offsets, n, k, content, ticket, state, gstate, garr, oh, oc, oo, ctx().tune_dbg); break;
    case 4: count_wave_kernel<4, H, SB><<<grid(reads_per_wave<4>()), block, 0, st>>>(bases, offsets, n, k, content, ticket, state, gstate, garr, oh, oc, oo, ctx().tune_dbg); break;
    case 8: count_wave_kernel<8, H, SB><<<grid(reads_per_wave<8>()), block, 0, st>>>(bases, offsets, n, k, content, ticket, state, gstate, garr, oh, oc, oo, ctx().tune_dbg); break;
    default: KPOP_FAIL(KPOP_ERR_INVALID, "launch_count_wave: R=%d", R);
  }
  KPOP_LAUNCH_CHECK();
  return 0;
}

// keys are 32-bit while the hash (2 bits per base, 5 per residue) leaves the all-ones sentinel free: <= 30 bits.
// oh / oc need one entry per window (the worst case); oo gets n + 1 offsets.
static int launch_count_wave(int R, const uint8_t *bases, const uint64_t *offsets, uint32_t n, int k, int content, void *scratch,
                             uint64_t *oh, uint32_t *oc, uint64_t *oo, hipStream_t st) {
  const bool narrow = hash_bits(k, content) <= 30;
  if (content == KPOP_PROTEIN)
    return narrow ? launch_count_wave_sb<uint32_t, 5>(R, bases, offsets, n, k, content, scratch, oh, oc, oo, st)
                  : launch_count_wave_sb<uint64_t, 5>(R, bases, offsets, n, k, content, scratch, oh, oc, oo, st);
  return narrow ? launch_count_wave_sb<uint32_t, 2>(R, bases, offsets, n, k, content, scratch, oh, oc, oo, st)
                : launch_count_wave_sb<uint64_t, 2>(R, bases, offsets, n, k, content, scratch, oh, oc, oo, st);
}

// k range and content of the counting entry points (bin/KPopCount.ml:113: <= 30 for DNA, <= 12 for protein)
static int check_count_args(int k, int content, const char *who) {
  if (content != KPOP_DNA_DS && content != KPOP_DNA_SS && content != KPOP_PROTEIN)
    KPOP_FAIL(KPOP_ERR_INVALID, "%s: Invalid_content(%d)", who, content);
  const int kmax = content == KPOP_PROTEIN ? kMaxKProtein : kMaxK;
  if (k < 1 || k > kmax) KPOP_FAIL(KPOP_ERR_INVALID, "%s: k=%d out of range 1..%d", who, k, kmax);
  return 0;
}

template <typename H, int U, bool NT>
static int launch_count_twist_wave_v(int R, const TwisterView &tv, const uint8_t *bases, const uint64_t *offsets,
                                     const uint32_t *ids, uint32_t n, int content, int normalize, double *out,
                                     hipStream_t st) {
  dim3 grid(div_up(n, kWavesPerBlock)), block(64 * kWavesPerBlock);
  switch (R) {
    case 1: count_twist_wave_kernel<1, H, U, NT><<<grid, block, (size_t)ctx().tune_ldspad, st>>>(tv, bases, offsets, ids, n, content, normalize, out); break;
    case 2: count_twist_wave_kernel<2, H, U, NT><<<grid, block, (size_t)ctx().tune_ldspad, st>>>(tv, bases, offsets, ids, n, content, normalize, out); break;
    case 4: count_twist_wave_kernel<4, H, U, NT><<<grid, block, (size_t)ctx().tune_ldspad, st>>>(tv, bases, offsets, ids, n, content, normalize, out); break;
    case 8: count_twist_wave_kernel<8, H, U, NT><<<grid, block, (size_t)ctx().tune_ldspad, st>>>(tv, bases, offsets, ids, n, content, normalize, out); break;
    default: KPOP_FAIL(KPOP_ERR_INVALID, "launch_count_twist_wave: R=%d", R);
  }
  KPOP_LAUNCH_CHECK();
  return 0;
}

// protein (bin/KPopCount.ml:246-248): five bits a residue, keys of 32 bits up to k = 6 and of 64 beyond
template <typename H>
static int launch_count_twist_wave_protein(int R, const TwisterView &tv, const uint8_t *bases, const uint64_t *offsets, const uint32_t *ids, uint32_t n,
                                           int normalize, double *out, hipStream_t st) {
  dim3 grid(div_up(n, kWavesPerBlock)), block(64 * kWavesPerBlock);
#define KPOP_WPR(RR) count_twist_wave_kernel<RR, H, 8, false, false, 5><<<grid, block, (size_t)ctx().tune_ldspad, st>>>(tv, bases, offsets, ids, n, KPOP_PROTEIN, normalize, out)
  switch (R) {
    case 1: KPOP_WPR(1); break;
    case 2: KPOP_WPR(2); break;
    case 4: KPOP_WPR(4); break;
    case 8: KPOP_WPR(8); break;
    default: KPOP_FAIL(KPOP_ERR_INVALID, "launch_count_twist_wave: R=%d", R);
  }
#undef KPOP_WPR
  KPOP_LAUNCH_CHECK();
  return 0;
}

template <typename H>
static int launch_count_twist_wave(int R, TwisterView tv, const uint8_t *bases, const uint64_t *offsets,
                                   const uint32_t *ids, uint32_t n, int content, int normalize, double *out,
                                   hipStream_t st) {
  if (content == KPOP_PROTEIN)  // (its own key width: H is the caller's choice for DNA)
    return hash_bits(tv.hk, content) <= 30 ? launch_count_twist_wave_protein<uint32_t>(R, tv, bases, offsets, ids, n, normalize, out, st)
                                           : launch_count_twist_wave_protein<uint64_t>(R, tv, bases, offsets, ids, n, normalize, out, st);
  const Context &c = ctx();
  const bool nt = c.tune_nt == 1 || (c.tune_nt == 2 && (uint64_t)tv.n_rows * tv.d_pad * 8 > kStreamingRowBytes);
  if (c.tune_unroll == 16)
    return nt ? launch_count_twist_wave_v<H, 16, true>(R, tv, bases, offsets, ids, n, content, normalize, out, st)
              : launch_count_twist_wave_v<H, 16, false>(R, tv, bases, offsets, ids, n, content, normalize, out, st);
  return nt ? launch_count_twist_wave_v<H, 8, true>(R, tv, bases, offsets, ids, n, content, normalize, out, st)
            : launch_count_twist_wave_v<H, 8, false>(R, tv, bases, offsets, ids, n, content, normalize, out, st);
}


// the pipelined tile kernel (tile_pipe.h) takes this twister: a rank-select index of words (k <= 14), a row's number in 29 bits (it
// shares a word with three bits of tag in the residual lists) and a row's offset in 128-byte units in 32 (beyond 64 dimensions)
static bool tile_route_pipe(const kpop_twister *tw) {
  return tw->d_rsel && (tw->hk ? tw->hk : tw->k) <= 15 && ctx().tune_tilepipe != 0 && tw->n_rows <= (1ull << 29) && (uint64_t)tw->n_rows * (tw->d_pad / 16) < 0xFFFFFFFFull;
}
// ... in its three-stage form (tile_pipe.h, WIDE: the columns unit by unit): more than 64 dimensions (kpop_tune("tilewide", 1): at any number of them -- up to 64 the same bits as the other)
static bool tile_route_wide(const kpop_twister *tw) { return tile_route_pipe(tw) && (tw->n_dims > 64 || ctx().tune_tilewide == 1); }
// bytes of per-slot tables a call of the wide route may take: 4 GiB per 64 columns, a quarter of the device's memory at most (the
// same for every call: how a batch is cut into sub-batches must not depend on what happens to be free)
static uint64_t tile_workspace_cap(uint32_t d_pad) {
  if (ctx().tune_tilecap_mb > 0) return (uint64_t)ctx().tune_tilecap_mb << 20;
  uint64_t cap = (4ull << 30) * div_up(d_pad, 64u);
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) cap = std::min<uint64_t>(cap, std::max<uint64_t>(4ull << 30, (uint64_t)total_b / 4));
  return cap;
}

// the same launch from the packed form of the batch (reads of up to 512 windows only: the caller has checked)
template <typename H>
static int launch_count_twist_wave_packed(int R, TwisterView tv, const uint32_t *codes, const uint32_t *invalid, const uint64_t *offsets, uint32_t n, int content,
                                          int normalize, double *out, hipStream_t st) {
  const Context &c = ctx();
  const bool nt = c.tune_nt == 1 || (c.tune_nt == 2 && (uint64_t)tv.n_rows * tv.d_pad * 8 > kStreamingRowBytes);
  dim3 grid(div_up(n, kWavesPerBlock)), block(64 * kWavesPerBlock);
  const uint8_t *b = reinterpret_cast<const uint8_t *>(codes);
#define KPOP_WP(RR, NTV) count_twist_wave_kernel<RR, H, 8, NTV, true><<<grid, block, (size_t)c.tune_ldspad, st>>>(tv, b, offsets, nullptr, n, content, normalize, out, invalid)
  switch (R) {
    case 1: if (nt) KPOP_WP(1, true); else KPOP_WP(1, false); break;
    case 2: if (nt) KPOP_WP(2, true); else KPOP_WP(2, false); break;
    case 4: if (nt) KPOP_WP(4, true); else KPOP_WP(4, false); break;
    case 8: if (nt) KPOP_WP(8, true); else KPOP_WP(8, false); break;
    default: KPOP_FAIL(KPOP_ERR_INVALID, "launch_count_twist_wave_packed: R=%d", R);
  }
#undef KPOP_WP
  KPOP_LAUNCH_CHECK();
  return 0;
}

static int check_offsets(const uint64_t *offsets, uint32_t n, uint64_t *max_len) {
  uint64_t m = 0;
  for (uint32_t r = 0; r < n; ++r) {
    if (offsets[r + 1] < offsets[r]) KPOP_FAIL(KPOP_ERR_INVALID, "offsets are not non-decreasing at read %u", r);
    m = std::max(m, offsets[r + 1] - offsets[r]);
  }
  *max_len = m;
  return 0;
}

}  // namespace kpop

using namespace kpop;

// ---------------------------------------------------------------------------
// C ABI: device-resident entry points
// ---------------------------------------------------------------------------
extern "C" int kpop_debug_counters(uint64_t *out, int n) {
  KPOP_TRY(require_init());
  if (!out || n < 0 || n > 16) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_debug_counters: out null or n not in 0..16");
  unsigned long long h[16], z[16] = {};
  KPOP_HIP(hipDeviceSynchronize());
  KPOP_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile_stamps), sizeof h));
  KPOP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stamps), z, sizeof z));
  for (int i = 0; i < n; ++i) out[i] = h[i];
  return KPOP_OK;
}

extern "C" int kpop_dev_synth_reads(uint64_t seed, uint64_t n_reads, uint32_t read_len, uint64_t first_read,
                                    uint8_t *d_bases, uint64_t *d_offsets, void *stream) {
  KPOP_TRY(require_init());
  if (!d_bases && n_reads && read_len) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_synth_reads: null bases");
  synth_reads_kernel<<<dim3(2048), dim3(256), 0, as_stream(stream)>>>(seed, n_reads, read_len, first_read, d_bases,
                                                                       d_offsets);
  KPOP_LAUNCH_CHECK();
  return KPOP_OK;
}

extern "C" int kpop_dev_count_twist(const kpop_twister *tw, const uint8_t *d_bases, const uint64_t *d_offsets,
                                    uint32_t n_reads, uint64_t n_bases, uint32_t max_len, int content, int normalize,
                                    double *d_out, void *stream) {
  KPOP_TRY(require_init());
  if (!tw || !d_offsets || !d_out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_count_twist: null argument");
  if (content != KPOP_DNA_DS && content != KPOP_DNA_SS && content != KPOP_PROTEIN)
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_count_twist: Invalid_content(%d)", content);
  const bool protein = content == KPOP_PROTEIN;  // five bits a residue (bin/KPopCount.ml:246-248): the wave and the streaming kernel, no tile route
  if (n_reads == 0) return KPOP_OK;
  hipStream_t st = as_stream(stream);
  const TwisterView tv = view_of(tw);
  if (protein && (tv.hk > kMaxKProtein || hash_bits(tv.hk, content) > 2 * tw->k))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_count_twist: protein k=%d (at most %d, and its %d-bit hashes must fit the %d bits the twister was loaded with)", tv.hk,
              kMaxKProtein, hash_bits(tv.hk, content), 2 * tw->k);
  const uint32_t max_windows = (max_len >= (uint32_t)tv.hk) ? max_len - tv.hk + 1 : 0;
  // Assemblies through more than 64 dimensions whose partial rows pass the tile route's bound (below) go through in SUB-BATCHES of
  // sequences, m of them at most m x max_len bases, each a call of its own.
  if (!protein && max_windows > kWaveMaxWindows && tile_route_wide(tw) && ctx().tune_dense != 0 && !ctx().tune_seg && n_reads >= kTileMinSeqs) {
    const uint64_t max_long0 = std::min<uint64_t>(n_reads, n_bases / kWaveMaxWindows + 1);
    const uint64_t per_slot = (uint64_t)tw->n_dims * 8 + 4 + 4 + 8, cap = tile_workspace_cap(tw->d_pad);
    if ((n_bases / kTileS + max_long0) * per_slot > cap) {
      const uint64_t m = cap / (((uint64_t)max_len / kTileS + 2) * per_slot);
      if (m >= 4096 && m < n_reads) {
        for (uint64_t r0 = 0; r0 < n_reads; r0 += m) {
          const uint32_t n = (uint32_t)std::min<uint64_t>(m, n_reads - r0);
          KPOP_TRY(kpop_dev_count_twist(tw, d_bases, d_offsets + r0, n, std::min<uint64_t>(n_bases, (uint64_t)n * max_len), max_len, content, normalize,
                                        d_out + r0 * tw->n_dims, stream));
        }
        return KPOP_OK;
      }
    }
  }
  // reads of up to 512 windows: one wavefront per read (the kernel skips longer reads)
  const int R = pick_R(std::min(max_windows, kWaveMaxWindows));
  const int flags = (normalize ? 1 : 0) | (max_windows > kWaveMaxWindows ? 2 : 0);  // bit 1: the streaming pass below takes the long ones
  if (tv.hk <= 15)
    KPOP_TRY(launch_count_twist_wave<uint32_t>(R, tv, d_bases, d_offsets, nullptr, n_reads, content, flags, d_out, st));
  else
    KPOP_TRY(launch_count_twist_wave<uint64_t>(R, tv, d_bases, d_offsets, nullptr, n_reads, content, flags, d_out, st));
  if (max_windows <= kWaveMaxWindows) return KPOP_OK;
  // longer sequences: segment table, streaming kernel, ordered combine.  Segment = the stretch of a sequence one block
  // sums: its rows should sit in one XCD's L2 (4 MB) next to those of the neighbouring segment, see the kernel.
  const Context &cx = ctx();
  uint32_t seg_windows = cx.tune_seg ? (uint32_t)cx.tune_seg : std::max<uint32_t>(1024u, std::min<uint32_t>(kSegWindows, (uint32_t)((3ull << 19) / ((uint64_t)tw->d_pad * 8)) / 64 * 64));
  // Batches of assemblies first go through count_twist_tile_kernel (the consensus rows of a stretch of 64 sequences gathered
  // once and multiplied on the matrix cores, the private rows listed for tile_residual_kernel); what it leaves -- stretches
  // that share little with their seeds, found from a sample -- is the streaming kernel's as before.  kpop_tune("dense", 0)
  // opts out (the streaming kernel alone: the reference's order of additions within a segment).
  bool tiles_wanted = !protein && cx.tune_dense != 0 && tv.rsel && n_reads >= kTileMinSeqs && tv.hk <= 15 && !cx.tune_seg;
  const bool nt = cx.tune_nt == 1;
  // (a sequence with segments has more than kWaveMaxWindows windows: at most this many of them)
  const uint32_t max_long = (uint32_t)std::min<uint64_t>(n_reads, n_bases / kWaveMaxWindows + 1);
  // The tile route cuts sequences into 512-window segments, so its per-slot tables are sized for n_bases / 512 slots whatever the
  // probe later finds (BASELINE config 3, D = 64: 1.5 GB of partial rows against 0.28 GB; beyond 64 dimensions 2 KB of residual
  // list a slot on top).  Where that would pass 4 GiB of workspace -- per stream -- the batch keeps the streaming kernel.
  // Beyond 64 dimensions the pipelined kernel takes the columns unit by unit (tile_pipe.h, WIDE); its bound is 4 GiB per 64
  // columns (BASELINE config 3 at 256 dimensions: 6.1 GB of partial rows; at the reference's 1,635, README.md:1029, 39 GB -- what 288 GB
  // are for), and a batch beyond THAT goes through in sub-batches of sequences, each sized from max_len (below).
  const bool pipe_able = tile_route_pipe(tw);
  const bool wide = tile_route_wide(tw);
  if (tiles_wanted) {
    const uint64_t slots = n_bases / std::min(seg_windows, kTileS) + max_long;
    const uint64_t per_slot = (uint64_t)tw->n_dims * 8 + 4 + 4 + 8 + (tw->n_dims > 64 && !wide ? (uint64_t)kTileS * 4 : 0);
    const uint64_t cap = wide ? tile_workspace_cap(tw->d_pad) : (4ull << 30);
    if (slots * per_slot > cap) tiles_wanted = false;  // (a batch that could go through in sub-batches has, above)
  }
  const bool tiles = tiles_wanted;
  const uint32_t seg_least = tiles ? std::min(seg_windows, kTileS) : seg_windows;  // (the shortest segment any sequence is cut into)
  const uint64_t max_slots = n_bases / seg_least + max_long;  // every such sequence adds at most W/seg + 1 segments
  const uint64_t nb = scan_blocks(n_reads);
  const uint32_t max_seg = div_up(max_windows, seg_least);
  const uint32_t max_groups = div_up(max_long, kTileProbeG);
  const int tile_g = cx.tune_tileg == 64 ? 64 : 32;  // sequences a chunk of the tile kernel (kpop_tune("tileg"))
  const uint64_t bytes_nseg = ((uint64_t)n_reads * 4 + 63) & ~63ull, bytes_off = ((uint64_t)(n_reads + 1) * 8 + 63) & ~63ull,
                 bytes_sums = ((nb + 1) * 8 + 63) & ~63ull, bytes_cnt = (max_slots * 4 + 63) & ~63ull,
                 bytes_part = (max_slots * tw->n_dims * 8 + 63) & ~63ull, bytes_done = tiles ? ((max_slots * 4 + 63) & ~63ull) : 0,
                 bytes_long = ((uint64_t)n_reads * 4 + 64 + 63) & ~63ull,
                 bytes_olong = tiles ? (((uint64_t)max_long * 4 + 63) & ~63ull) : 0, bytes_gmax = tiles ? (((uint64_t)max_groups * 4 + 63) & ~63ull) : 0,
                 bytes_res = tiles && tw->n_dims > 64 && !wide ? max_slots * kTileS * 4 : 0, bytes_todo = tiles ? ((max_slots * 8 + 1024 + 63) & ~63ull) : 0,
                 bytes_perread = tiles ? bytes_nseg : 0;
  // up to 64 dimensions: the pipelined kernel (tile_pipe.h; kpop_tune("tilepipe", 0): round 4's kernel, phases one after the other)
  // (kpop_tune("tilepipe", 0): round 4's kernel, phases one after the other, the residual rows in a launch of their own beyond 64 dimensions)
  const bool pipe = tiles && pipe_able;  // (a row's number shares a word with three bits of tag in the residual lists; beyond 64 dimensions a row's offset in 128-byte units is a word)
  const uint64_t bytes_wlists = !tiles ? 0 : pipe ? (uint64_t)cx.n_cus * (wide ? 3 : 2) * 8 * kPipeListCap * 4 : (uint64_t)cx.n_cus * 16 * 4 * kTileS * 4;
  void *ws = nullptr;
  KPOP_TRY(ctx().ws_for(st).ensure(bytes_nseg + bytes_off + 2 * bytes_sums + bytes_cnt + bytes_part + bytes_done + bytes_long + bytes_olong + 2 * bytes_gmax +
                                       bytes_res + bytes_todo + 2 * bytes_perread + bytes_wlists, &ws));
  char *wp = reinterpret_cast<char *>(ws);
  auto carve = [&](uint64_t bytes) {
    char *p = wp;
    wp += bytes;
    return p;
  };
  uint32_t *nseg = reinterpret_cast<uint32_t *>(carve(bytes_nseg));
  uint64_t *seg_off = reinterpret_cast<uint64_t *>(carve(bytes_off));
  uint64_t *sums = reinterpret_cast<uint64_t *>(carve(bytes_sums));
  uint64_t *sums2 = reinterpret_cast<uint64_t *>(carve(bytes_sums));
  uint32_t *pcnt = reinterpret_cast<uint32_t *>(carve(bytes_cnt));
  double *part = reinterpret_cast<double *>(carve(bytes_part));
  uint32_t *slot_done = tiles ? reinterpret_cast<uint32_t *>(carve(bytes_done)) : nullptr;
  uint32_t *n_long = reinterpret_cast<uint32_t *>(carve(bytes_long));
  uint32_t *long_ids = n_long + 16;
  uint32_t *olong = reinterpret_cast<uint32_t *>(carve(bytes_olong));
  uint32_t *gmax = reinterpret_cast<uint32_t *>(carve(bytes_gmax));
  uint32_t *grel = reinterpret_cast<uint32_t *>(carve(bytes_gmax));
  uint32_t *res_rows = reinterpret_cast<uint32_t *>(carve(bytes_res));
  uint32_t *n_todo = reinterpret_cast<uint32_t *>(carve(bytes_todo));
  uint2 *todo = tiles ? reinterpret_cast<uint2 *>(n_todo + 256) : nullptr;  // ([0] the list's length, [1..255] a take-the-next counter per streaming launch)
  uint32_t *lpos = tiles ? reinterpret_cast<uint32_t *>(carve(bytes_perread)) : nullptr;
  uint32_t *segw = tiles ? reinterpret_cast<uint32_t *>(carve(bytes_perread)) : nullptr;
  uint32_t *wave_lists = reinterpret_cast<uint32_t *>(carve(bytes_wlists));
  KPOP_HIP(hipMemsetAsync(n_long, 0, 64, st));
  if (tiles) {
    // the long sequences in batch order, groups of 64 of them probed for being one organism -- before the segments are laid out
    KPOP_TRY(exclusive_scan(LoadLongW{d_offsets, tv.hk}, StoreLong{olong, lpos}, n_reads, sums2, st));
    tile_group_probe_kernel<<<dim3(max_groups), dim3(256), 0, st>>>(d_bases, d_offsets, tv.hk, content, olong, sums2 + nb, gmax, grel);
    KPOP_LAUNCH_CHECK();
  }
  segment_count_kernel<<<dim3(div_up(n_reads, 256)), dim3(256), 0, st>>>(d_offsets, n_reads, tv.hk, seg_windows, nseg, long_ids, n_long, lpos, grel, segw);
  KPOP_LAUNCH_CHECK();
  KPOP_TRY(exclusive_scan(LoadU32{nseg}, StoreU64{seg_off}, n_reads, sums, st));
  if (tiles) {
    KPOP_HIP(hipMemsetAsync(slot_done, 0, bytes_done, st));
    const size_t lds = (size_t)kTileH * 8 + tile_set_rows(tile_g) * 4 + (size_t)tile_g * (tile_set_rows(tile_g) + 2) * 2;
    static PerSlotOnce once;
    if (!once()) {
      KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&count_twist_tile_kernel<uint32_t, 64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((size_t)kTileH * 8 + tile_set_rows(64) * 4 + (size_t)64 * (tile_set_rows(64) + 2) * 2)));
      KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&count_twist_tile_kernel<uint32_t, 32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((size_t)kTileH * 8 + tile_set_rows(32) * 4 + (size_t)32 * (tile_set_rows(32) + 2) * 2)));
      once() = true;
    }
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((uint64_t)div_up(max_long, tile_g) * max_seg, (uint64_t)cx.n_cus * (tile_g == 32 ? 2 : 1));
    if (pipe) {
      static PerSlotOnce once_pipe;
      if (!once_pipe()) {
        KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&count_twist_tile_pipe_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPipeLdsBytes));
        KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&count_twist_tile_pipe_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPipeLdsBytes));
        KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&count_twist_tile_pipe_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPipeLdsBytes));
        KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&count_twist_tile_pipe_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPipeLdsBytes));
        once_pipe() = true;
      }
      const uint32_t pblocks = (uint32_t)std::min<uint64_t>((uint64_t)div_up(max_long, kPipeG) * max_seg, (uint64_t)cx.n_cus);
      const int tdbg = ((ctx().tune_dbg >> 24) & 127) | ((ctx().tune_pipeprio & 3) << 8) | ((ctx().tune_pipeprio & 4) ? 128 : 0);  // (bit 7: the producers' gather with plain loads)
#define KPOP_PIPE(A, W) count_twist_tile_pipe_kernel<A, W><<<dim3(pblocks), dim3(1024), kPipeLdsBytes, st>>>(tv, d_bases, d_offsets, content, nseg, seg_off, part, pcnt, olong, sums2 + nb, gmax, grel, max_seg, slot_done, wave_lists, tdbg)
      // (the ablation switches: a build of the kernel of their own, so that the product's loops carry no test of them;
      //  kpop_tune("tilewide", 1): the three-stage kernel at any number of dimensions -- at up to 64 the same bits as the other)
      const bool wide_k = wide;
      if (tdbg & 15) {
        if (wide_k) KPOP_PIPE(true, true); else KPOP_PIPE(true, false);
      } else {
        if (wide_k) KPOP_PIPE(false, true); else KPOP_PIPE(false, false);
      }
#undef KPOP_PIPE
    } else if (tile_g == 64)
      count_twist_tile_kernel<uint32_t, 64><<<dim3(blocks), dim3(1024), lds, st>>>(tv, d_bases, d_offsets, content, nseg, seg_off, part, pcnt, olong, sums2 + nb,
                                                                                  gmax, grel, max_seg, slot_done, res_rows, wave_lists, ctx().tune_dbg >> 24);
    else
      count_twist_tile_kernel<uint32_t, 32><<<dim3(blocks), dim3(512), lds, st>>>(tv, d_bases, d_offsets, content, nseg, seg_off, part, pcnt, olong, sums2 + nb,
                                                                                 gmax, grel, max_seg, slot_done, res_rows, wave_lists, ctx().tune_dbg >> 24);
    KPOP_LAUNCH_CHECK();
    if (tw->n_dims > 64 && !pipe) {  // (the pipelined kernel -- and round 4's up to 64 dimensions -- has gathered the residual rows itself)
      const dim3 rgrid(capped_grid((max_slots + 3) / 4));
      if (nt)
        tile_residual_kernel<true><<<rgrid, dim3(256), 0, st>>>(tv, slot_done, res_rows, part, sums + nb);
      else
        tile_residual_kernel<false><<<rgrid, dim3(256), 0, st>>>(tv, slot_done, res_rows, part, sums + nb);
      KPOP_LAUNCH_CHECK();
    }
    KPOP_HIP(hipMemsetAsync(n_todo, 0, 1024, st));
    tile_todo_kernel<<<dim3(std::min<uint32_t>(div_up(max_long, 256), 4096u)), dim3(256), 0, st>>>(nseg, seg_off, olong, sums2 + nb, slot_done, todo, n_todo);
    KPOP_LAUNCH_CHECK();
  }
  // (with a to-do list the pairs all have work, and how many is on the device: a grid that fills the chip, striding)
  dim3 grid(tiles ? (uint32_t)std::min<uint64_t>(max_slots, (uint64_t)cx.n_cus * 8) : capped_grid((uint64_t)max_long * max_seg));  // (eight blocks of 256 are resident per CU)
  // The dimensions go in passes of up to 256 (four blocks of 64 a lane), 128 or 64 -- one LAUNCH per pass, so that a pass never
  // loads blocks it has no dimensions for (300 dimensions: 256 + 44; a run-time guard on the loads brought the per-window
  // branches back: 2.4 -> 5.2 ms at 64 dimensions).  Every launch hashes the windows again: once per 256 dimensions, not per 64.
#define KPOP_STREAM_B(H, NT, B, D0, D1) count_twist_stream_kernel<H, NT, B, KPOP_STREAM_SB><<<grid, dim3(256), 0, st>>>(tv, d_bases, d_offsets, content, nseg, seg_off, part, pcnt, long_ids, tiles ? n_todo : n_long, max_seg, seg_windows, todo, segw, tiles ? n_todo + 1 + launch_no++ : nullptr, D0, D1)
  int launch_no = 0;  // (a take-the-next counter per launch)
  if (tiles && tw->n_dims > 255u * 256u) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_dev_count_twist: %u dimensions (at most 65,280 with the tile route on; kpop_tune(\"dense\", 0) lifts that)", tw->n_dims);
#define KPOP_STREAM(H, NT)                                                        \
  do {                                                                            \
    for (uint32_t pos = 0; pos < tw->n_dims;) {                                    \
      const uint32_t rem = tw->n_dims - pos;                                      \
      if (rem > 128) {                                                            \
        const uint32_t take = std::min(rem, 256u);                                \
        KPOP_STREAM_B(H, NT, 4, pos, pos + take);                                 \
        pos += take;                                                              \
      } else if (rem > 64) {                                                      \
        KPOP_STREAM_B(H, NT, 2, pos, pos + rem);                                  \
        pos += rem;                                                               \
      } else {                                                                    \
        KPOP_STREAM_B(H, NT, 1, pos, pos + rem);                                  \
        pos += rem;                                                               \
      }                                                                           \
    }                                                                             \
  } while (0)
#define KPOP_STREAM_SB 2
  if (content == KPOP_PROTEIN) {
#undef KPOP_STREAM_SB
#define KPOP_STREAM_SB 5
    if (hash_bits(tv.hk, content) <= 30) KPOP_STREAM(uint32_t, false); else KPOP_STREAM(uint64_t, false);
#undef KPOP_STREAM_SB
#define KPOP_STREAM_SB 2
  } else if (tv.hk <= 15) {
    if (nt) KPOP_STREAM(uint32_t, true); else KPOP_STREAM(uint32_t, false);
  } else {
    if (nt) KPOP_STREAM(uint64_t, true); else KPOP_STREAM(uint64_t, false);
  }
#undef KPOP_STREAM_SB
#undef KPOP_STREAM
#undef KPOP_STREAM_B
  KPOP_LAUNCH_CHECK();
  combine_partials_kernel<<<dim3(max_long), dim3(256), 0, st>>>(nseg, seg_off, part, pcnt, tw->n_dims, normalize, d_out, long_ids, n_long);
  KPOP_LAUNCH_CHECK();
  return KPOP_OK;
}

// packed.hip's fast way: a batch whose reads ALL fit the one-wavefront-per-read kernel is twisted straight from its packed words (no
// bytes are ever made); returns 1 when it did, 0 when the batch holds longer sequences (the caller spreads the bases and takes the usual way)
int count_twist_wave_from_packed(const kpop_twister *tw, const uint32_t *d_codes, const uint32_t *d_invalid, const uint64_t *d_offsets, uint32_t n_reads,
                                 uint32_t max_len, int content, int normalize, double *d_out, hipStream_t st, int *done) {
  const TwisterView tv = view_of(tw);
  const uint32_t max_windows = (max_len >= (uint32_t)tv.hk) ? max_len - tv.hk + 1 : 0;
  *done = 0;
  if (max_windows > kWaveMaxWindows || (content != KPOP_DNA_DS && content != KPOP_DNA_SS)) return KPOP_OK;
  const int R = pick_R(max_windows);
  if (tv.hk <= 15)
    KPOP_TRY(launch_count_twist_wave_packed<uint32_t>(R, tv, d_codes, d_invalid, d_offsets, n_reads, content, normalize ? 1 : 0, d_out, st));
  else
    KPOP_TRY(launch_count_twist_wave_packed<uint64_t>(R, tv, d_codes, d_invalid, d_offsets, n_reads, content, normalize ? 1 : 0, d_out, st));
  *done = 1;
  return KPOP_OK;
}

// device-resident -L counting for reads of up to 512 windows: one launch (count_wave_kernel), enqueue only
extern "C" uint64_t kpop_dev_count_reads_scratch_bytes(uint32_t n_reads, uint32_t max_len, int k) {
  (void)max_len;
  (void)k;
  return count_wave_scratch_bytes(n_reads) + 256;
}

extern "C" int kpop_dev_count_reads(const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads, uint32_t max_len,
                                    int k, int content, void *d_scratch, uint64_t *d_out_hash, uint32_t *d_out_count,
                                    uint64_t *d_out_offsets, void *stream) {
  KPOP_TRY(require_init());
  if (!d_offsets || !d_scratch || !d_out_hash || !d_out_count || !d_out_offsets)
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_count_reads: null argument");
  KPOP_TRY(check_count_args(k, content, "kpop_dev_count_reads"));
  const uint32_t max_windows = (max_len >= (uint32_t)k) ? max_len - k + 1 : 0;
  if (max_windows > kWaveMaxWindows)
    KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_dev_count_reads: reads of more than %u windows go through kpop_count_reads", kWaveMaxWindows);
  hipStream_t st = as_stream(stream);
  if (n_reads == 0) {
    KPOP_HIP(hipMemsetAsync(d_out_offsets, 0, 8, st));
    return KPOP_OK;
  }
  void *scratch = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(d_scratch) + 63) & ~(uintptr_t)63);
  return launch_count_wave(pick_R(max_windows), d_bases, d_offsets, n_reads, k, content, scratch, d_out_hash, d_out_count,
                           d_out_offsets, st);
}

extern "C" int kpop_dev_twist(const kpop_twister *tw, const uint64_t *d_hash, const double *d_value,
                              const uint64_t *d_offsets, uint32_t n_spectra, uint64_t max_lines, int normalize,
                              double *d_out, void *stream) {
  KPOP_TRY(require_init());
  if (!tw || !d_offsets || !d_out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_twist: null argument");
  if (n_spectra == 0) return KPOP_OK;
  // max_lines = the longest spectrum of the batch (0 = unknown): up to 512 lines the columns found while summing the
  // counts stay in registers for the gather.  Callers pass 0 when they do not know; a spectrum longer than max_lines says
  // comes back as a row of NaNs.
  return launch_twist_csr<double>(view_of(tw), d_hash, d_value, d_offsets, n_spectra, max_lines, normalize, d_out, as_stream(stream));
}

// ---------------------------------------------------------------------------
// C ABI: host-buffer entry points
// ---------------------------------------------------------------------------
extern "C" int kpop_count_reads(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content,
                                int per_read, uint64_t *out_hash, uint32_t *out_count, uint64_t *out_offsets,
                                uint64_t out_capacity) {
  KPOP_TRY(require_init());
  ArenaScope scratch;
  if (!offsets || !out_offsets || (!out_hash && out_capacity) || (!out_count && out_capacity))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_count_reads: null argument");
  KPOP_TRY(check_count_args(k, content, "kpop_count_reads"));
  if (per_read) out_offsets[0] = 0; else out_offsets[0] = out_offsets[1] = 0;
  if (n_reads == 0) return KPOP_OK;
  uint64_t max_len = 0;
  KPOP_TRY(check_offsets(offsets, n_reads, &max_len));
  const uint64_t max_windows = (max_len >= (uint64_t)k) ? max_len - k + 1 : 0;
  if (!per_read) {  // -l: one spectrum for everything (bin/KPopCount.ml:60)
    uint64_t written = 0;
    return sorted_count_batch(bases, offsets, n_reads, k, content, 0, out_hash, out_count, out_offsets, out_capacity,
                              &written);
  }
  if (max_windows > kWaveMaxWindows) {
    // genomes: sort path, in sub-batches whose (spectrum id | hash) keys fit 63 bits
    const int id_bits_max = 63 - hash_bits(k, content);
    const uint64_t sub = id_bits_max >= 32 ? n_reads : std::max<uint64_t>(1, 1ull << id_bits_max);
    uint64_t pos = 0;
    std::vector<uint64_t> loc;
    for (uint64_t r0 = 0; r0 < n_reads; r0 += sub) {
      const uint32_t nr = (uint32_t)std::min<uint64_t>(sub, n_reads - r0);
      loc.assign(nr + 1, 0);
      uint64_t written = 0;
      KPOP_TRY(sorted_count_batch(bases, offsets + r0, nr, k, content, 1, out_hash + pos, out_count + pos, loc.data(),
                                  out_capacity - pos, &written));
      for (uint32_t i = 0; i <= nr; ++i) out_offsets[r0 + i] = pos + loc[i];
      pos += written;
    }
    out_offsets[n_reads] = pos;
    return KPOP_OK;
  }
  const int R = pick_R((uint32_t)max_windows);
  const uint64_t base0 = offsets[0], n_bases = offsets[n_reads] - base0;
  hipStream_t st = nullptr;
  DevBuf d_bases, d_off, d_scr, d_oo, d_oh, d_oc;
  uint64_t worst = 0;  // one entry per window
  std::vector<uint64_t> rel(n_reads + 1);
  for (uint32_t r = 0; r <= n_reads; ++r) rel[r] = offsets[r] - base0;
  for (uint32_t r = 0; r < n_reads; ++r) {
    const uint64_t len = rel[r + 1] - rel[r];
    worst += len >= (uint64_t)k ? len - k + 1 : 0;
  }
  KPOP_TRY(d_bases.alloc(n_bases));
  KPOP_TRY(d_off.alloc((uint64_t)(n_reads + 1) * 8));
  KPOP_TRY(d_scr.alloc(count_wave_scratch_bytes(n_reads)));
  KPOP_TRY(d_oo.alloc((uint64_t)(n_reads + 1) * 8));
  KPOP_TRY(d_oh.alloc(worst * 8));
  KPOP_TRY(d_oc.alloc(worst * 4));
  if (n_bases) KPOP_HIP(hipMemcpyAsync(d_bases.p, bases + base0, n_bases, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(d_off.p, rel.data(), (uint64_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
  KPOP_TRY(launch_count_wave(R, d_bases.as<uint8_t>(), d_off.as<uint64_t>(), n_reads, k, content, d_scr.p, d_oh.as<uint64_t>(),
                             d_oc.as<uint32_t>(), d_oo.as<uint64_t>(), st));
  KPOP_HIP(hipMemcpyAsync(out_offsets, d_oo.p, (uint64_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  const uint64_t total = out_offsets[n_reads];
  if (total > out_capacity)
    KPOP_FAIL(KPOP_ERR_CAPACITY, "kpop_count_reads: %llu distinct (read,k-mer) pairs, capacity %llu",
              (unsigned long long)total, (unsigned long long)out_capacity);
  if (total) {
    KPOP_HIP(hipMemcpyAsync(out_hash, d_oh.p, total * 8, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipMemcpyAsync(out_count, d_oc.p, total * 4, hipMemcpyDeviceToHost, st));
  }
  KPOP_HIP(hipStreamSynchronize(st));
  return KPOP_OK;
}

extern "C" int kpop_count_twist(const kpop_twister *tw, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                                int content, int normalize, double *out) {
  KPOP_TRY(require_init());
  ArenaScope scratch;
  if (!tw || !offsets || (!out && n_reads)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_count_twist: null argument");
  if (n_reads == 0) return KPOP_OK;
  uint64_t max_len = 0;
  KPOP_TRY(check_offsets(offsets, n_reads, &max_len));
  const uint64_t base0 = offsets[0], n_bases = offsets[n_reads] - base0;
  hipStream_t st = nullptr;
  DevBuf d_bases, d_off, d_out;
  KPOP_TRY(d_bases.alloc(n_bases));
  KPOP_TRY(d_off.alloc((uint64_t)(n_reads + 1) * 8));
  KPOP_TRY(d_out.alloc((uint64_t)n_reads * tw->n_dims * 8));
  std::vector<uint64_t> rel(n_reads + 1);
  for (uint32_t r = 0; r <= n_reads; ++r) rel[r] = offsets[r] - base0;
  if (n_bases) KPOP_HIP(hipMemcpyAsync(d_bases.p, bases + base0, n_bases, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(d_off.p, rel.data(), (uint64_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
  if (max_len > 0xFFFFFFFFull) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_count_twist: sequence longer than 2^32 bases");
  // kpop_tune("dense", 2): batches of assemblies at small k (every sequence holds a fifth or more of the twister's k-mers on
  // average, the twister small enough for the dense image) go through the u32 image and the f64 matrix cores; results
  // agree with the sparse kernels to rounding
  const bool dense_image = content != KPOP_PROTEIN && ctx().tune_dense == 2 && tw->n_rows > 0 && tw->n_rows <= 36864 && tw->n_dims <= 256 && n_reads >= 64 &&
                           (double)n_bases >= 0.2 * (double)tw->n_rows * (double)n_reads;
  if (dense_image) {
    DevBuf d_work;
    KPOP_TRY(d_work.alloc(kpop_dev_count_twist_dense_workspace_bytes(tw, n_reads)));
    KPOP_TRY(kpop_dev_count_twist_dense(tw, d_bases.as<uint8_t>(), d_off.as<uint64_t>(), n_reads, content, normalize, d_work.p,
                                        d_out.as<double>(), st));
  } else
  KPOP_TRY(kpop_dev_count_twist(tw, d_bases.as<uint8_t>(), d_off.as<uint64_t>(), n_reads, n_bases, (uint32_t)max_len,
                                content, normalize, d_out.as<double>(), st));
  KPOP_HIP(hipMemcpyAsync(out, d_out.p, (uint64_t)n_reads * tw->n_dims * 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  return KPOP_OK;
}

// Reads -> the twisted rows that kpop_count_reads(k, content, per_read = 1) followed by kpop_twist would give, bit for
// bit, with the spectra never leaving the device: what `KPopCount -L | KPopTwistDB -k /dev/stdin` computes
// (bin/KPopCount.ml:36-50 into lib/Twister.ml:146-188) minus the text in between.  The count uses the caller's k; the
// twister may have been loaded with the other k of the pair that shares its name width (kpop_twister_set_count_k).
// When every sequence fits one wavefront and n_dims > 32 the fused kernel IS that computation (same ascending chain of
// unfused multiply-adds); otherwise the spectra are built on the device and twisted line by line by twist_csr_kernel.
extern "C" int kpop_spectra_twist(const kpop_twister *tw, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                                  int k, int content, int normalize, double *out) {
  KPOP_TRY(require_init());
  ArenaScope scratch;
  if (!tw || !offsets || (!out && n_reads)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_spectra_twist: null argument");
  KPOP_TRY(check_count_args(k, content, "kpop_spectra_twist"));
  if (hash_bits(k, content) > 2 * tw->k)
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_spectra_twist: k=%d (%d-bit hashes) above the twister's k=%d (%d bits)", k, hash_bits(k, content), tw->k, 2 * tw->k);
  if (n_reads == 0) return KPOP_OK;
  uint64_t max_len = 0;
  KPOP_TRY(check_offsets(offsets, n_reads, &max_len));
  if (max_len > 0xFFFFFFFFull) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_spectra_twist: sequence longer than 2^32 bases");
  const uint64_t max_windows = (max_len >= (uint64_t)k) ? max_len - k + 1 : 0;
  hipStream_t st = nullptr;
  kpop_twister view = *tw;  // same device arrays, the caller's counting k
  view.hk = k;
  const TwisterView tv = view_of(&view);
  const uint64_t out_bytes = (uint64_t)n_reads * tw->n_dims * 8;
  DevBuf d_out;
  KPOP_TRY(d_out.alloc(out_bytes));
  if (max_windows <= kWaveMaxWindows) {
    const uint64_t base0 = offsets[0], n_bases = offsets[n_reads] - base0;
    DevBuf d_bases, d_off;
    KPOP_TRY(d_bases.alloc(n_bases));
    KPOP_TRY(d_off.alloc((uint64_t)(n_reads + 1) * 8));
    std::vector<uint64_t> rel(n_reads + 1);
    for (uint32_t r = 0; r <= n_reads; ++r) rel[r] = offsets[r] - base0;
    if (n_bases) KPOP_HIP(hipMemcpyAsync(d_bases.p, bases + base0, n_bases, hipMemcpyHostToDevice, st));
    KPOP_HIP(hipMemcpyAsync(d_off.p, rel.data(), (uint64_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
    const int R = pick_R((uint32_t)max_windows);
    if (tw->n_dims > 32) {
      if (k <= 15)
        KPOP_TRY(launch_count_twist_wave<uint32_t>(R, tv, d_bases.as<uint8_t>(), d_off.as<uint64_t>(), nullptr, n_reads, content, normalize, d_out.as<double>(), st));
      else
        KPOP_TRY(launch_count_twist_wave<uint64_t>(R, tv, d_bases.as<uint8_t>(), d_off.as<uint64_t>(), nullptr, n_reads, content, normalize, d_out.as<double>(), st));
    } else {
      DevBuf d_scr, d_oo, d_oh, d_oc;
      uint64_t worst = 0;
      for (uint32_t r = 0; r < n_reads; ++r) {
        const uint64_t len = rel[r + 1] - rel[r];
        worst += len >= (uint64_t)k ? len - k + 1 : 0;
      }
      KPOP_TRY(d_scr.alloc(count_wave_scratch_bytes(n_reads)));
      KPOP_TRY(d_oo.alloc((uint64_t)(n_reads + 1) * 8));
      KPOP_TRY(d_oh.alloc(worst * 8));
      KPOP_TRY(d_oc.alloc(worst * 4));
      KPOP_TRY(launch_count_wave(R, d_bases.as<uint8_t>(), d_off.as<uint64_t>(), n_reads, k, content, d_scr.p, d_oh.as<uint64_t>(),
                                 d_oc.as<uint32_t>(), d_oo.as<uint64_t>(), st));
      KPOP_TRY(launch_twist_csr<uint32_t>(tv, d_oh.as<uint64_t>(), d_oc.as<uint32_t>(), d_oo.as<uint64_t>(), n_reads, max_windows, normalize,
                                          d_out.as<double>(), st));
    }
  } else {
    // genomes: sort path, in sub-batches whose (spectrum id | hash) keys fit 63 bits, then the line-by-line twist
    const int id_bits_max = 63 - hash_bits(k, content);
    const uint64_t sub = id_bits_max >= 32 ? n_reads : std::max<uint64_t>(1, 1ull << id_bits_max);
    for (uint64_t r0 = 0; r0 < n_reads; r0 += sub) {
      const uint32_t nr = (uint32_t)std::min<uint64_t>(sub, n_reads - r0);
      ArenaScope batch;
      SortedSpectra S;
      KPOP_TRY(sorted_count_device(bases, offsets + r0, nr, k, content, 1, ~0ull, S, st));
      // (no spectrum has more lines than its sequence has windows: a few long genomes then take the segmented launch)
      KPOP_TRY(launch_twist_csr<uint32_t>(tv, S.d_oh.as<uint64_t>(), S.d_oc.as<uint32_t>(), S.d_oo.as<uint64_t>(), nr, max_windows, normalize,
                                          d_out.as<double>() + r0 * tw->n_dims, st));
      KPOP_HIP(hipStreamSynchronize(st));  // the batch's scratch goes back to the arena at the end of this scope
    }
  }
  KPOP_HIP(hipMemcpyAsync(out, d_out.p, out_bytes, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  return KPOP_OK;
}

extern "C" int kpop_twist(const kpop_twister *tw, const uint64_t *hash, const double *value, const uint64_t *offsets,
                          uint32_t n_spectra, int normalize, double *out) {
  KPOP_TRY(require_init());
  ArenaScope scratch;
  if (!tw || !offsets || (!out && n_spectra)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twist: null argument");
  if (n_spectra == 0) return KPOP_OK;
  uint64_t max_lines = 0;
  KPOP_TRY(check_offsets(offsets, n_spectra, &max_lines));
  const uint64_t base0 = offsets[0], n_lines = offsets[n_spectra] - base0;
  if (n_lines && (!hash || !value)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twist: null hash/value");
  hipStream_t st = nullptr;
  DevBuf d_h, d_v, d_off, d_out;
  KPOP_TRY(d_h.alloc(n_lines * 8));
  KPOP_TRY(d_v.alloc(n_lines * 8));
  KPOP_TRY(d_off.alloc((uint64_t)(n_spectra + 1) * 8));
  KPOP_TRY(d_out.alloc((uint64_t)n_spectra * tw->n_dims * 8));
  std::vector<uint64_t> rel(n_spectra + 1);
  for (uint32_t r = 0; r <= n_spectra; ++r) rel[r] = offsets[r] - base0;
  if (n_lines) {
    KPOP_HIP(hipMemcpyAsync(d_h.p, hash + base0, n_lines * 8, hipMemcpyHostToDevice, st));
    KPOP_HIP(hipMemcpyAsync(d_v.p, value + base0, n_lines * 8, hipMemcpyHostToDevice, st));
  }
  KPOP_HIP(hipMemcpyAsync(d_off.p, rel.data(), (uint64_t)(n_spectra + 1) * 8, hipMemcpyHostToDevice, st));
  // kpop_tune("dense", 1): always the contraction on the matrix cores; 2: when the batch is dense enough for it to win
  // (measured crossover, DESIGN.md 5.9: >= 256 spectra holding >= 40 % of the twister's k-mers each); 0 (default): never --
  // the sparse form is the reference's order of additions, the dense one agrees with it to rounding only
  const int dense = ctx().tune_dense;
  if (dense == 1 || (dense == 2 && n_spectra >= 256 && tw->n_rows > 0 && (double)n_lines >= 0.4 * (double)n_spectra * (double)tw->n_rows)) {
    DevBuf d_work;
    KPOP_TRY(d_work.alloc(kpop_dev_twist_dense_workspace_bytes(tw, n_spectra)));
    // lines ascending by hash in every spectrum (what KPopCount writes; the reference's own Hashtbl order is unspecified):
    // the fused kernel densifies them inside the contraction; any other order goes through the dense image in HBM
    bool ascending = true;
    for (uint32_t r = 0; r < n_spectra && ascending; ++r)
      for (uint64_t i = offsets[r] + 1; i < offsets[r + 1]; ++i)
        if (hash[i] < hash[i - 1]) {
          ascending = false;
          break;
        }
    // (beyond ~160 dimensions the image route's GEMM tiles are full and it overtakes the fused kernel: 4,096 genome spectra,
    // k = 7: D = 128 0.54 against 0.82 ms, D = 200 1.51 against 0.97 -- tools/ab_dense_twist.py with AB_DIMS)
    if (ascending && tw->n_dims <= 160)
      KPOP_TRY(kpop_dev_twist_dense_sorted(tw, d_h.as<uint64_t>(), d_v.as<double>(), d_off.as<uint64_t>(), n_spectra, normalize, d_work.p,
                                           d_out.as<double>(), st));
    else
      KPOP_TRY(kpop_dev_twist_dense(tw, d_h.as<uint64_t>(), d_v.as<double>(), d_off.as<uint64_t>(), n_spectra, normalize, d_work.p,
                                    d_out.as<double>(), st));
  } else {
    KPOP_TRY(kpop_dev_twist(tw, d_h.as<uint64_t>(), d_v.as<double>(), d_off.as<uint64_t>(), n_spectra, max_lines,
                            normalize, d_out.as<double>(), st));
  }
  KPOP_HIP(hipMemcpyAsync(out, d_out.p, (uint64_t)n_spectra * tw->n_dims * 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  return KPOP_OK;
}

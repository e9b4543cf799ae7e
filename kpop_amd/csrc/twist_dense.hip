// twist_dense.hip -- the twist as a dense contraction on the f64 matrix cores (north_star's "LDS-tiled batched GEMM on
// MFMA"): twisted[B x D] = X[B x n_kmers] * T[n_kmers x D], X = the normalised spectra laid out
// densely, T = the twister's rows as they stand in HBM (k-mer-major, [n_rows][d_pad]).
//
// This does 2 * n_kmers * D flops per spectrum whatever the spectrum holds, against 2 * nnz * D for the reference's
// sparse mat-vec (lib/Twister.ml:183), so it only pays when spectra are dense AND many: it reads the twister once per
// batch tile instead of once per line.  DESIGN.md 5.9 has the measured crossover (tools/ab_dense_twist.py): genomes at
// k <= 8-9; never for 150 bp reads (139 of 8.39 M columns) and never at k = 12 (the dense image of a batch is too large).
// The order of additions differs from the reference's chain (blocked K, split-K slabs added in slab order): results agree
// to rounding (tests: 1e-12), and the CLI path keeps the chain unless kpop_tune("dense", 1|2) asks for this one.
#include <algorithm>

#include "gemm_f64.h"
#include "kmer.h"
#include "twister.h"

namespace kpop {

// acc[s] = sum of the values of the lines the twister knows (lib/Twister.ml:158); one wave per spectrum
__global__ __launch_bounds__(256) void dense_acc_kernel(TwisterView tv, const uint64_t *__restrict__ hash, const double *__restrict__ value,
                                                        const uint64_t *__restrict__ offsets, uint32_t s0, uint32_t n, double *__restrict__ acc) {
  const int lane = threadIdx.x & 63;
  const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n) return;
  const uint64_t lo = offsets[s0 + s], hi = offsets[s0 + s + 1];
  double part = 0.0;
  for (uint64_t i = lo + lane; i < hi; i += 64)
    if (lookup_col(tv, hash[i]) != kNoCol) part += value[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) acc[s] = part;
}

// X[s][row(hash)] += value / acc  (duplicate lines of a spectrum add up, :160-163)
__global__ __launch_bounds__(256) void dense_fill_kernel(TwisterView tv, const uint64_t *__restrict__ hash, const double *__restrict__ value,
                                                         const uint64_t *__restrict__ offsets, uint32_t s0, uint32_t n, int normalize,
                                                         const double *__restrict__ acc, double *__restrict__ X, uint64_t ldx) {
  const int lane = threadIdx.x & 63;
  const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n) return;
  const uint64_t lo = offsets[s0 + s], hi = offsets[s0 + s + 1];
  const double a = acc[s];
  const bool norm = normalize && a != 0.0;
  for (uint64_t i = lo + lane; i < hi; i += 64) {
    const uint32_t col = lookup_col(tv, hash[i]);
    if (col != kNoCol) atomicAdd(&X[(uint64_t)s * ldx + col], norm ? value[i] / a : value[i]);
  }
}

__global__ void dense_copy_out_kernel(const double *__restrict__ C, uint32_t n, uint32_t n_dims, uint32_t ldc, double *__restrict__ out) {
  const uint64_t total = (uint64_t)n * n_dims, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) out[e] = C[(e / n_dims) * ldc + e % n_dims];
}

constexpr uint64_t kDenseTileBytes = 2ull << 30;  // dense image of one batch tile

static uint32_t dense_tile_rows(const kpop_twister *tw, uint32_t n_spectra) {
  const uint64_t per = std::max<uint64_t>(1, tw->n_rows) * 8;
  uint64_t b = std::max<uint64_t>(128, kDenseTileBytes / per / 128 * 128);
  return (uint32_t)std::min<uint64_t>(b, std::max<uint32_t>(n_spectra, 1));
}
static uint32_t dense_splits(uint32_t rows, const kpop_twister *tw) {
  const uint32_t tiles = div_up(rows, kGT) * div_up(tw->n_dims, kGT);
  const uint32_t want = std::max(1u, (uint32_t)ctx().n_cus * 2 / std::max(1u, tiles));
  const uint64_t max_by_k = std::max<uint64_t>(1, tw->n_rows / 512);
  return (uint32_t)std::min<uint64_t>(std::min<uint64_t>(want, max_by_k), 64);
}


// ---------------------------------------------------------------------------
// The fused form: spectra whose lines ascend by hash (what the counting kernels produce) are densified INSIDE the
// contraction -- no image of X in HBM, no memset, no f64 atomics on global memory.
//
//   block   = 64 spectra x all dims (D = 64 fills N: four 16-wide MFMA column tiles, one per wave) x one slab of k-mers;
//   panel   = 128 consecutive twister rows.  The lines of a spectrum that fall into a panel are a contiguous run of its
//             CSR (sorted lines, rows in ascending hash order), at most 128 of them unless lines repeat: every wave pulls
//             two coalesced 64-line loads per spectrum at the spectrum's cursor, looks the columns up, keeps the prefix
//             that lies inside the panel and adds it into the LDS panel A[spectrum][column] (ds_add_f64: repeated lines
//             add up, lib/Twister.ml:160-163); the cursor moves on by what was taken;
//   MFMA    = per panel 32 steps of v_mfma_f64_16x16x4_f64 per wave and M tile, A from LDS, B = the twister's rows
//             straight from L2 (prefetched into registers before the panel is built: 4 MB at k = 7, read once per
//             block and shared by the 64 blocks of a slab);
//   two blocks per CU (66.5 KB of LDS each): one builds its panel while the other multiplies.
// acc (lib/Twister.ml:158) is summed on the way and the division is applied ONCE to the finished sums (the sparse kernel
// divides every count first): equal to rounding, like the rest of the dense route.  Slabs are added in slab order.
// A spectrum whose lines turn out NOT to ascend comes back as a row of NaNs.
// ---------------------------------------------------------------------------
constexpr int kDM = 64, kDK = 128, kDStride = kDK + 2;
constexpr uint32_t kDenseBatch = 16384;  // spectra per launch

__device__ __forceinline__ uint64_t first_hash_of_row(const TwisterView &tv, uint64_t row) {
  if (row >= tv.n_rows) return ~0ull;
  if (tv.sorted_hash) return tv.sorted_hash[row];
  // select on the rank index: the word whose run of ranks holds `row`, then the bit
  uint64_t lo = 0, hi = ((1ull << (2 * tv.k)) + 63) / 64;
  while (hi - lo > 1) {
    const uint64_t mid = (lo + hi) >> 1;
    if (tv.rsel[mid].prefix <= row) lo = mid; else hi = mid;
  }
  uint64_t bits = tv.rsel[lo].bits;
  uint32_t r = (uint32_t)(row - tv.rsel[lo].prefix);
  while (r--) bits &= bits - 1;
  return lo * 64 + (uint64_t)(__ffsll((long long)bits) - 1);
}

// one 8-spectrum group of a wave: its loads, and what is done with them once they have landed
struct DenseGroup {
  uint64_t h1[8], h2[8];
  double v1[8], v2[8];
};

// cursor / end of spectrum j of the wave live in lane j (j < 16): wave-uniform reads by v_readlane
__device__ __forceinline__ uint64_t lane_u64(uint64_t v, int j) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, j);
}

template <bool TWO>
__device__ __forceinline__ void dense_issue(DenseGroup &G, int g, uint64_t my_cur, uint64_t my_end, const uint64_t *__restrict__ hash,
                                            const double *__restrict__ value, int lane) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint64_t cur = lane_u64(my_cur, 8 * g + j), end = lane_u64(my_end, 8 * g + j);
    const uint64_t i1 = cur + lane, i2 = i1 + 64;
    G.h1[j] = i1 < end ? hash[i1] : ~0ull;
    G.v1[j] = i1 < end ? value[i1] : 0.0;
    if (TWO) {
      G.h2[j] = i2 < end ? hash[i2] : ~0ull;
      G.v2[j] = i2 < end ? value[i2] : 0.0;
    }
  }
}

// TWO: two 64-line loads per spectrum and panel (dense spectra: up to 128 distinct lines fall into a panel); otherwise one
template <bool TWO>
__device__ __forceinline__ void dense_scatter(const TwisterView &tv, DenseGroup &G, int g, int wv, int lane, uint64_t &my_cur, uint64_t my_end,
                                              uint32_t k0, uint32_t k1, const uint64_t *__restrict__ hash, const double *__restrict__ value,
                                              double *A_s, double (&accp)[16], uint32_t &bad_mask) {
  constexpr uint32_t kFull = TWO ? 128u : 64u;
  // all the columns first: independent walks of the name -> row index, in flight together
  uint32_t ca[8], cb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ca[j] = G.h1[j] != ~0ull ? lookup_col(tv, G.h1[j]) : kNoCol;
    if (TWO) cb[j] = G.h2[j] != ~0ull ? lookup_col(tv, G.h2[j]) : kNoCol;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int sj = 8 * g + j, mm = 16 * wv + sj;
    uint64_t c = lane_u64(my_cur, sj);
    const uint64_t end = lane_u64(my_end, sj);
    uint64_t ha = G.h1[j], hb = TWO ? G.h2[j] : ~0ull;
    double va = G.v1[j], vb = TWO ? G.v2[j] : 0.0;
    uint32_t col_a = ca[j], col_b = TWO ? cb[j] : kNoCol;
    for (;;) {
      const uint64_t left = end - c;
      const uint32_t n_a = (uint32_t)min<uint64_t>(left, 64);
      const bool in_a = col_a != kNoCol && col_a < k1;  // (lines past `end` carry kNoCol)
      const uint64_t beyond_a = __ballot(col_a != kNoCol && col_a >= k1);
      const uint32_t f_a = beyond_a ? (uint32_t)__ffsll((long long)beyond_a) - 1u : 64u;
      const uint32_t take_a = min(f_a, n_a);
      // known lines must ascend: none of this panel (or an earlier one) after the first line of a later panel
      const uint64_t late_a = f_a < 64 ? (__ballot(in_a) >> f_a) : 0ull;
      uint32_t bad = (late_a != 0) || __ballot(col_a != kNoCol && col_a < k0) != 0;
      if ((uint32_t)lane < take_a && in_a) {
        atomicAdd(&A_s[mm * kDStride + (col_a - k0)], va);
        accp[sj] += va;
      }
      uint32_t take_b = 0;
      if (TWO && take_a == 64) {
        const uint32_t n_b = (uint32_t)min<uint64_t>(left - 64, 64);
        const bool in_b = col_b != kNoCol && col_b < k1;
        const uint64_t beyond_b = __ballot(col_b != kNoCol && col_b >= k1);
        const uint32_t f_b = beyond_b ? (uint32_t)__ffsll((long long)beyond_b) - 1u : 64u;
        take_b = min(f_b, n_b);
        const uint64_t late_b = f_b < 64 ? (__ballot(in_b) >> f_b) : 0ull;
        bad |= (late_b != 0) || __ballot(col_b != kNoCol && col_b < k0) != 0;
        if ((uint32_t)lane < take_b && in_b) {
          atomicAdd(&A_s[mm * kDStride + (col_b - k0)], vb);
          accp[sj] += vb;
        }
      }
      if (bad) bad_mask |= 1u << sj;
      c += take_a + take_b;
      if (take_a + take_b < kFull || c >= end) break;
      // every loaded line lay in the panel and the spectrum has more (repeated or unknown k-mers, or a single load at a
      // density it was not chosen for): keep going
      const uint64_t i1 = c + lane, i2 = i1 + 64;
      ha = i1 < end ? hash[i1] : ~0ull;
      va = i1 < end ? value[i1] : 0.0;
      col_a = i1 < end ? lookup_col(tv, ha) : kNoCol;
      if (TWO) {
        hb = i2 < end ? hash[i2] : ~0ull;
        vb = i2 < end ? value[i2] : 0.0;
        col_b = i2 < end ? lookup_col(tv, hb) : kNoCol;
      }
    }
    if (lane == sj) my_cur = c;
  }
}

template <int NB>
__global__ __launch_bounds__(256, (NB == 1 ? 2 : 1)) void twist_dense_fused_kernel(TwisterView tv, const uint64_t *__restrict__ hash,
                                                                                  const double *__restrict__ value,
                                                                                  const uint64_t *__restrict__ offsets, uint32_t n,
                                                                                  uint32_t tiles_m, uint32_t panels_per_slab, uint32_t n_panels,
                                                                                  double *__restrict__ slabs, double *__restrict__ acc_slabs,
                                                                                  int dbg) {
  extern __shared__ __attribute__((aligned(16))) double A_s[];  // [kDM][kDStride]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x % tiles_m, z = blockIdx.x / tiles_m;
  const uint32_t m0 = tile * kDM;
  const uint32_t p_lo = z * panels_per_slab, p_hi = min(n_panels, p_lo + panels_per_slab);
  // lane j < 16 of wave wv: cursor and end of spectrum m0 + 16 wv + j inside this slab
  uint64_t my_cur = 0, my_end = 0, lines = 0;
  if (lane < 16) {
    const uint32_t m = m0 + 16 * wv + lane;
    if (m < n) {
      const uint64_t lo = offsets[m], hi = offsets[m + 1];
      const uint64_t h_lo = first_hash_of_row(tv, (uint64_t)p_lo * kDK);
      uint64_t a = lo, b = hi;  // first line with hash >= h_lo
      while (a < b) {
        const uint64_t mid = (a + b) >> 1;
        if (hash[mid] < h_lo) a = mid + 1; else b = mid;
      }
      my_cur = a;
      my_end = hi;
      lines = hi - lo;
    }
  }
  // Two 64-line loads per spectrum and panel when the wave's spectra are dense (a panel then holds up to 128 of a
  // spectrum's lines), one when they are not: decided per wave from the lines its spectra have per panel on average.
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) lines += __shfl_xor((unsigned long long)lines, o, 64);
  const bool two = __builtin_amdgcn_readfirstlane((int)(lines / (16ull * n_panels) > 40)) != 0;
  f64x4 acc[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[nb][i] = f64x4{0.0, 0.0, 0.0, 0.0};
  double accp[16];  // this lane's share of acc (lib/Twister.ml:158) of the wave's sixteen spectra
#pragma unroll
  for (int j = 0; j < 16; ++j) accp[j] = 0.0;
  uint32_t bad_mask = 0;
  const uint32_t dcol = 16u * wv + (lane & 15);  // this lane's dimension inside a 64-wide block of dims
  for (uint32_t e = threadIdx.x; e < kDM * kDStride / 2; e += 256) reinterpret_cast<double2 *>(A_s)[e] = double2{0.0, 0.0};
  DenseGroup G0, G1;
  if (two) dense_issue<true>(G0, 0, my_cur, my_end, hash, value, lane);  // the first group's lines of the first panel
  else dense_issue<false>(G0, 0, my_cur, my_end, hash, value, lane);
  __syncthreads();
  for (uint32_t p = p_lo; p < p_hi; ++p) {
    const uint32_t k0 = p * kDK, k1 = (uint32_t)min<uint64_t>((uint64_t)k0 + kDK, tv.n_rows);
    // ---- build the panel: wave wv owns spectra 16 wv .. 16 wv + 15 in two groups of eight; the second group's lines are
    // requested before the first group's are used
    if (!(dbg & 1)) {
      if (two) {
        dense_issue<true>(G1, 1, my_cur, my_end, hash, value, lane);
        dense_scatter<true>(tv, G0, 0, wv, lane, my_cur, my_end, k0, k1, hash, value, A_s, accp, bad_mask);
        dense_scatter<true>(tv, G1, 1, wv, lane, my_cur, my_end, k0, k1, hash, value, A_s, accp, bad_mask);
      } else {
        dense_issue<false>(G1, 1, my_cur, my_end, hash, value, lane);
        dense_scatter<false>(tv, G0, 0, wv, lane, my_cur, my_end, k0, k1, hash, value, A_s, accp, bad_mask);
        dense_scatter<false>(tv, G1, 1, wv, lane, my_cur, my_end, k0, k1, hash, value, A_s, accp, bad_mask);
      }
    }
    __syncthreads();
    // the next panel's first group travels while this panel is multiplied
    if (p + 1 < p_hi && !(dbg & 1)) {
      if (two) dense_issue<true>(G0, 0, my_cur, my_end, hash, value, lane);
      else dense_issue<false>(G0, 0, my_cur, my_end, hash, value, lane);
    }
    // ---- multiply: wave wv owns dims 16 wv .. 16 wv + 15 of every 64-wide block, all 64 spectra (4 M tiles); B = the
    // twister's rows from L2, eight steps ahead
    if (!(dbg & 2))
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const uint32_t dc = 64u * nb + dcol;
        const bool dok = dc < tv.d_pad;
        double bb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const uint32_t row = k0 + 4 * q + (lane >> 4);
          bb[q] = (row < tv.n_rows && dok) ? tv.rows[(uint64_t)row * tv.d_pad + dc] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < kDK / 4; ++q) {
          const double b = bb[q & 7];
          if (q + 8 < kDK / 4) {
            const uint32_t row = k0 + 4 * (q + 8) + (lane >> 4);
            bb[q & 7] = (row < tv.n_rows && dok) ? tv.rows[(uint64_t)row * tv.d_pad + dc] : 0.0;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const double a = A_s[(16 * i + (lane & 15)) * kDStride + 4 * q + (lane >> 4)];
            acc[nb][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[nb][i], 0, 0, 0);
          }
        }
      }
    __syncthreads();  // the panel's readers are done
    if (p + 1 < p_hi) {
      for (uint32_t e = threadIdx.x; e < kDM * kDStride / 2; e += 256) reinterpret_cast<double2 *>(A_s)[e] = double2{0.0, 0.0};
      __syncthreads();
    }
  }
  // ---- this slab's partial sums
  const uint32_t ldo = NB * 64;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t m = m0 + 16 * i + (lane >> 4) + 4 * r;
        if (m < n) slabs[((uint64_t)z * n + m) * ldo + 64 * nb + dcol] = acc[nb][i][r];
      }
  // acc of the wave's spectra: the lanes' shares added up once, at the end
  const uint32_t bad_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)bad_mask);  // (set from wave-uniform conditions)
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double t = accp[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    const uint32_t m = m0 + 16 * wv + j;
    if (lane == 0 && m < n) acc_slabs[(uint64_t)z * n + m] = ((bad_all >> j) & 1u) ? __longlong_as_double(0x7FF8000000000000ll) : t;
  }
}

// out = (sum of the slabs, in slab order) / (sum of the slabs' acc) when normalising and acc <> 0
__global__ void twist_dense_combine_kernel(const double *__restrict__ slabs, const double *__restrict__ acc_slabs, uint32_t n_slabs, uint32_t n,
                                           uint32_t n_dims, uint32_t ldo, int normalize, double *__restrict__ out) {
  const uint64_t total = (uint64_t)n * n_dims, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const uint32_t m = (uint32_t)(e / n_dims), d = (uint32_t)(e % n_dims);
    double t = 0.0, a = 0.0;
    for (uint32_t z = 0; z < n_slabs; ++z) {
      t += slabs[((uint64_t)z * n + m) * ldo + d];
      a += acc_slabs[(uint64_t)z * n + m];
    }
    if (a != a) t = a;  // the lines of this spectrum did not ascend
    else if (normalize && a != 0.0) t = t / a;
    out[e] = t;
  }
}

// ---------------------------------------------------------------------------
// Sequences -> twisted rows with the counts handed over DENSE (round 3): for a twister of up to 36,864 k-mers (every
// canonical k-mer up to k = 8) a sequence's whole spectrum fits a block's LDS as one u32 counter per twister row, so
//   count_dense_kernel       one block per sequence: hash rolled over the windows, name -> row, ds_add_u32 into the LDS
//                            table, the table written out as one row of X (u32 [n_sequences][n_rows]) + acc = windows found;
//   twist_dense_counts_kernel 64 sequences x all dims x one slab of rows per block, panels of 128 rows: the panel's counts are
//                            loaded (4 B per element, coalesced, one panel ahead in registers), turned into x = c / acc
//                            (the reference's per-element division, lib/Twister.ml:177-178) as they are put into LDS, and
//                            multiplied by the twister's rows on the f64 matrix cores as in the kernel above.
// No CSR, no name -> row walk per LINE, no scatter, no zeroing: the image costs 4 bytes per (sequence, k-mer) once written
// and once read, against 16 bytes per line of a CSR spectrum (which at 97 % density is four times as much).
// ---------------------------------------------------------------------------
constexpr uint32_t kDenseCountRows = 36864;  // u32 counters per block: 144 KB of LDS

__global__ __launch_bounds__(1024) void count_dense_kernel(TwisterView tv, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets,
                                                           uint32_t n, int content, uint32_t *__restrict__ X, uint64_t ldx,
                                                           double *__restrict__ acc, uint32_t rsel_words) {
  extern __shared__ uint32_t s_tab[];
  __shared__ uint32_t s_found[16];
  const uint32_t r = blockIdx.x;
  if (r >= n) return;
  const uint32_t n_rows = (uint32_t)tv.n_rows;
  for (uint32_t q = threadIdx.x; q < n_rows; q += 1024) s_tab[q] = 0;
  // the name -> row index beside the counters when it fits (16 KB at k = 8): a window's lookup is then an LDS read, not a
  // trip to L1 / L2 (30 windows a thread, one lookup each: they were most of this kernel)
  RankWord *s_rsel = reinterpret_cast<RankWord *>(s_tab + ((n_rows + 3u) & ~3u));
  if (rsel_words)
    for (uint32_t q = threadIdx.x; q < rsel_words; q += 1024) s_rsel[q] = tv.rsel[q];
  TwisterView tl = tv;
  if (rsel_words) tl.rsel = s_rsel;
  __syncthreads();
  const uint64_t off = offsets[r], len = offsets[r + 1] - off;
  const int k = tv.hk;
  const uint64_t n_win = len >= (uint64_t)k ? len - k + 1 : 0;
  const uint8_t *seq = bases + off;
  const int shift = 2 * (k - 1);
  const uint32_t mask = (uint32_t)bits_mask(2 * k);
  // a thread rolls the hash over a run of consecutive windows (k - 1 + run base loads instead of k per window)
  const uint64_t per = (n_win + 1023) / 1024;
  const uint64_t a = (uint64_t)threadIdx.x * per, b = min(n_win, a + per);
  uint32_t found = 0;
  if (a < b) {
    uint32_t fwd = 0, rc = 0;
    int run = 0;
    for (int j = 0; j < k - 1; ++j) {
      const uint32_t c = base_code(seq[a + j]);
      fwd = ((fwd << 2) | (c & 3u)) & mask;
      rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
      run = c < 4u ? run + 1 : 0;
    }
    for (uint64_t w = a; w < b; ++w) {
      const uint32_t c = base_code(seq[w + k - 1]);
      fwd = ((fwd << 2) | (c & 3u)) & mask;
      rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
      run = c < 4u ? run + 1 : 0;
      if (run >= k) {
        const uint32_t col = lookup_col(tl, (uint64_t)((content == KPOP_DNA_DS && rc < fwd) ? rc : fwd));
        if (col != kNoCol) {
          atomicAdd(&s_tab[col], 1u);
          ++found;
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) found += (uint32_t)__shfl_xor((int)found, o, 64);
  if ((threadIdx.x & 63) == 0) s_found[threadIdx.x >> 6] = found;
  __syncthreads();
  uint32_t *xr = X + (uint64_t)r * ldx;
  for (uint32_t q = threadIdx.x; q < n_rows; q += 1024) xr[q] = s_tab[q];
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < 16; ++w) t += s_found[w];
    acc[r] = (double)t;  // exact: a count
  }
}

template <int NB>
__global__ __launch_bounds__(256, 2) void twist_dense_counts_kernel(TwisterView tv, const uint32_t *__restrict__ X, uint64_t ldx,
                                                                    const double *__restrict__ accv, uint32_t n, int normalize,
                                                                    uint32_t tiles_m, uint32_t panels_per_slab, uint32_t n_panels,
                                                                    double *__restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) double A_s[];  // [kDM][kDStride]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x % tiles_m, z = blockIdx.x / tiles_m;
  const uint32_t m0 = tile * kDM;
  const uint32_t p_lo = z * panels_per_slab, p_hi = min(n_panels, p_lo + panels_per_slab);
  // staging: thread t owns sequence m0 + t / 4 and the 32 columns (t % 4) * 32 .. + 32 of every panel
  const uint32_t sm = threadIdx.x >> 2, sc = (threadIdx.x & 3u) * 32u;
  const bool row_ok = m0 + sm < n;
  const uint32_t *xrow = X + (uint64_t)(row_ok ? m0 + sm : 0) * ldx;
  const double a_m = row_ok ? accv[m0 + sm] : 0.0;
  const bool norm = normalize && a_m != 0.0;
  const uint32_t n_rows = (uint32_t)tv.n_rows;
  uint4 cur[8];
  auto fetch = [&](uint32_t p) {
    const uint32_t c0 = p * kDK + sc;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t c = c0 + 4 * q;
      // (ldx is a multiple of 4 and the rows 16-byte aligned: whole uint4 loads; columns past n_rows are zero padding)
      cur[q] = (row_ok && c < ldx) ? *reinterpret_cast<const uint4 *>(xrow + c) : uint4{0, 0, 0, 0};
    }
  };
  f64x4 acc[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[nb][i] = f64x4{0.0, 0.0, 0.0, 0.0};
  const uint32_t dcol = 16u * wv + (lane & 15);
  if (p_lo < p_hi) fetch(p_lo);
  for (uint32_t p = p_lo; p < p_hi; ++p) {
    const uint32_t k0 = p * kDK;
    __syncthreads();  // the previous panel's readers are done
    {
      double *dst = A_s + sm * kDStride + sc;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t v[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double c = (double)v[e];
          dst[4 * q + e] = norm ? c / a_m : c;  // x_h = v_h / acc (lib/Twister.ml:177-178), element by element
        }
      }
    }
    __syncthreads();
    if (p + 1 < p_hi) fetch(p + 1);  // the next panel's counts travel under this panel's MFMAs
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const uint32_t dc = 64u * nb + dcol;
      const bool dok = dc < tv.d_pad;
      double bb[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t row = k0 + 4 * q + (lane >> 4);
        bb[q] = (row < n_rows && dok) ? tv.rows[(uint64_t)row * tv.d_pad + dc] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < kDK / 4; ++q) {
        const double b = bb[q & 7];
        if (q + 8 < kDK / 4) {
          const uint32_t row = k0 + 4 * (q + 8) + (lane >> 4);
          bb[q & 7] = (row < n_rows && dok) ? tv.rows[(uint64_t)row * tv.d_pad + dc] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const double a = A_s[(16 * i + (lane & 15)) * kDStride + 4 * q + (lane >> 4)];
          acc[nb][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[nb][i], 0, 0, 0);
        }
      }
    }
  }
  const uint32_t ldo = NB * 64;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t m = m0 + 16 * i + (lane >> 4) + 4 * r;
        if (m < n) slabs[((uint64_t)z * n + m) * ldo + 64 * nb + dcol] = acc[nb][i][r];
      }
}

// out = the slabs added in slab order (the division by acc was applied to the operands)
__global__ void twist_dense_sum_slabs_kernel(const double *__restrict__ slabs, uint32_t n_slabs, uint32_t n, uint32_t n_dims, uint32_t ldo,
                                             double *__restrict__ out) {
  const uint64_t total = (uint64_t)n * n_dims, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const uint32_t m = (uint32_t)(e / n_dims), d = (uint32_t)(e % n_dims);
    double t = 0.0;
    for (uint32_t z = 0; z < n_slabs; ++z) t += slabs[((uint64_t)z * n + m) * ldo + d];
    out[e] = t;
  }
}

constexpr uint32_t kDenseCountBatch = 4096;  // sequences per image: 4,096 x 32,896 x 4 B = 539 MB at k = 8

struct FusedPlan {
  uint32_t nb, tiles_m, n_panels, panels_per_slab, splits;
};
static FusedPlan fused_plan(const kpop_twister *tw, uint32_t n) {
  FusedPlan p;
  p.nb = div_up(tw->n_dims, 64);
  p.tiles_m = div_up(n, kDM);
  p.n_panels = std::max<uint32_t>(1, div_up(tw->n_rows, kDK));
  // two blocks a CU and a couple of rounds: enough slabs that every CU has work, few enough that a slab is many panels
  const uint32_t want = std::max(1u, (uint32_t)ctx().n_cus * 2 / std::max(1u, p.tiles_m));
  p.splits = std::min<uint32_t>(std::min<uint32_t>(want, p.n_panels), 64);
  p.panels_per_slab = div_up(p.n_panels, p.splits);
  p.splits = div_up(p.n_panels, p.panels_per_slab);
  return p;
}
static uint64_t fused_workspace_bytes(const kpop_twister *tw, uint32_t n_spectra) {
  const uint32_t n = std::min(n_spectra, kDenseBatch);
  uint64_t worst = 0;
  for (uint32_t t = n; t; t = (t > kDM) ? t / 2 : 0) {  // (the plan of the last, shorter batch may use more slabs)
    const FusedPlan p = fused_plan(tw, t);
    worst = std::max<uint64_t>(worst, (uint64_t)p.splits * t * (p.nb * 64 + 1) * 8);
  }
  return worst + 4096;
}

}  // namespace kpop

using namespace kpop;

extern "C" uint64_t kpop_dev_twist_dense_workspace_bytes(const kpop_twister *tw, uint32_t n_spectra) {
  if (!tw) return 0;
  const uint32_t rows = dense_tile_rows(tw, n_spectra);
  const uint32_t splits = dense_splits(rows, tw);
  const uint64_t image = (uint64_t)rows * std::max<uint64_t>(1, tw->n_rows) * 8 + (uint64_t)rows * 8 + (uint64_t)(splits + 1) * rows * tw->n_dims * 8 + 4096;
  return std::max(image, fused_workspace_bytes(tw, n_spectra));
}

// spectra whose lines ascend by hash (kpop_count_reads / kpop_dev_count_reads order): densified inside the contraction
extern "C" int kpop_dev_twist_dense_sorted(const kpop_twister *tw, const uint64_t *d_hash, const double *d_value, const uint64_t *d_offsets,
                                           uint32_t n_spectra, int normalize, void *d_work, double *d_out, void *stream) {
  KPOP_TRY(require_init());
  if (!tw || !d_offsets || !d_out || !d_work) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_twist_dense_sorted: null argument");
  if (n_spectra == 0) return KPOP_OK;
  if (tw->n_dims > 256)  // (four 64-wide blocks of accumulators is what the kernel is built for)
    return kpop_dev_twist_dense(tw, d_hash, d_value, d_offsets, n_spectra, normalize, d_work, d_out, stream);
  hipStream_t st = as_stream(stream);
  const TwisterView tv = view_of(tw);
  char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(d_work) + 255) & ~(uintptr_t)255);
  const size_t lds = (size_t)kDM * kDStride * 8;
  static PerSlotOnce once[4];
  for (uint32_t s0 = 0; s0 < n_spectra; s0 += kDenseBatch) {
    const uint32_t n = std::min(kDenseBatch, n_spectra - s0);
    const FusedPlan p = fused_plan(tw, n);
    double *slabs = reinterpret_cast<double *>(w);
    double *accs = slabs + (uint64_t)p.splits * n * p.nb * 64;
    const dim3 grid(p.tiles_m * p.splits), block(256);
#define KPOP_FUSED(NB)                                                                                                               \
  {                                                                                                                                  \
    if (!once[NB - 1]()) {                                                                                                           \
      KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&twist_dense_fused_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)lds));                                                                                       \
      once[NB - 1]() = true;                                                                                                         \
    }                                                                                                                                \
    twist_dense_fused_kernel<NB><<<grid, block, lds, st>>>(tv, d_hash, d_value, d_offsets + s0, n, p.tiles_m, p.panels_per_slab, p.n_panels, \
                                                           slabs, accs, ctx().tune_dbg >> 20);                                        \
  }
    switch (p.nb) {
      case 1: KPOP_FUSED(1) break;
      case 2: KPOP_FUSED(2) break;
      case 3: KPOP_FUSED(3) break;
      default: KPOP_FUSED(4) break;
    }
#undef KPOP_FUSED
    KPOP_LAUNCH_CHECK();
    twist_dense_combine_kernel<<<dim3(std::min<uint32_t>(div_up((uint64_t)n * tw->n_dims, 256), 4096)), dim3(256), 0, st>>>(
        slabs, accs, p.splits, n, tw->n_dims, p.nb * 64, normalize, d_out + (uint64_t)s0 * tw->n_dims);
    KPOP_LAUNCH_CHECK();
  }
  return KPOP_OK;
}

extern "C" int kpop_dev_twist_dense(const kpop_twister *tw, const uint64_t *d_hash, const double *d_value, const uint64_t *d_offsets,
                                    uint32_t n_spectra, int normalize, void *d_work, double *d_out, void *stream) {
  KPOP_TRY(require_init());
  if (!tw || !d_offsets || !d_out || !d_work) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_twist_dense: null argument");
  if (n_spectra == 0) return KPOP_OK;
  hipStream_t st = as_stream(stream);
  const TwisterView tv = view_of(tw);
  const uint64_t K = tw->n_rows;
  const uint32_t D = tw->n_dims, tile = dense_tile_rows(tw, n_spectra), splits = dense_splits(tile, tw);
  char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(d_work) + 255) & ~(uintptr_t)255);
  double *X = reinterpret_cast<double *>(w);
  double *acc = X + (uint64_t)tile * std::max<uint64_t>(1, K);
  double *C = acc + tile;                       // [tile][D]
  double *slabs = C + (uint64_t)tile * D;       // [splits][tile][D]
  for (uint32_t s0 = 0; s0 < n_spectra; s0 += tile) {
    const uint32_t nb = std::min(tile, n_spectra - s0);
    KPOP_HIP(hipMemsetAsync(X, 0, (uint64_t)nb * std::max<uint64_t>(1, K) * 8, st));
    dense_acc_kernel<<<dim3(div_up(nb, 4)), dim3(256), 0, st>>>(tv, d_hash, d_value, d_offsets, s0, nb, acc);
    KPOP_LAUNCH_CHECK();
    dense_fill_kernel<<<dim3(div_up(nb, 4)), dim3(256), 0, st>>>(tv, d_hash, d_value, d_offsets, s0, nb, normalize, acc, X, K);
    KPOP_LAUNCH_CHECK();
    if (K == 0) {
      KPOP_HIP(hipMemsetAsync(d_out + (uint64_t)s0 * D, 0, (uint64_t)nb * D * 8, st));
      continue;
    }
    // C[nb x D] = X[nb x K] * rows[K x d_pad]
    KPOP_TRY(gemm_f64<false>(X, K, tw->d_rows, tw->d_pad, d_out + (uint64_t)s0 * D, nb, D, K, splits, slabs, 0, st));
  }
  return KPOP_OK;
}

// ---------------------------------------------------------------------------
// sequences -> twisted rows through the dense image of their counts (see count_dense_kernel)
// ---------------------------------------------------------------------------
extern "C" uint64_t kpop_dev_count_twist_dense_workspace_bytes(const kpop_twister *tw, uint32_t n_reads) {
  if (!tw) return 0;
  const uint32_t n = std::min(n_reads, kDenseCountBatch);
  const uint64_t ldx = (tw->n_rows + 3) & ~3ull;
  uint64_t slabs = 0;
  for (uint32_t t = n; t; t = (t > kDM) ? t / 2 : 0) {
    const FusedPlan p = fused_plan(tw, t);
    slabs = std::max<uint64_t>(slabs, (uint64_t)p.splits * t * p.nb * 64 * 8);
  }
  return (uint64_t)n * ldx * 4 + (uint64_t)n * 8 + slabs + 8192;
}

extern "C" int kpop_dev_count_twist_dense(const kpop_twister *tw, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads,
                                          int content, int normalize, void *d_work, double *d_out, void *stream) {
  KPOP_TRY(require_init());
  if (!tw || !d_offsets || !d_out || !d_work) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_count_twist_dense: null argument");
  if (content != KPOP_DNA_DS && content != KPOP_DNA_SS) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_dev_count_twist_dense: DNA only");
  if (tw->n_rows > kDenseCountRows || tw->n_rows == 0 || tw->n_dims > 256 || (tw->hk ? tw->hk : tw->k) > 16)
    KPOP_FAIL(KPOP_ERR_UNSUPPORTED,
              "kpop_dev_count_twist_dense: a twister of %llu k-mers x %u dimensions (the dense image needs at most %u k-mers -- every "
              "canonical k-mer up to k = 8 -- and 256 dimensions); use kpop_dev_count_twist",
              (unsigned long long)tw->n_rows, tw->n_dims, kDenseCountRows);
  if (n_reads == 0) return KPOP_OK;
  hipStream_t st = as_stream(stream);
  const TwisterView tv = view_of(tw);
  const uint64_t ldx = (tw->n_rows + 3) & ~3ull;
  char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(d_work) + 255) & ~(uintptr_t)255);
  const uint32_t cap = std::min(n_reads, kDenseCountBatch);
  uint32_t *X = reinterpret_cast<uint32_t *>(w);
  double *accv = reinterpret_cast<double *>(w + (((uint64_t)cap * ldx * 4 + 255) & ~255ull));
  double *slabs = accv + ((cap + 31) & ~31u);
  // LDS of the counting kernel: the counters, and the rank-select index behind them when both fit 156 KB
  const uint64_t words = tw->d_rsel ? ((1ull << (2 * tw->k)) + 63) / 64 : 0;
  const size_t tab_bytes = (((size_t)tw->n_rows + 3) & ~(size_t)3) * 4;
  const uint32_t rsel_words = (words && tab_bytes + words * sizeof(RankWord) <= (156u << 10)) ? (uint32_t)words : 0u;
  const size_t lds_count = tab_bytes + (size_t)rsel_words * sizeof(RankWord), lds_mm = (size_t)kDM * kDStride * 8;
  static PerSlotOnce once_c, once_m[4];
  if (!once_c()) {
    KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&count_dense_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(156u << 10)));
    once_c() = true;
  }
  for (uint32_t s0 = 0; s0 < n_reads; s0 += kDenseCountBatch) {
    const uint32_t n = std::min(kDenseCountBatch, n_reads - s0);
    count_dense_kernel<<<dim3(n), dim3(1024), lds_count, st>>>(tv, d_bases, d_offsets + s0, n, content, X, ldx, accv, rsel_words);
    KPOP_LAUNCH_CHECK();
    const FusedPlan p = fused_plan(tw, n);
    const dim3 grid(p.tiles_m * p.splits), block(256);
#define KPOP_DC(NB)                                                                                                                  \
  {                                                                                                                                  \
    if (!once_m[NB - 1]()) {                                                                                                         \
      KPOP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&twist_dense_counts_kernel<NB>),                                   \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mm));                                        \
      once_m[NB - 1]() = true;                                                                                                       \
    }                                                                                                                                \
    twist_dense_counts_kernel<NB><<<grid, block, lds_mm, st>>>(tv, X, ldx, accv, n, normalize, p.tiles_m, p.panels_per_slab, p.n_panels, \
                                                              slabs);                                                                \
  }
    switch (p.nb) {
      case 1: KPOP_DC(1) break;
      case 2: KPOP_DC(2) break;
      case 3: KPOP_DC(3) break;
      default: KPOP_DC(4) break;
    }
#undef KPOP_DC
    KPOP_LAUNCH_CHECK();
    twist_dense_sum_slabs_kernel<<<dim3(std::min<uint32_t>(div_up((uint64_t)n * tw->n_dims, 256), 4096)), dim3(256), 0, st>>>(
        slabs, p.splits, n, tw->n_dims, p.nb * 64, d_out + (uint64_t)s0 * tw->n_dims);
    KPOP_LAUNCH_CHECK();
  }
  return KPOP_OK;
}

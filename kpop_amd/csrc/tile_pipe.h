// tile_pipe.h -- count_twist_tile_pipe_kernel: the consensus + residual scheme of count_twist_tile_kernel (count_twist.hip, which
// includes this file after the tile route's constants) as a two-stage software pipeline inside ONE persistent block per CU.
//
// Round 4's kernel ran its phases one after the other between block-wide barriers: the matrix cores were busy 0.35 of a chunk's
// time (0.17 of the launch) while bases were staged, rows looked up, the set built and the windows counted.  Here the block's
// sixteen wavefronts are two halves that never meet at a hardware barrier:
//   PRODUCER wavefronts 8..15 (two a SIMD) prepare chunk c + 1: the stretch's bases into LDS, the consensus set of the four seed
//     sequences' K-MER HASHES (ordered linear probing, as before -- but hashes, not rows: only the set's members and the windows
//     that MISS it are ever looked up in the twister's index, a few hundred look-ups a chunk instead of 32,768), the members
//     numbered in table order and their rows found, every window counted into X[64][768] (ONE BYTE a count: two chunks of X fit
//     the LDS; a stretch that holds one k-mer 256 times is left to the streaming kernel) or put on its wavefront's residual list,
//   CONSUMER wavefronts 0..7 (two a SIMD: wavefronts w and w + 4 share a SIMD, 16 dimensions and a half of the sequences each)
//     multiply chunk c: partial[64 x 64] = X[64 x U] * T_U on the f64 matrix cores, the rows of T four blocks of 16 ahead, and
//     gather the residual rows of eight sequences each under their MFMAs, as before.
// X, the set's rows and the lists are double-buffered; the halves hand chunks over through LDS counters (full / empty) and keep
// step among themselves with arrival counters in LDS (all wavefronts of a block are resident: spinning cannot deadlock).
// The sums: the set's rows in set order (a fixed interleaving of it: a lane group takes four consecutive members), then the
// residual rows in window order -- the same every run, equal to the streaming kernel's up to rounding.  Up to 64 dimensions
// (beyond: count_twist_tile_kernel).  lib/Twister.ml:146-188.
#pragma once
#include <type_traits>

namespace kpop {

constexpr uint32_t kPipeG = 64;                  // sequences a chunk
constexpr uint32_t kPipeXW = 194;                // dwords of a row of X: 768 one-byte counts + 8 (2 mod 32: a half-wavefront's 16 rows x 2 columns of a ds_read_b32 on 32 banks)
constexpr uint32_t kPipeAbsent = 0xFFFFFFFFu;    // a member's number when the twister has no row for it
constexpr uint32_t kPipeListCap = 8 * kTileS;    // entries of a producer wavefront's residual list: 8 sequences x 512 windows
// A sequence's stretch is staged as 2-BIT CODES, in 16-base units (33 of them: 512 windows + up to 14 bases more):
//   F[33]  the codes first base most significant (a window's forward hash is a funnel shift of two of these dwords),
//   R[33]  the COMPLEMENTS' codes first base least significant (its reverse complement's hash likewise),
//   V[17]  one bit a base, set where the byte was not one of ACGT acgt (or lies past the sequence's end): a window is a k-mer
//          when the k bits from its first base on are clear.
// Bytes were decoded window by window before (twelve instructions a base, in every window pass); now once per base, four
// bytes at a time, when the stretch is staged -- and a window's hash no longer rolls out of its predecessor's, so the windows that
// miss the set are hashed again one by one, not all 64 of a thread.
constexpr uint32_t kPipeUnits = 33, kPipeRowW = 85;  // dwords of a row: F at 0, R at 33, V at 66; 85 = 21 mod 32: four rows x eight threads' dwords on 32 banks
constexpr uint32_t kPipeMissLds = 192;  // entries of a producer wavefront's list of missed hashes that stay in LDS until their rows are found (beyond: global)
constexpr uint32_t kPipePF = 3;         // WIDE: blocks of 16 members the MFMA wavefronts ask for the rows of T ahead (a ring of kPipePF + 1 slots of four rows)
constexpr uint32_t kPipePrioList = 300;  // a consumer wavefront with more residual rows than this to gather goes ahead of the producers (s_setprio)
constexpr size_t kPipeLdsBytes = (size_t)2 * kPipeG * kPipeXW * 4 + (size_t)kTileH * 8 + (size_t)2 * kTileSetCap * 4 + (size_t)kPipeG * kPipeRowW * 4 +
                                 (size_t)8 * kPipeMissLds * 4;
typedef uint32_t pipe_u32x4_any __attribute__((ext_vector_type(4), aligned(1)));  // sixteen bytes from any address: one global_load_dwordx4

__device__ __forceinline__ uint32_t pipe_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// What the halves share is in LDS, and a wavefront's LDS operations complete in order: "everything I wrote is there" is
// s_waitcnt lgkmcnt(0).  The compiler's workgroup fences also wait for every GLOBAL load and store in flight (vmcnt(0)) -- here
// that is the look-ahead's loads of the next chunk's bases and the consumers' rows of T -- so they are used only where global
// memory carries data between wavefronts (the residual lists, at the hand-over).
__device__ __forceinline__ void pipe_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// spin until the LDS counter has reached `target`
__device__ __forceinline__ void pipe_wait(const uint32_t *p, uint32_t target) {
  while ((int32_t)(pipe_ld(p) - target) < 0) __builtin_amdgcn_s_sleep(1);
  pipe_lds_fence();
}
// a barrier among the eight wavefronts of a half: everybody adds one, everybody waits for eight more than last time
template <bool GLOBAL = false, uint32_t N = 8>
__device__ __forceinline__ void pipe_half_barrier(uint32_t *ctr, uint32_t &target, int lane) {
  if (GLOBAL)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  else
    pipe_lds_fence();
  target += N;
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  pipe_wait(ctr, target);
}

// WIDE (more than 64 dimensions): the work changes hands.  A chunk's MFMAs are D / 64 times the 64-dimension kernel's while its
// preparation is what it was, and its residual rows -- 2.3 MB a chunk from HBM at 256 dimensions and 0.3 % divergence, every load a DRAM
// round trip -- are as much time again at HBM's rate: THREE stages run at once in a block,
//   wavefronts 8..15  PRODUCERS: chunk c + 1's preparation, as up to 64 dimensions;
//   wavefronts 0..3   MFMA wavefronts, one a SIMD: chunk c, all 64 sequences x 16 columns a unit, the units of the twister's columns
//                     in turn against the same X; the sums go straight from the accumulators' registers to the sequences' slots;
//   wavefronts 4..7   GATHER wavefronts, one a SIMD: chunk c - 1's residual rows, sixteen 16-byte loads in flight each the whole time,
//                     their sums added to the slots once the chunk's MFMAs are done (lists and slots are kept three chunks deep).
// (Measured on the way: consumers gathering for their own columns under their MFMAs -- one in-order counter of loads, every four blocks of
// MFMAs waited for a DRAM round trip: 0.22 of the matrix peak; producers gathering before they hand a chunk over -- preparation and
// gather one after the other: 0.35-0.37, the gather alone 2.64, the MFMAs alone 2.57, both 3.28 ms on 5,000 mutants at 256 dimensions.)
// The chunks are dealt so that the blocks of one XCD work on the SAME stretch at a time: at 256 dimensions a stretch's members are
// 1.5 MB of rows (10 MB at 1,635) -- one stretch an L2, not eight.
template <bool ABLATE, bool WIDE = false>  // (ABLATE: the timing switches of kpop_tune("dbg", (1 | 2 | 4 | 8) << 24) are compiled in -- results are wrong under them)
__global__ __launch_bounds__(1024) void count_twist_tile_pipe_kernel(
    TwisterView tv, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets, int content,
    const uint32_t *__restrict__ nseg, const uint64_t *__restrict__ seg_off, double *__restrict__ partial,
    uint32_t *__restrict__ partial_cnt, const uint32_t *__restrict__ olong, const uint64_t *__restrict__ n_long_ptr,
    const uint32_t *__restrict__ gmax, const uint32_t *__restrict__ grel, uint32_t max_seg, uint32_t *__restrict__ slot_done,
    uint32_t *__restrict__ lists, int dbg_in) {
  constexpr uint32_t G = kPipeG, XW = kPipeXW;
  const int dbg = ABLATE ? dbg_in : (dbg_in & ~15);
  extern __shared__ __attribute__((aligned(16))) unsigned char pipe_lds[];
  uint32_t *Xw = reinterpret_cast<uint32_t *>(pipe_lds);          // [2][G][XW] four one-byte counts a word
  uint2 *ht = reinterpret_cast<uint2 *>(Xw + 2 * G * XW);         // [kTileH] {k-mer hash (kNoCol = empty), its number in the set}
  uint32_t *ucol = reinterpret_cast<uint32_t *>(ht + kTileH);     // [2][kTileSetCap] twister row of member u
  uint32_t *stage = ucol + 2 * kTileSetCap;                       // [G][kPipeRowW] the stretch's bases as 2-bit codes: F, R, V
  uint32_t *mlist = stage + G * kPipeRowW;                        // [8][kPipeMissLds] a producer wavefront's missed hashes, before their rows
  __shared__ uint64_t s_slot[3][G];   // the group's (sequence, segment) slots, ~0: the sequence has no such segment ([3]: WIDE keeps three chunks' worth)
  __shared__ uint32_t s_gcnt[3][G], s_gbeg[3][G];  // WIDE: a sequence's entries in its producer wavefront's residual list, and where they begin
  // WIDE: chunks every MFMA wavefront has multiplied / every gather wavefront has added the residual rows of -- ONE COUNTER A WAVEFRONT:
  // nothing keeps the four in step (wavefront 0 has a unit more than the others at 72 dimensions), and with one counter for all, three
  // of them two chunks ahead stood in for the fourth (the sub-batch test's 9,000 assemblies: columns 64..71 differed from call to call)
  __shared__ uint32_t s_mdone[4], s_gdone[4];
  __shared__ uint32_t s_rtot[2][8];   // entries of every producer wavefront's residual list (row | sequence of its eight << 29)
  __shared__ uint32_t s_U[2];         // members as multiplied (padded to 64)
  __shared__ uint32_t s_pbar, s_cbar4[2], s_full, s_empty2[2], s_done;  // (s_empty2: chunks released, counted per HALF of the consumers -- the halves are not in step any more, and one counter let a half that was two chunks ahead stand in for the other)
  __shared__ uint32_t s_new, s_samp, s_over, s_add[4], s_wbase[8];
  __shared__ uint32_t s_ref[kPipeRowW];    // the set's REFERENCE: the staged stretch of the primary seed it was built from ...
  __shared__ __attribute__((aligned(16))) uint16_t s_refnum[kTileS];    // ... the member every window of it is (0xFFFF none, 0xFFFE a member without a row),
  __shared__ __attribute__((aligned(16))) uint32_t s_xref[kPipeXW];     // ... its own row of X (what a sequence IDENTICAL to it counts: every row of X starts as a copy),
  __shared__ uint32_t s_refok[kTileS / 32], s_refabs[kTileS / 32];      // ... and a bit a window: its member has a row / is a member without one
  __shared__ uint32_t s_refbad;                                          // ... 1: a count of that row passed 255 (the set is of no use: its chunks are the streaming kernel's)
  __shared__ unsigned long long s_stamp[16];  // the phase clocks of this block (kpop_tune("dbg", 16 << 24)), added to g_tile_stamps at the end
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t n_long = (uint32_t)*n_long_ptr;
  if (n_long < kTileMinSeqs) return;  // (too few sequences to share anything: the streaming kernel's)
  if (threadIdx.x < 16) s_stamp[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    s_pbar = 0;
    s_cbar4[0] = s_cbar4[1] = 0;
    s_full = 0;
    s_empty2[0] = s_empty2[1] = 0;
    s_done = 0;
  }
  if (threadIdx.x < 4) s_mdone[threadIdx.x] = s_gdone[threadIdx.x] = 0;
  __syncthreads();  // the one hardware barrier: from here on the halves go their own ways
  const int k = tv.hk;
  const bool stamps = (dbg & 16) && lane == 0 && (wv == 0 || wv == 8);
  unsigned long long t_last = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  auto stamp = [&](int phase) {
    if (stamps) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      atomicAdd(&s_stamp[phase], now - t_last);  // (in LDS: a global atomic per phase was waited for at the next barrier's fence)
      t_last = now;
    }
  };
  if (wv >= 8) {
    // =================================================================== PRODUCER
    // Issue priority.  An f64 MFMA holds its SIMD for 64 cycles, and with two consumers queueing MFMAs on every SIMD the producers'
    // chain -- the launch, at low divergence -- waited its turn behind them: producers go ahead of consumers that only multiply
    // (5,000 mutants at 0.1 / 0.3 %: 1.17 -> 1.02, 1.44 -> 1.20 ms).  A consumer whose chunk carries a long residual list is the
    // launch instead (1 %, 3 %: producers first cost 3-4 % there) and raises itself above them for that chunk: below
    // (more than kPipePrioList entries for its eight sequences; 150 and 700 measured worse than 300 at 0.3 % and at 1 %).
    // kpop_tune("dbg", 64 << 24): everybody at the same priority, as before.
    if (!(dbg_in & 64)) __builtin_amdgcn_s_setprio(2);
    const uint32_t pw = (uint32_t)wv - 8u, pt = pw * 64u + (uint32_t)lane;
    const uint32_t sq = pt >> 3, tq = pt & 7u;  // the thread's sequence of the group and eighth of the stretch (64 windows)
    const uint32_t n_groups = (n_long + G - 1) / G;
    const uint64_t n_chunks = (uint64_t)n_groups * max_seg;
    const uint32_t mask = (uint32_t)bits_mask(2 * k), kmask = (1u << k) - 1u;
    const uint32_t rowq = tv.d_pad >> 4;  // a row of the twister in 128-byte units (d_pad is a multiple of 16 doubles)
    const uint32_t ssmask = content == KPOP_DNA_DS ? 0u : ~0u;  // (single-stranded: the forward hash whatever the other strand's)
    uint32_t pbar_t = 0, n_pub = 0;
    int misses = 0;
    uint32_t skip = 0, backoff = 8;
    auto pbar = [&]() {  // (under the phase clocks the time spent waiting here is its own entry, not the phase's)
      const unsigned long long t0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
      pipe_half_barrier(&s_pbar, pbar_t, lane);
      if (stamps) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0;
        atomicAdd(&s_stamp[6], dt);
        t_last += dt;
      }
    };
    auto set_slot = [](uint32_t h) { return (h * 2654435761u) >> 21; };  // 11 bits
    // ORDERED linear probing (the smaller key keeps the slot): the table, hence the numbering, hence the order of the additions
    // on the matrix cores, does not depend on which thread arrives when.  True: an empty slot was taken.
    auto insert = [&](uint32_t h) -> bool {
      uint32_t slot = set_slot(h);
#pragma unroll 1
      for (uint32_t t = 0; t < 2 * kTileH; ++t) {
        const uint32_t prev = atomicMin(&ht[slot].x, h);
        if (prev == h) return false;
        if (prev == kNoCol) return true;
        if (prev > h) h = prev;
        slot = (slot + 1) & (kTileH - 1);
      }
      return false;
    };
    auto find = [&](uint32_t h) -> uint2 {  // the k-mer's entry, or an empty one
      uint32_t slot = set_slot(h);
      uint2 e = ht[slot];
#pragma unroll 1
      for (uint32_t t = 0; t < kTileH && e.x != h && e.x != kNoCol; ++t) {
        slot = (slot + 1) & (kTileH - 1);
        e = ht[slot];
      }
      return e;
    };
    auto row_of = [&](uint4 q, uint32_t h) -> uint32_t {  // a rank-select word and a hash: the row, kNoCol when the twister has none
      const uint64_t bits = ((uint64_t)q.y << 32) | q.x;
      const uint32_t b = h & 63u;
      return ((bits >> b) & 1ull) ? q.z + (uint32_t)__popcll(bits & ((1ull << b) - 1ull)) : kNoCol;
    };
    // The next chunk's (sequence, offset, length, slot) and then its bases are loaded a chunk AHEAD, into registers: three dependent
    // trips to memory (list of long sequences -> offsets -> bases) that used to open every chunk.
    // chunks are dealt with the groups fastest: blocks running together work on one stretch of all sequences
    // A block takes a CONTIGUOUS range of the chunks (groups fastest within a stretch): its chunks in a row are the same stretch of
    // group after group of ONE organism -- the same consensus.  The set built for the first is kept for the following ones
    // (phases 1 to 3 below only when the stretch changes, or a chunk found less than three quarters of its windows in it): the
    // seeds hashed, the set built, its members numbered and their rows looked up were a third of a chunk's preparation.
    // (blocks go to the eight XCDs in turn: the ranges are dealt so that the blocks of ONE XCD hold neighbouring ranges -- a few
    // stretches' worth of twister rows per L2, not every stretch's)
    // WIDE: an XCD's blocks take the chunks of the XCD's range IN TURN (block j of it: chunks j, j + 32, ...): they multiply the same
    // stretch's members at the same time, unit after unit, out of one L2 -- and a block's chunks in a row are still one stretch's.
    const uint32_t vblock = gridDim.x % 8u == 0 ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
    const uint64_t per_block = (n_chunks + gridDim.x - 1) / gridDim.x;
    const bool by_xcd = WIDE && gridDim.x % 8u == 0;
    const uint64_t per_xcd = (n_chunks + 7) / 8;
    const uint32_t cstep = !WIDE ? 1u : by_xcd ? gridDim.x / 8u : gridDim.x;
    const uint64_t c_first = !WIDE ? (uint64_t)vblock * per_block : by_xcd ? (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u : (uint64_t)blockIdx.x;
    const uint64_t c_end = !WIDE ? min(n_chunks, ((uint64_t)vblock + 1) * per_block) : by_xcd ? min(n_chunks, (blockIdx.x % 8u + 1) * per_xcd) : n_chunks;
    auto next_chunk = [&](uint64_t from) -> uint64_t {  // the first chunk of the block's range at or after `from` that is this route's
      for (; from < c_end; from += cstep) {
        const uint32_t sg = (uint32_t)(from / n_groups), gp = (uint32_t)(from % n_groups);
        const uint32_t pg = (uint32_t)(((uint64_t)gp * G) / kTileProbeG);
        if (grel[pg] && sg < gmax[pg]) break;  // (one organism: tile_group_probe_kernel; and some sequence of the group is this long)
      }
      return from < c_end ? from : n_chunks;
    };
    struct Meta {
      uint32_t r;
      uint64_t off, len, slot0;
      uint32_t ns;
    };
    auto load_meta = [&](uint64_t c) -> Meta {
      Meta m{0u, 0ull, 0ull, 0ull, 0u};
      if (c < n_chunks) {
        const uint32_t li = (uint32_t)(c % n_groups) * G + sq;
        if (li < n_long) {
          m.r = olong[li];
          m.off = offsets[m.r];
          m.len = offsets[m.r + 1] - m.off;
          m.ns = nseg[m.r];
          m.slot0 = seg_off[m.r];
        }
      }
      return m;
    };
    // the thread's five 16-byte units of a stretch (units tq, tq + 8, ...: eight threads, one sequence), ALL asked for before any
    // is looked at.  A unit that the sequence ends in is loaded as the sequence's LAST sixteen bytes and shifted down when it is
    // stored; one past the end is not loaded (the stretch's first bytes stand in: nothing reads past the caller's buffer).
    auto load_units = [&](const Meta &m, uint64_t c, uint4 (&v)[5], int &avail) {
      const uint32_t sg = (uint32_t)(c / n_groups);
      const bool mine = c < n_chunks && sg < m.ns;
      const uint64_t s_beg = (uint64_t)sg * kTileS;
      const uint8_t *ga = bases + (mine ? m.off + s_beg : 0ull);
      avail = mine ? (int)min<uint64_t>(m.len - s_beg, (uint64_t)(kTileS + k - 1)) : 0;  // (a stretch of this route has >= 16 bases: its sequence has more than 512 windows)
#pragma unroll
      for (uint32_t r = 0; r < 5; ++r) {
        const int b0 = 16 * (int)(tq + 8u * r);
        const int at = b0 + 16 <= avail ? b0 : (b0 < avail ? avail - 16 : 0);
        const pipe_u32x4_any w = *reinterpret_cast<const pipe_u32x4_any *>(ga + at);
        v[r] = make_uint4(w.x, w.y, w.z, w.w);
      }
    };
    // four ASCII bytes -> their 2-bit codes one a byte (A0 C1 G2 T3, either case), and bit 7 of every byte that is none of them
    auto decode4 = [](uint32_t w, uint32_t &code, uint32_t &bad) {
      const uint32_t u = w & 0xDFDFDFDFu;                       // fold case
      const uint32_t x = (u >> 1) & 0x03030303u;                // A0 C1 T2 G3
      code = x ^ ((x >> 1) & 0x01010101u);                      // A0 C1 G2 T3
      const uint32_t back = __builtin_amdgcn_perm(0u, 0x54474341u, code);  // the letter every code stands for: "ACGT"[code]
      const uint32_t d = back ^ u;                              // a byte that is not zero: not that letter, i.e. none of the four
      bad = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;
    };
    // ... and into LDS, a chunk later, as codes
    auto store_units = [&](const uint4 (&v)[5], int avail) {
      uint32_t *row = stage + sq * kPipeRowW;
#pragma unroll
      for (uint32_t r = 0; r < 5; ++r) {
        const uint32_t u = tq + 8u * r;
        if (u < kPipeUnits) {
          const int b0 = 16 * (int)u;
          uint32_t w[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
          if (b0 + 16 > avail) {  // (the unit the sequence ends in, or one past it: rare)
            if (b0 >= avail) {
              w[0] = w[1] = w[2] = w[3] = 0u;
            } else {  // loaded as bytes [avail - 16, avail): down by sh = b0 + 16 - avail bytes, zeros behind
              const uint32_t sh = (uint32_t)(b0 + 16 - avail), ds = sh >> 2, bs = sh & 3u;
              uint32_t t[5];
#pragma unroll
              for (uint32_t j = 0; j < 5; ++j) {
                t[j] = 0u;
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) t[j] = (j + ds == q) ? w[q] : t[j];
              }
#pragma unroll
              for (uint32_t j = 0; j < 4; ++j) w[j] = __builtin_amdgcn_alignbyte(t[j + 1], t[j], bs);
            }
          }
          uint32_t L = 0, V = 0;  // codes first base least significant; a bit a base that is no base
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j) {
            uint32_t c, bad;
            decode4(w[j], c, bad);
            c |= c >> 6;                        // bytes 0,1 -> bits 0..3; bytes 2,3 -> bits 16..19
            c = (c | (c >> 12)) & 0xFFu;        // four codes, first base lowest
            L |= c << (8u * j);
            const uint32_t t = bad >> 7;        // bits 0, 8, 16, 24
            const uint32_t g = ((t | (t >> 7)) | ((t >> 14) | (t >> 21))) & 0xFu;
            V |= g << (4u * j);
          }
          uint32_t F = __builtin_bitreverse32(L);                           // first base highest, every code's two bits swapped ...
          F = ((F >> 1) & 0x55555555u) | ((F & 0x55555555u) << 1);          // ... and back
          row[u] = F;
          row[kPipeUnits + u] = ~L;
          reinterpret_cast<uint16_t *>(row + 2 * kPipeUnits)[u] = (uint16_t)V;
        }
      }
      if (tq == 0) reinterpret_cast<uint16_t *>(row + 2 * kPipeUnits)[kPipeUnits] = 0xFFFFu;  // (bases 528..543: none)
    };
    uint64_t chunk = next_chunk(c_first);
    uint32_t set_seg = ~0u, set_UP = 0, prim_seed = 0;  // the stretch the set in LDS was built for, its members as multiplied
    bool set_stale = true, rows_in[2] = {false, false};  // (rows_in: that buffer's copy of the members' rows is the set's)
    Meta cm = load_meta(chunk);
    uint4 cv[5];
    int cavail;
    load_units(cm, chunk, cv, cavail);
    while (chunk < n_chunks) {
      if (skip) {  // (a block that met four chunks in a row that shared too little skips ahead: the look-ahead starts again)
        --skip;
        chunk = next_chunk(chunk + cstep);
        cm = load_meta(chunk);
        load_units(cm, chunk, cv, cavail);
        continue;
      }
      const uint32_t seg = (uint32_t)(chunk / n_groups);
      const uint32_t buf = n_pub & 1u;
      const bool mine = seg < cm.ns;
      const uint64_t my_slot = mine ? cm.slot0 + seg : ~0ull;
      // ---- 0. the stretch's bases into LDS, the set cleared.  The barrier first: everybody is done with the last chunk's stage and set.
      pbar();
      store_units(cv, cavail);
      // the chunk after this one: its sequences now, its bases once those are known (after the seeds' turn)
      const uint64_t nchunk = next_chunk(chunk + cstep);
      const Meta nm = load_meta(nchunk);
      const bool rebuild = set_stale || seg != set_seg;  // (uniform)
      if (rebuild) {
#pragma unroll
        for (uint32_t q = 0; q < kTileH / 512; ++q) ht[pt + 512u * q] = make_uint2(kNoCol, kPipeAbsent);
        if (pt < kPipeXW) s_xref[pt] = 0u;
        if (pt == 0) s_refbad = 0u;
      }
      if (pt == 0) {
        s_new = 0;
        s_samp = 0;
        s_over = 0;
      }
      if (pt < 4) s_add[pt] = 0;
      pbar();
      stamp(0);  // the bases staged, the set cleared
      uint32_t UP = set_UP;
      if (rebuild) {
      // ---- 1. the seeds: window pt of each of the four, a thread (every step of the set's building is all 512 threads' then:
      // a seed a quarter of the threads left six wavefronts waiting at the barriers while two inserted)
      uint32_t sh4[4];
      {
        const uint32_t w = pt, j = w >> 4, o = w & 15u, jv = w >> 5;
#pragma unroll
        for (uint32_t sd = 0; sd < 4; ++sd) {
          const uint32_t *srow = stage + 16u * sd * kPipeRowW;
          const uint64_t T = ((uint64_t)srow[j] << 32) | srow[j + 1], U = ((uint64_t)srow[kPipeUnits + j + 1] << 32) | srow[kPipeUnits + j];
          const uint64_t VV = (((uint64_t)srow[2 * kPipeUnits + jv + 1] << 32) | srow[2 * kPipeUnits + jv]) >> (w & 31u);
          const uint32_t fwd = (uint32_t)(T >> (64u - 2u * o - 2u * (uint32_t)k)) & mask, rc = (uint32_t)(U >> (2u * o)) & mask;
          sh4[sd] = ((uint32_t)VV & kmask) == 0u ? min(fwd, rc | ssmask) : kNoCol;
        }
      }
      stamp(13);  // the seeds hashed
      // ---- 2. the consensus set: a PRIMARY seed's k-mers whole; then the other seeds, admitted in order while the set stays
      // within kTileSetCap members, by what each would add at most.  The primary is seed 0 -- unless every other seed finds fewer
      // than half of its k-mers there (sequence 0 of the group is the odd one out): then the set is started again from seed 1.
#pragma unroll 1
      for (uint32_t primary = 0; primary < 2; ++primary) {
        {
          const uint32_t hp = primary ? sh4[1] : sh4[0];
          const bool took = hp != kNoCol && insert(hp);
          const uint32_t n = (uint32_t)__popcll(__ballot(took));
          if (lane == 0 && n) atomicAdd(&s_new, n);
        }
        pbar();
        // the others, in their order of admission (1, 2, 3 after seed 0; 2, 3, 0 after seed 1): which of my three the set lacks
        uint32_t oh[3], em = 0;
#pragma unroll
        for (uint32_t q = 0; q < 3; ++q) oh[q] = primary ? (q == 0 ? sh4[2] : q == 1 ? sh4[3] : sh4[0]) : sh4[q + 1];
        {
          uint2 e[3];
#pragma unroll
          for (uint32_t q = 0; q < 3; ++q) e[q] = ht[set_slot(oh[q] != kNoCol ? oh[q] : 0u)];
#pragma unroll
          for (uint32_t q = 0; q < 3; ++q) {
            bool lacks = false;
            if (oh[q] != kNoCol) {
              uint2 ee = e[q];
              if (ee.x != oh[q] && ee.x != kNoCol) ee = find(oh[q]);  // (a first probe that met another k-mer: walk on)
              lacks = ee.x == kNoCol;
            }
            em |= lacks ? (1u << q) : 0u;
            const uint32_t nadd = (uint32_t)__popcll(__ballot(lacks)), nhas = (uint32_t)__popcll(__ballot(oh[q] != kNoCol));
            if (lane == 0 && nhas) atomicAdd(&s_add[q + 1], nadd | (nhas << 16));  // k-mers it would add | k-mers it has
          }
        }
        pbar();
        uint32_t total = s_new, in = 1u, strangers = 0, others = 0;  // in, bit pos: that seed is admitted
#pragma unroll
        for (uint32_t q = 1; q < 4; ++q) {
          const uint32_t add = s_add[q] & 0xFFFFu, has = s_add[q] >> 16;
          total += add;
          in |= (((in >> (q - 1)) & 1u) && total <= kTileSetCap) ? (1u << q) : 0u;
          others += has ? 1u : 0u;
          strangers += (has && add * 2u > has) ? 1u : 0u;
        }
        if (primary == 0 && others >= 2 && strangers == others) {  // (uniform) start again from seed 1
          pbar();
#pragma unroll
          for (uint32_t q = 0; q < kTileH / 512; ++q) ht[pt + 512u * q] = make_uint2(kNoCol, kPipeAbsent);
          if (pt == 0) s_new = 0;
          if (pt < 4) s_add[pt] = 0;
          pbar();
          continue;
        }
#pragma unroll
        for (uint32_t q = 0; q < 3; ++q)
          if (((em >> q) & 1u) && ((in >> (q + 1)) & 1u)) (void)insert(oh[q]);
        prim_seed = primary;
        break;
      }
      stamp(1);  // the set built
      }
      load_units(nm, nchunk, cv, cavail);  // (in flight under everything below)
      stamp(14);  // the next chunk's bases asked for (its sequences' offsets waited for)
      if (rebuild) pbar();
      // ---- 3. the members' rows (four slots of the table a thread: the only look-ups in the twister's index besides the misses),
      // the members that HAVE a row numbered in table order; then this chunk's X cleared (once the consumers are done with it)
      if (rebuild) {
        uint32_t kk[4], rows[4], occ = 0;
        uint4 q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) kk[i] = ht[4u * pt + i].x;
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const uint4 *>(tv.rsel + ((kk[i] != kNoCol ? kk[i] : 0u) >> 6));
        // (under the index words' trip to memory: the wait for this buffer)
        if (n_pub >= 2) {  // the consumers are done with this buffer's last chunk
          const unsigned long long t0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
          if constexpr (WIDE) {
#pragma unroll
            for (int w = 0; w < 4; ++w) pipe_wait(&s_mdone[w], n_pub - 1u);
          } else {
            pipe_wait(&s_empty2[0], 4u * (n_pub - 1u));
            pipe_wait(&s_empty2[1], 4u * (n_pub - 1u));
          }
          if (stamps) {
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0;
            atomicAdd(&s_stamp[7], dt);
            t_last += dt;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rows[i] = kk[i] != kNoCol ? row_of(q[i], kk[i]) : kNoCol;
          occ += rows[i] != kNoCol;
        }
        uint32_t incl = occ;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
          if (lane >= o) incl += up;
        }
        if (lane == 63) s_wbase[pw] = incl;
        pbar();
        uint32_t before = incl - occ, U = 0;
#pragma unroll
        for (uint32_t w = 0; w < 8; ++w) {
          const uint32_t n = s_wbase[w];
          before += w < pw ? n : 0u;
          U += n;
        }
        UP = (U + 63u) & ~63u;  // (<= kTileSetCap, a multiple of 64: the consumers take four blocks of 16 at a time)
        uint32_t *uc = ucol + buf * kTileSetCap;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (rows[i] != kNoCol) {
            ht[4u * pt + i].y = (before >> 2) | ((before & 3u) << 30);  // (the member's count: dword of the row | byte of it, as << 27 gives the shift)
            uc[before] = rows[i] * rowq;  // (the row's offset in the twister, in 128-byte units: the consumers add it as it is)
            ++before;
          }
        if (pt < UP - U) uc[U + pt] = 0;  // (padded with row 0 of the twister against zero counts)
        if (pt == 0) s_U[buf] = UP;
        set_seg = seg;
        set_UP = UP;
        set_stale = false;
        rows_in[buf] = true;
        rows_in[buf ^ 1u] = false;
        // the reference the window pass compares every sequence with: the primary seed's stretch, and its windows' members
        pbar();  // (the members' numbers are all written)
        {
          const uint32_t *prow = stage + 16u * prim_seed * kPipeRowW;
          if (pt < kPipeRowW) s_ref[pt] = prow[pt];
          const uint32_t w = pt, j = w >> 4, o = w & 15u, jv = w >> 5;
          const uint64_t T = ((uint64_t)prow[j] << 32) | prow[j + 1], U2 = ((uint64_t)prow[kPipeUnits + j + 1] << 32) | prow[kPipeUnits + j];
          const uint64_t VV = (((uint64_t)prow[2 * kPipeUnits + jv + 1] << 32) | prow[2 * kPipeUnits + jv]) >> (w & 31u);
          const uint32_t fwd = (uint32_t)(T >> (64u - 2u * o - 2u * (uint32_t)k)) & mask, rc = (uint32_t)(U2 >> (2u * o)) & mask;
          uint32_t num = 0xFFFFu;
          if (((uint32_t)VV & kmask) == 0u) {
            const uint2 e = find(min(fwd, rc | ssmask));
            if (e.x != kNoCol) num = e.y == kPipeAbsent ? 0xFFFEu : (((e.y & 0xFFFFu) << 2) | (e.y >> 30));
          }
          s_refnum[pt] = (uint16_t)num;
          if (num < 0xFFFEu) atomicAdd(&s_xref[num >> 2], 1u << (8u * (num & 3u)));
          const uint64_t okm = __ballot(num < 0xFFFEu), abm = __ballot(num == 0xFFFEu);
          if (lane == 0) {
            s_refok[2u * pw] = (uint32_t)okm;
            s_refok[2u * pw + 1u] = (uint32_t)(okm >> 32);
            s_refabs[2u * pw] = (uint32_t)abm;
            s_refabs[2u * pw + 1u] = (uint32_t)(abm >> 32);
          }
        }
        pbar();
        if (pw == 0) {  // a count of the reference's own row past 255: its bytes add up to less than the windows counted (see below)
          uint32_t sum = 0, cnt = 0;
          for (uint32_t i = (uint32_t)lane; i < 192u; i += 64u) sum = __builtin_amdgcn_sad_u8(s_xref[i], 0u, sum);
          if (lane < (int)(kTileS / 32)) cnt = (uint32_t)__popc(s_refok[lane]);
          int diff = (int)sum - (int)cnt;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) diff += __shfl_xor(diff, o, 64);
          if (lane == 0 && diff != 0) s_refbad = 1u;
        }
      } else {
        // the set stands: this buffer's X cleared (once the consumers are done with it), the members' rows copied over if it has
        // not had them yet (the other buffer has: its consumers only read it)
        if (n_pub >= 2) {
          const unsigned long long t0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
          if constexpr (WIDE) {
#pragma unroll
            for (int w = 0; w < 4; ++w) pipe_wait(&s_mdone[w], n_pub - 1u);
          } else {
            pipe_wait(&s_empty2[0], 4u * (n_pub - 1u));
            pipe_wait(&s_empty2[1], 4u * (n_pub - 1u));
          }
          if (stamps) {
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0;
            atomicAdd(&s_stamp[7], dt);
            t_last += dt;
          }
        }
        if (!rows_in[buf]) {
          const uint32_t *from = ucol + (buf ^ 1u) * kTileSetCap;
          uint32_t *to = ucol + buf * kTileSetCap;
          for (uint32_t u = pt; u < UP; u += 512) to[u] = from[u];
          rows_in[buf] = true;
        }
        if (pt == 0) s_U[buf] = UP;
      }
      // every sequence's row of X starts as the REFERENCE's: what a sequence identical to it counts.  The window pass then only
      // touches the windows that differ (a count taken back where the reference's k-mer is not there, the sequence's own added).
      {
        uint32_t *Xs = Xw + buf * G * XW + sq * XW + 24u * tq;
#pragma unroll
        for (uint32_t i = 0; i < 12; ++i) *reinterpret_cast<uint2 *>(Xs + 2u * i) = *reinterpret_cast<const uint2 *>(s_xref + 24u * tq + 2u * i);
      }
      pbar();
      stamp(2);  // the members' rows found and numbered, X set
      // ---- 4. every window against the set, eight at a time: a hit counts into X, a miss is remembered (a bit a window)
      uint64_t missm = 0;
      uint32_t found = 0;
      const uint32_t *row = stage + sq * kPipeRowW;
      // the hash of window w of the stretch (any w: the dwords come from LDS), kNoCol where there is no k-mer
      auto hash_at = [&](uint32_t w) -> uint32_t {
        const uint32_t j = w >> 4, o = w & 15u, jv = w >> 5;
        const uint64_t T = ((uint64_t)row[j] << 32) | row[j + 1], U = ((uint64_t)row[kPipeUnits + j + 1] << 32) | row[kPipeUnits + j];
        const uint64_t VV = (((uint64_t)row[2 * kPipeUnits + jv + 1] << 32) | row[2 * kPipeUnits + jv]) >> (w & 31u);
        const uint32_t fwd = (uint32_t)(T >> (64u - 2u * o - 2u * (uint32_t)k)) & mask, rc = (uint32_t)(U >> (2u * o)) & mask;
        return ((uint32_t)VV & kmask) == 0u ? min(fwd, rc | ssmask) : kNoCol;
      };
      {
        // Pass A, a few bit operations per 16 windows: which windows hold exactly the REFERENCE's k bases (the primary seed of the
        // set: assemblies of one organism are it almost everywhere)?  Those are already counted -- the row of X started as the
        // reference's.  A window that differs (around a substitution; everything after an insertion or deletion) takes the
        // reference's count back and goes the slow way: hashed, looked up in the set, counted or listed as a miss.
        // (X counts in single bytes; a count that passed 255 shows in the row's sum: below.)
        uint32_t valid = 0, absent = 0, sameok = 0;
        uint64_t slowm = 0, decm = 0;
        uint32_t *Xs = Xw + buf * G * XW + sq * XW;
        // a bit a base that differs from the reference's, 16 bases a dword of codes: the even bits of (x | x >> 1), pushed together
        auto differ16 = [](uint32_t x) -> uint32_t {
          x = (x | (x >> 1)) & 0x55555555u;
          x = (x | (x >> 1)) & 0x33333333u;
          x = (x | (x >> 2)) & 0x0F0F0F0Fu;
          x = (x | (x >> 4)) & 0x00FF00FFu;
          return (x | (x >> 8)) & 0xFFFFu;
        };
        // bit o of the result: any of bits o .. o + k - 1 of b set (o < 16, k <= 16)
        auto any_of_k = [&](uint32_t b) -> uint32_t {
          const uint32_t r2 = b | (b >> 1), r4 = r2 | (r2 >> 2), r8 = r4 | (r4 >> 4);  // any of 2, 4, 8 bits from o on
          return k >= 8 ? (r8 | (r8 >> (k - 8))) : k >= 4 ? (r4 | (r4 >> (k - 4))) : k >= 2 ? (r2 | (r2 >> (k - 2))) : b;  // (uniform; two runs that overlap)
        };
        uint32_t dif[5], vv[3], vr[3];
#pragma unroll
        for (uint32_t q = 0; q < 5; ++q) dif[q] = differ16(row[kPipeUnits + 4u * tq + q] ^ s_ref[kPipeUnits + 4u * tq + q]);
#pragma unroll
        for (uint32_t q = 0; q < 3; ++q) {
          vv[q] = row[2 * kPipeUnits + 2u * tq + q];
          vr[q] = s_ref[2 * kPipeUnits + 2u * tq + q];
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {  // sixteen windows a turn: bases 16 j .. 16 j + 31
          const uint32_t VV = (q & 1u) ? __builtin_amdgcn_alignbit(vv[(q >> 1) + 1], vv[q >> 1], 16) : vv[q >> 1];
          const uint32_t VR = (q & 1u) ? __builtin_amdgcn_alignbit(vr[(q >> 1) + 1], vr[q >> 1], 16) : vr[q >> 1];
          uint32_t ns = any_of_k(dif[q] | (dif[q + 1] << 16) | VV | VR) & 0xFFFFu;  // not the reference's k bases
          const uint32_t iv = any_of_k(VV) & 0xFFFFu;                                      // no k-mer of the sequence's at all
          if (dbg & 8) ns = 0xFFFFu;  // (a check of the shortcut: every window the slow way -- the results must not change)
          const uint32_t j = 4u * tq + q;
          const uint32_t ok16 = (s_refok[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu, ab16 = (s_refabs[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu;
          valid += (uint32_t)__popc(~iv & 0xFFFFu);
          sameok += (uint32_t)__popc(~ns & ok16);
          absent += (uint32_t)__popc(~ns & ab16);
          decm |= (uint64_t)(ns & ok16) << (16u * q);
          slowm |= (uint64_t)(ns & ~iv & 0xFFFFu) << (16u * q);
        }
        while (decm) {  // the reference's k-mer is not this sequence's here: its count taken back (it was copied in: never below zero)
          const uint32_t i = (uint32_t)__ffsll((long long)decm) - 1u;
          decm &= decm - 1ull;
          const uint32_t num = s_refnum[64u * tq + i];
          if (!(dbg & 2)) (void)__hip_atomic_fetch_sub(Xs + (num >> 2), 1u << (8u * (num & 3u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (!(dbg & 32)) stamp(15);  // pass A
        // Pass B: the slow windows, one by one (a thread that holds a substitution has about k of them)
        while (slowm) {
          const uint32_t i = (uint32_t)__ffsll((long long)slowm) - 1u;
          slowm &= slowm - 1ull;
          const uint32_t h = hash_at(64u * tq + i);
          const uint2 ee = find(h);
          if (ee.x == h) {
            if (ee.y != kPipeAbsent) {  // .y: the member's byte of X -- the dword's offset in the row | which byte << 30
              if (!(dbg & 2)) (void)__hip_atomic_fetch_add(Xs + (ee.y & 0xFFFFu), 1u << (ee.y >> 27), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else
              ++absent;
          } else
            missm |= 1ull << i;
        }
        const uint32_t nmiss = (uint32_t)__popcll(missm), hits = valid - nmiss;
        found = hits - absent;  // (= the reference's windows with a row that are this sequence's too + the slow windows that found one)
        (void)sameok;
        const uint32_t vh = valid | (hits << 16);
        if (vh) atomicAdd(&s_samp, vh);
      }
      pbar();
      {
        // a count that passed 255 carried into its neighbour (or out of its dword): the row's bytes then add up to 255 or 256 less
        // than the windows counted into it, per carry -- never more, so carries cannot cancel
        const uint32_t *Xs = Xw + buf * G * XW + sq * XW;
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t i = 0; i < 12; ++i) {  // 192 dwords of counts a row, eight threads: 24 each (a row starts on 8 bytes, not 16)
          const uint2 x = *reinterpret_cast<const uint2 *>(Xs + 24u * tq + 2u * i);
          sum = __builtin_amdgcn_sad_u8(x.x, 0u, sum);
          sum = __builtin_amdgcn_sad_u8(x.y, 0u, sum);
        }
        int diff = (int)sum - (int)found;  // (found: the windows this thread counted into X)
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) diff += __shfl_xor(diff, o, 8);
        if (diff != 0 && !(dbg & 2)) s_over = 1u;
      }
      pbar();
      stamp(3);  // the windows counted
      {
        // sequences that share little with the seeds (fewer than half the windows are of the consensus), or a k-mer 256 times
        // in one sequence's stretch: the chunk is left to the streaming kernel
        const uint32_t fh = s_samp;
        if ((dbg & 32) && pt == 0) {  // (development counts: chunks seen, with a count past 255, with an unusable reference, sharing too little, rebuilt)
          atomicAdd(&g_tile_stamps[8], 1ull);
          if (s_over) atomicAdd(&g_tile_stamps[9], 1ull);
          if (s_refbad) atomicAdd(&g_tile_stamps[10], 1ull);
          if ((fh >> 16) * 2u < (fh & 0xFFFFu)) atomicAdd(&g_tile_stamps[11], 1ull);
          if (rebuild) atomicAdd(&g_tile_stamps[12], 1ull);
        }
        if (s_over || s_refbad || (fh >> 16) * 2u < (fh & 0xFFFFu)) {
          if (++misses >= 4) {
            misses = 0;
            skip = backoff;
            backoff = min(backoff * 2u, 1u << 20);
          }
          set_stale = true;
          chunk = nchunk;
          cm = nm;
          continue;
        }
      }
      misses = 0;
      backoff = 8;
      {
        const uint32_t fh = s_samp;  // (fewer than three quarters of the windows in the set: the next chunk builds its own)
        if ((fh >> 16) * 4u < (fh & 0xFFFFu) * 3u) set_stale = true;
      }
      // ---- 5. the residual list of the wavefront's eight sequences: hashes in (sequence, window) order, then their rows
      {
        const uint32_t rcnt = (uint32_t)__popcll(missm);
        uint32_t incl = rcnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
          if (lane >= o) incl += up;
        }
        const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        // (WIDE: the lists and what goes with them are kept THREE chunks deep -- the gather wavefronts read a chunk's after its MFMAs)
        const uint32_t lb = WIDE ? n_pub % 3u : buf;
        if (WIDE && n_pub >= 3u) {
#pragma unroll
          for (int w = 0; w < 4; ++w) pipe_wait(&s_gdone[w], n_pub - 2u);
        }
        uint32_t *wl = lists + (((uint64_t)blockIdx.x * (WIDE ? 3 : 2) + lb) * 8 + pw) * kPipeListCap;
        uint32_t present = 0;  // entries that have a row, of MY sequence (lanes of a sequence all count it)
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) found += (uint32_t)__shfl_xor((int)found, o, 8);
        uint32_t wout = 0;  // entries that HAVE a row: the list as the consumers read it
        if (wtot) {  // (uniform)
          // (the hashes wait in LDS for their rows -- through global memory, written, fenced and read back, they cost two trips there;
          // a list longer than the LDS has room for goes on in global memory, as before)
          uint32_t *ml = mlist + pw * kPipeMissLds;
          {
            uint32_t pos = incl - rcnt;
            uint64_t mm = missm;
            while (mm) {  // (the thread's misses, in window order: each one hashed again by itself)
              const uint32_t i = (uint32_t)__ffsll((long long)mm) - 1u;
              mm &= mm - 1ull;
              const uint32_t h = hash_at(64u * tq + i);
              if (pos < kPipeMissLds) ml[pos] = h; else wl[pos] = h;
              ++pos;
            }
          }
          if (wtot > kPipeMissLds) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); else pipe_lds_fence();  // (uniform)
          stamp(4);  // the misses' hashes listed
          // the entries' rows, 64 at a time: the ones the twister has stay (compacted in place: a batch is read before anything is
          // written at or after it), tagged with their sequence of the eight (bits 29..31: a row of this route is below 2^29) and
          // counted for it
          uint32_t sbeg[8];  // where every sequence's entries begin (scalars)
#pragma unroll
          for (int j = 0; j < 8; ++j) sbeg[j] = (uint32_t)__builtin_amdgcn_readlane((int)(incl - rcnt), 8 * j);
          for (uint32_t p0 = 0; p0 < wtot; p0 += 64) {
            const uint32_t e = p0 + (uint32_t)lane;
            const uint32_t h = e >= wtot ? 0u : e < kPipeMissLds ? ml[e] : __hip_atomic_load(wl + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint4 q = *reinterpret_cast<const uint4 *>(tv.rsel + (h >> 6));
            const uint32_t row = e < wtot ? row_of(q, h) : kNoCol;
            uint32_t tag = 0;
#pragma unroll
            for (int j = 1; j < 8; ++j) tag += e >= sbeg[j] ? 1u : 0u;
            const bool has = row != kNoCol;
            const uint64_t pm = __ballot(has);
            if (has) wl[wout + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull))] = row | (tag << 29);
            wout += (uint32_t)__popcll(pm);
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
              const uint32_t n = (uint32_t)__popcll(__ballot(has && tag == j));
              present += ((uint32_t)lane >> 3) == j ? n : 0u;
            }
          }
          found += present;
        }
        if constexpr (WIDE) {
          // (the residual rows are the GATHER wavefronts': what they need of this chunk -- every sequence's entries and where they begin)
          uint32_t begv = 0;
          {
            uint32_t pre = 0;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
              begv = ((uint32_t)lane >> 3) == j ? pre : begv;
              pre += (uint32_t)__builtin_amdgcn_readlane((int)present, 8 * (int)j);
            }
          }
          if (tq == 0) {
            s_gcnt[lb][sq] = (dbg & 4) ? 0u : present;
            s_gbeg[lb][sq] = begv;
          }
        }
        if (tq == 0) {
          s_slot[lb][sq] = my_slot;
          if (mine) {
            slot_done[my_slot] = 1u;
            partial_cnt[my_slot] = found;
          }
        }
        if (lane == 0) s_rtot[buf][pw] = wout;
      }
      if ((dbg & 32) && pt == 0) {  // (bench.py's count of the matrix cores' work: chunks taken, members of their sets as multiplied)
        atomicAdd(&g_tile_stamps[14], 1ull);
        atomicAdd(&g_tile_stamps[15], (unsigned long long)UP);
      }
      pipe_half_barrier<true>(&s_pbar, pbar_t, lane);  // (the lists are in global memory: this barrier waits for the stores too)
      ++n_pub;
      if (pt == 0) __hip_atomic_store(&s_full, n_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (after the barrier: everybody's LDS and global writes are done)
      stamp(5);  // the misses' rows found, the chunk handed over
      chunk = nchunk;
      cm = nm;
    }
    pbar();
    if (pt == 0) __hip_atomic_store(&s_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (stamps) {
      for (int i = 0; i < 8; ++i) atomicAdd(&g_tile_stamps[i], s_stamp[i]);
      if (!(dbg & 32)) {
        atomicAdd(&g_tile_stamps[13], s_stamp[13]);
        atomicAdd(&g_tile_stamps[14], s_stamp[14]);
        atomicAdd(&g_tile_stamps[15], s_stamp[15]);
      }
    }
    return;
  }
  // ===================================================================== CONSUMER
  const uint32_t cw = (uint32_t)wv;
  const int ni = (int)(cw & 3u), mh = (int)(cw >> 2);  // 16 dimensions, 32 sequences (wavefronts cw and cw + 4 share a SIMD and their rows of T)
  const uint32_t g4 = (uint32_t)lane >> 4, c16 = (uint32_t)lane & 15u;
  if constexpr (WIDE) {
    // ------------------------------------------------------------------- the consumers of more than 64 dimensions
    const uint32_t nunits = tv.d_pad >> 4;  // the twister's columns in units of 16
    if (cw < 4u) {
      // ----------------------------------------------------------------- MFMA wavefronts: one a SIMD
      // A wavefront takes ALL 64 sequences (four accumulator tiles) x 16 columns a UNIT, units cw, cw + 4, ... : a piece of a member's
      // row is loaded once a block of the grid (with 32 sequences a wavefront, as up to 64 dimensions, two wavefronts loaded every piece),
      // and ONE wavefront keeps its SIMD's matrix pipe busy: four independent accumulators, the rows of T three blocks (48 MFMAs) ahead.
      uint32_t n_conw = 0;
      auto byte_f64w = [](uint32_t w, uint32_t j) -> double {  // byte j of w as a double (see below)
        const uint64_t bits = 0x4330000000000000ull | (uint64_t)((w >> (8u * j)) & 0xFFu);
        return __longlong_as_double((long long)bits) - 4503599627370496.0;
      };
      if (!(dbg_in & 64)) {  // (kpop_tune("pipeprio"); the producers: 2)
        switch ((dbg_in >> 8) & 3) {
          case 0: __builtin_amdgcn_s_setprio(0); break;
          case 1: __builtin_amdgcn_s_setprio(1); break;
          case 2: __builtin_amdgcn_s_setprio(2); break;
          default: __builtin_amdgcn_s_setprio(3); break;
        }
      }
      for (;;) {
        bool got = false;
        for (;;) {
          const uint32_t dn = pipe_ld(&s_done);
          pipe_lds_fence();
          const uint32_t fl = pipe_ld(&s_full);
          if ((int32_t)(fl - n_conw) > 0) {
            got = true;
            break;
          }
          if (dn) break;
          __builtin_amdgcn_s_sleep(2);
        }
        if (!got) break;
        pipe_lds_fence();
        stamp(8);  // waited for a chunk
        const uint32_t buf = n_conw & 1u, lb = n_conw % 3u;
        const uint32_t UP = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_U[buf]), nb = UP / 16u;
        const uint32_t *uc = ucol + buf * kTileSetCap + 4u * g4;
        const uint32_t *xa = Xw + buf * G * XW + c16 * XW + g4;  // tile t: 16 t rows on
        // The wavefront's units (cw, cw + 4, ...) x the set's blocks of 16 members are ONE run of blocks: the rows of T are asked for
        // kPipePF blocks ahead ACROSS the units' ends (a load of a member's row from an XCD's L2 under everybody's traffic takes some
        // 1.2 us -- three blocks of MFMAs, the distance up to 64 dimensions, covered half of it; and a prologue a unit was 6 % of a unit).
        const uint32_t n_my = cw < nunits ? (nunits - cw + 3u) / 4u : 0u;
        const uint32_t total = n_my * nb;
        uint32_t col = 16u * cw + c16;                   // the unit being multiplied: its column of this lane
        uint32_t bcur = 0;                               // ... the block of it
        uint32_t pb = 0, punit = cw;                     // the block being asked for, its unit
        const double *ptrow = tv.rows + min(col, tv.d_pad - 1u);
        f64x4 acc[4];
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
        double bs[kPipePF + 1][4];
        uint32_t a[4] = {0u, 0u, 0u, 0u};
        uint4 uqn = make_uint4(0u, 0u, 0u, 0u);
        auto ask = [&](double (&dst)[4]) {  // the rows of block pb of unit punit (uqn: its members), then on to the next block
          dst[0] = ptrow[(uint64_t)uqn.x << 4];
          dst[1] = ptrow[(uint64_t)uqn.y << 4];
          dst[2] = ptrow[(uint64_t)uqn.z << 4];
          dst[3] = ptrow[(uint64_t)uqn.w << 4];
          ++pb;
          const bool wrap = pb >= nb, more = punit + 4u < nunits;  // (uniform; past the last unit the last one's rows are asked for again)
          pb = wrap ? 0u : pb;
          ptrow += (wrap && more) ? 64 : 0;
          punit += (wrap && more) ? 4u : 0u;
          uqn = *reinterpret_cast<const uint4 *>(uc + 16u * pb);
        };
        if (total) {
          uqn = *reinterpret_cast<const uint4 *>(uc);
#pragma unroll
          for (uint32_t s = 0; s < kPipePF; ++s) ask(bs[s]);
#pragma unroll
          for (uint32_t t = 0; t < 4; ++t) a[t] = xa[16u * t * XW];
        }
        auto blocks4 = [&](auto phase) {  // four blocks, the ring's slots known at compile time (no branch inside: see the other consumers)
          constexpr uint32_t P = decltype(phase)::value;
#pragma unroll
          for (uint32_t s = 0; s < 4; ++s) {
            constexpr uint32_t R = kPipePF + 1u;
            const uint32_t slot = (P + s) % R, pslot = (P + s + kPipePF) % R;
            ask(bs[pslot]);
            const uint32_t bx = bcur + 1u >= nb ? 0u : bcur + 1u;
            uint32_t an[4];
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t) an[t] = xa[16u * t * XW + 4u * bx];
#pragma unroll
            for (uint32_t j = 0; j < 4 && !(dbg & 1); ++j) {
#pragma unroll
              for (uint32_t t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(byte_f64w(a[t], j), bs[slot][j], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t) a[t] = an[t];
            bcur = bx;
          }
          if (bcur == 0u) {  // (uniform) the unit's last block: its sums of the set's rows, straight from the registers (rows g4 + 4 rr of
                             // accumulator tile t = sequence 16 t + g4 + 4 rr); the gather wavefronts add the residual rows' to them
            if (col < tv.n_dims) {
#pragma unroll
              for (uint32_t q = 0; q < 16; ++q) {
                const uint64_t sl = s_slot[lb][16u * (q >> 2) + g4 + 4u * (q & 3u)];
                if (sl != ~0ull) partial[sl * tv.n_dims + col] = acc[q >> 2][q & 3u];
              }
            }
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
            col += 64u;
          }
        };
        static_assert((kPipePF + 1u) % 4u == 0u, "the ring of rows is whole groups of four blocks");
        {
          uint32_t L = 0;
#pragma unroll 1
          for (; L + (kPipePF + 1u) <= total; L += kPipePF + 1u) {
            blocks4(std::integral_constant<uint32_t, 0>{});
            if constexpr (kPipePF + 1u >= 8u) blocks4(std::integral_constant<uint32_t, 4>{});
          }
          if (L < total) blocks4(std::integral_constant<uint32_t, 0>{});  // (total is a multiple of four: nb is)
        }
        if (nb == 0u)  // (uniform) a set without a single row in the twister: the slots still start from zero
          for (uint32_t unit = cw; unit < nunits; unit += 4u) {
            const uint32_t c = 16u * unit + c16;
            if (c < tv.n_dims)
              for (uint32_t q = 0; q < 16; ++q) {
                const uint64_t sl = s_slot[lb][16u * (q >> 2) + g4 + 4u * (q & 3u)];
                if (sl != ~0ull) partial[sl * tv.n_dims + c] = 0.0;
              }
          }
        stamp(9);  // the matrix cores, every unit
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the sums are the gather wavefronts' to read)
        pipe_lds_fence();  // (done reading this buffer's X, rows and slots)
        ++n_conw;
        if (lane == 0) __hip_atomic_store(&s_mdone[cw], n_conw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        stamp(10);
      }
      if (stamps)
        for (int i = 8; i < 11; ++i) atomicAdd(&g_tile_stamps[i], s_stamp[i]);
      return;
    }
    // ------------------------------------------------------------------- GATHER wavefronts: one a SIMD
    // The residual rows -- 2.3 MB a chunk from HBM at 256 dimensions and 0.3 % divergence, every load a DRAM round trip -- of the 16
    // sequences 16 gq .. 16 gq + 15 of a chunk, once its MFMAs are done: every sequence's rows added in list (= window) order, lane = two
    // columns of a pass of 128, and the sums ADDED to what the MFMA wavefronts left in the sequence's slot.  SIXTEEN 16-byte loads are in
    // flight a wavefront THE WHOLE TIME: a ring of sixteen registers, slot u of revolution r + 1 asked for as soon as slot u of revolution
    // r is added.  A revolution is 16 / W entries of one sequence x W passes (W = up to four passes of 128 columns: one row number serves
    // W consecutive kilobytes of its row), the sequences' pass-groups one after the other; a sequence's 64 entries around the issuing
    // revolution sit in a register (lane = entry), the next sequence's are asked for a sequence ahead; the slot's sums are asked for
    // with a pass-group's last revolution and added when it is.
    {
      const uint32_t gq = cw - 4u;
      const uint32_t npass = (tv.d_pad + 127u) >> 7;
      uint32_t n_g = 0;
      if (!(dbg_in & 64)) __builtin_amdgcn_s_setprio(1);
      for (;;) {
        bool got = false;
        for (;;) {
          const uint32_t dn = pipe_ld(&s_done);
          pipe_lds_fence();
          const uint32_t fl = pipe_ld(&s_full);
          if ((int32_t)(fl - n_g) > 0) {
            got = true;
            break;
          }
          if (dn) break;
          __builtin_amdgcn_s_sleep(8);
        }
        if (!got) break;
#pragma unroll
        for (int w = 0; w < 4; ++w) pipe_wait(&s_mdone[w], n_g + 1u);  // the chunk's MFMAs are done, their sums in the slots
        const uint32_t lb = n_g % 3u;
        const uint32_t sqi = 16u * gq + ((uint32_t)lane & 15u);
        const uint32_t cntv = s_gcnt[lb][sqi], begv = s_gbeg[lb][sqi];  // lane j (< 16): sequence 16 gq + j
        const uint64_t slotv = s_slot[lb][sqi];
        const uint32_t slo = (uint32_t)slotv, shi = (uint32_t)(slotv >> 32);
        const uint32_t *lbase = lists + (((uint64_t)blockIdx.x * 3 + lb) * 8 + 2u * gq) * kPipeListCap;
        auto cnt_of = [&](uint32_t j) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)cntv, (int)j); };
        auto slot_of = [&](uint32_t j) -> uint64_t {
          return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)shi, (int)j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)slo, (int)j);
        };
        auto next_seq = [&](uint32_t j) -> uint32_t {  // the first sequence at or after j that has a slot and entries
          while (j < 16u && (cnt_of(j) == 0u || slot_of(j) == ~0ull)) ++j;
          return j;
        };
        auto load_ev = [&](uint32_t j, uint32_t blk) -> uint32_t {  // entries 64 blk .. 64 blk + 63 of sequence j, lane = entry
          const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)begv, (int)j);
          return __hip_atomic_load(lbase + (uint64_t)(j >> 3) * kPipeListCap + min(b + 64u * blk + (uint32_t)lane, kPipeListCap - 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        auto gather = [&](auto Wc) {
          constexpr uint32_t W = decltype(Wc)::value, E = 16u / W;
          const uint32_t ngrp = (npass + W - 1u) / W;
          uint32_t ij = next_seq(0);  // the ISSUING revolution: sequence, pass-group, first entry
          if (ij >= 16u) return;
          uint32_t ig = 0, ie0 = 0, icnt = cnt_of(ij);
          uint32_t ev = load_ev(ij, 0), evj = ij, evb = 0;  // the entries in the register: of sequence evj, block evb
          uint32_t pj = next_seq(ij + 1u);
          uint32_t evp = pj < 16u ? load_ev(pj, 0) : 0u;  // ... and of the next sequence, asked for a sequence ahead
          uint32_t cg = 0, ce0 = 0, ccnt = 0;  // the revolution IN FLIGHT (added next)
          uint64_t cslot = 0;
          bool cvalid = false;
          double2 v[16], acc[W], pv[W];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = make_double2(0.0, 0.0);
#pragma unroll
          for (uint32_t w = 0; w < W; ++w) acc[w] = pv[w] = make_double2(0.0, 0.0);
#pragma unroll 1
          for (;;) {
            const bool ivalid = ij < 16u;
            if (!ivalid && !cvalid) break;
            if (ivalid && (evj != ij || evb != (ie0 >> 6))) {  // (uniform) another sequence's entries, or a sequence's next 64
              if (ie0 == 0u && ij == pj) {
                ev = evp;
                pj = next_seq(ij + 1u);
                evp = pj < 16u ? load_ev(pj, 0) : 0u;
              } else
                ev = load_ev(ij, ie0 >> 6);
              evj = ij;
              evb = ie0 >> 6;
            }
            const double *gcol[W];
#pragma unroll
            for (uint32_t w = 0; w < W; ++w) gcol[w] = tv.rows + min(128u * (W * ig + w) + 2u * (uint32_t)lane, tv.d_pad - 2u);  // (d_pad is a multiple of 16: a pair of columns is inside or outside)
#pragma unroll
            for (uint32_t u = 0; u < 16; ++u) {
              const uint32_t eo = u / W, w = u % W;
              const bool cok = cvalid && ce0 + eo < ccnt;  // (uniform)
              acc[w].x = __dadd_rn(acc[w].x, cok ? v[u].x : 0.0);
              acc[w].y = __dadd_rn(acc[w].y, cok ? v[u].y : 0.0);
              const bool iok = ivalid && ie0 + eo < icnt;  // (uniform)
              const uint32_t rw = (uint32_t)__builtin_amdgcn_readlane((int)ev, (int)((ie0 + eo) & 63u)) & 0x1FFFFFFFu;
              v[u] = *reinterpret_cast<const double2 *>(gcol[w] + (uint64_t)(iok ? rw : 0u) * tv.d_pad);
            }
            if (cvalid && ce0 + E >= ccnt) {  // (uniform) the sequence's last entries of this pass-group: its sums, on top of the slot's
#pragma unroll
              for (uint32_t w = 0; w < W; ++w) {
                const uint32_t col = 128u * (W * cg + w) + 2u * (uint32_t)lane;
                double *dst = partial + cslot * tv.n_dims + col;
                if (col < tv.n_dims) dst[0] = __dadd_rn(pv[w].x, acc[w].x);
                if (col + 1u < tv.n_dims) dst[1] = __dadd_rn(pv[w].y, acc[w].y);
                acc[w] = make_double2(0.0, 0.0);
              }
            }
            if (ivalid && ie0 + E >= icnt) {  // (uniform) the slot's sums of the pass-group just asked for to its end
              const uint64_t islot = slot_of(ij);
#pragma unroll
              for (uint32_t w = 0; w < W; ++w) {
                const uint32_t col = 128u * (W * ig + w) + 2u * (uint32_t)lane;
                const double *src = partial + islot * tv.n_dims + col;
                pv[w].x = col < tv.n_dims ? src[0] : 0.0;
                pv[w].y = col + 1u < tv.n_dims ? src[1] : 0.0;
              }
            }
            cvalid = ivalid;
            cg = ig;
            ce0 = ie0;
            ccnt = icnt;
            if (ivalid) {
              cslot = slot_of(ij);
              ie0 += E;
              if (ie0 >= icnt) {
                ie0 = 0;
                if (++ig == ngrp) {
                  ig = 0;
                  ij = next_seq(ij + 1u);
                  if (ij < 16u) icnt = cnt_of(ij);
                }
              }
            }
          }
        };
        if (npass >= 3u) gather(std::integral_constant<uint32_t, 4>{});
        else if (npass == 2u) gather(std::integral_constant<uint32_t, 2>{});
        else gather(std::integral_constant<uint32_t, 1>{});
        pipe_lds_fence();  // (done with this chunk's lists and slots)
        ++n_g;
        if (lane == 0) __hip_atomic_store(&s_gdone[gq], n_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      return;
    }
  }
  const double *trow = tv.rows + min(16u * (uint32_t)ni + c16, tv.d_pad - 1);  // (columns past the twister's are not written below)
  const double *grow = tv.rows + min((uint32_t)lane, tv.n_dims - 1);
  uint32_t n_con = 0, cbar_t = 0;
  auto cbar = [&]() {
    const unsigned long long t0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    pipe_half_barrier<false, 4>(&s_cbar4[mh], cbar_t, lane);  // (the four wavefronts of one half of the sequences: see the sums below)
    if (stamps) {
      const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0;
      atomicAdd(&s_stamp[11], dt);
      t_last += dt;
    }
  };
  for (;;) {
    bool got = false;
    for (;;) {
      const uint32_t dn = pipe_ld(&s_done);  // (read before s_full: done is set after the last chunk's hand-over)
      pipe_lds_fence();
      const uint32_t fl = pipe_ld(&s_full);
      if ((int32_t)(fl - n_con) > 0) {
        got = true;
        break;
      }
      if (dn) break;
      __builtin_amdgcn_s_sleep(2);
    }
    if (!got) break;
    pipe_lds_fence();
    stamp(8);  // waited for a chunk
    const uint32_t buf = n_con & 1u;
    const uint32_t UP = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_U[buf]), nb = UP / 16u;
    const uint32_t *uc = ucol + buf * kTileSetCap + 4u * g4;
    const uint32_t *xa = Xw + buf * G * XW + (32u * (uint32_t)mh + c16) * XW + g4;  // M tile 0 (tile 1: 16 rows on)
    // ---- the residual gather, lane = dimension: eight rows loaded before two blocks' MFMAs, added after them, in list order
    const uint32_t *wl = lists + (((uint64_t)blockIdx.x * 2 + buf) * 8 + cw) * kPipeListCap;
    const uint32_t wcnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_rtot[buf][cw]);
    if (!(dbg_in & 64)) {
      if (wcnt > kPipePrioList) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);  // (see the producers' s_setprio)
    }
    constexpr int GR = 8;
    double rsum[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, cur = 0.0;
    uint32_t cur_j = 0, gpos = 0, ipos = 0;  // the next batch to add, the next batch to issue
    // The list's entries reach the wavefront eight at a time (every lane loads entry pos + lane % 8), TWO batches ahead of their
    // use, in two registers that take turns (A, B, A, ...): the counter of loads in flight is one and in order, so an entry load
    // the gather had to wait for at once (64 entries a time, refilled when they ran out) made it wait for every row of T and
    // every residual row issued before it.  Waiting for entries loaded two batches ago waits for nothing that is still needed.
    auto entries_at = [&](uint32_t pos) {
      return __hip_atomic_load(wl + min(pos + ((uint32_t)lane & 7u), kPipeListCap - 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    uint32_t entA = entries_at(0), entB = entries_at(GR);
    double gv[GR], gv2[GR];
    uint32_t gj[GR], gj2[GR];  // (scalars)
    auto stash = [&]() {
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) rsum[j] = cur_j == j ? cur : rsum[j];
    };
    // the batch at ipos: its GR rows on their way (row 0 past the end of the list), `ent` refilled for the batch after next
    auto gather_issue = [&](double (&v)[GR], uint32_t (&jj)[GR], uint32_t &ent) {
#pragma unroll
      for (int u = 0; u < GR; ++u) {
        const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)ent, u);  // (scalar)
        jj[u] = cj;
        v[u] = grow[(uint64_t)(ipos + (uint32_t)u < wcnt ? (cj & 0x1FFFFFFFu) : 0u) * tv.d_pad];
      }
      ent = entries_at(ipos + 2 * GR);
      ipos += GR;
    };
    // the batch at gpos added, in list order.  (Every row's use is OUTSIDE the branches: a load whose only use sits behind a
    // condition is moved down to it by the compiler, i.e. issued when it is needed instead of two blocks of MFMAs earlier.)
    auto gather_add = [&](const double (&v)[GR], const uint32_t (&jj)[GR]) {
#pragma unroll
      for (int u = 0; u < GR; ++u) {
        const bool ok = gpos + (uint32_t)u < wcnt;  // (uniform)
        const uint32_t j = jj[u] >> 29;
        if (ok && j != cur_j) {  // (the list goes sequence by sequence: at most seven changes)
          stash();
          cur = 0.0;
          cur_j = j;
        }
        cur = __dadd_rn(cur, ok ? v[u] : 0.0);
      }
      gpos += GR;
    };
    // ---- partial[64 x 64] = X[64 x UP] * T_U.  Block b is 16 members; lane group g4 takes members 16 b + 4 g4 + j in step j:
    // one 16-byte read gives its four rows of T, one dword its four counts of a sequence.  The rows of T are loaded three
    // blocks ahead (four register slots), the counts one block ahead.
    f64x4 acc0 = f64x4{0.0, 0.0, 0.0, 0.0}, acc1 = f64x4{0.0, 0.0, 0.0, 0.0};
    double bs[4][4];
    uint32_t a0 = 0, a1 = 0;
    uint4 uqn = make_uint4(0u, 0u, 0u, 0u);
    // byte j of w as a double: 2^52 + n has n in its low mantissa bits, and the subtraction is exact (one full-rate f64 add where
    // v_cvt_f64_u32 is a quarter-rate instruction, eight times per block of 16 members)
    auto byte_f64 = [](uint32_t w, uint32_t j) -> double {
      const uint64_t bits = 0x4330000000000000ull | (uint64_t)((w >> (8u * j)) & 0xFFu);
      return __longlong_as_double((long long)bits) - 4503599627370496.0;
    };
    auto load_rows = [&](double (&dst)[4], const uint4 u) {
      dst[0] = trow[(uint64_t)u.x << 4];  // (ucol holds the rows' offsets in 128-byte units: no 64-bit multiply per load)
      dst[1] = trow[(uint64_t)u.y << 4];
      dst[2] = trow[(uint64_t)u.z << 4];
      dst[3] = trow[(uint64_t)u.w << 4];
    };
    if (nb) {
#pragma unroll
      for (uint32_t s = 0; s < 3; ++s) load_rows(bs[s], *reinterpret_cast<const uint4 *>(uc + 16u * min(s, nb - 1u)));
      uqn = *reinterpret_cast<const uint4 *>(uc + 16u * min(3u, nb - 1u));
      a0 = xa[0];
      a1 = xa[16u * XW];
    }
    asm volatile("; the list's first entries are here" ::"v"(entA), "v"(entB));
    // (no branch inside the four blocks: the compiler counts the loads in flight exactly and waits for a block's rows of T only --
    // with the gather or a block behind a condition it assumed the shorter path and waited for the rows issued one block ago.
    // Past the end of the list a batch loads row 0 and adds nothing.)
    // Two copies of the loop: with the gather's batches while the list lasts, without them after it (at 0.1 % divergence a wavefront's
    // list is three batches and its set thirty blocks: the batches past the end -- row 0 loaded, nothing added -- were a tenth of the
    // loop's instructions).  A batch is issued and added inside one turn of four blocks, so the change-over leaves nothing in flight.
    auto four_blocks = [&](uint32_t b0, auto with_gather) {
      constexpr bool WG = decltype(with_gather)::value;
#pragma unroll
      for (uint32_t s = 0; s < 4; ++s) {
        const uint32_t b = b0 + s;
        if (WG && s == 0 && !(dbg & 4)) gather_issue(gv, gj, entA);
        if (WG && s == 2 && !(dbg & 4)) gather_issue(gv, gj, entB);
        load_rows(bs[(s + 3u) & 3u], uqn);  // block b + 3's rows of T
        uqn = *reinterpret_cast<const uint4 *>(uc + 16u * min(b + 4u, nb - 1u));
        const uint32_t bx = min(b + 1u, nb - 1u);
        const uint32_t an0 = xa[4u * bx], an1 = xa[16u * XW + 4u * bx];
#pragma unroll
        for (uint32_t j = 0; j < 4 && !(dbg & 1); ++j) {
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(byte_f64(a0, j), bs[s][j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(byte_f64(a1, j), bs[s][j], acc1, 0, 0, 0);
        }
        a0 = an0;
        a1 = an1;
        if (WG && (s & 1u) == 1 && !(dbg & 4)) gather_add(gv, gj);
      }
    };
    uint32_t b0 = 0;
    for (; b0 < nb && ipos < wcnt; b0 += 4) four_blocks(b0, std::true_type{});  // (nb is a multiple of four)
    for (; b0 < nb; b0 += 4) four_blocks(b0, std::false_type{});
    stamp(9);  // the matrix cores
    // The sums leave through X's room, [sequence][dimension].  A wavefront multiplies 32 sequences (its half mh) by 16 dimensions and
    // gathers for eight sequences of the SAME half: the four wavefronts of a half exchange among themselves only, in the rows of X
    // that are their half's (32 x 776 bytes for 32 x 512 of sums) -- a barrier of four, not of all eight consumers
    cbar();    // the half's wavefronts are done with its rows of X
    double *R = reinterpret_cast<double *>(Xw + buf * G * XW + 32u * (uint32_t)mh * XW);
    static_assert((32u * kPipeXW * 4u) % 8u == 0 && 32u * 64u * 8u <= 32u * kPipeXW * 4u, "a half's sums fit its own rows of X");
#pragma unroll
    for (uint32_t rr = 0; rr < 4; ++rr) {  // lane l holds rows (l >> 4) + 4 r of an M tile, column l & 15 of the wavefront's 16 dimensions
      R[(g4 + 4u * rr) * 64u + 16u * (uint32_t)ni + c16] = acc0[rr];
      R[(16u + g4 + 4u * rr) * 64u + 16u * (uint32_t)ni + c16] = acc1[rr];
    }
    // the wavefront's list (all of it after the MFMAs' share), then its eight sequences' sums
    // (two batches in flight: sixteen rows a wavefront)
    if (gpos < wcnt && !(dbg & 4)) {
      gather_issue(gv, gj, entA);
      while (gpos < wcnt) {
        gather_issue(gv2, gj2, entB);
        gather_add(gv, gj);
        gather_issue(gv, gj, entA);
        gather_add(gv2, gj2);
      }
    }
    stash();
    stamp(12);  // the rest of the list gathered
    cbar();
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
      const uint32_t sqn = 8u * cw + j;
      const uint64_t sl = s_slot[buf][sqn];
      const double v = __dadd_rn(R[(8u * (uint32_t)ni + j) * 64u + (uint32_t)lane], rsum[j]);
      if (sl != ~0ull && (uint32_t)lane < tv.n_dims) partial[sl * tv.n_dims + lane] = v;
    }
    pipe_lds_fence();  // (done reading this buffer's X, rows and slots)
    if (lane == 0) __hip_atomic_fetch_add(&s_empty2[mh], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ++n_con;
    stamp(10);  // the sums written
  }
  if (stamps)
    for (int i = 8; i < 13; ++i) atomicAdd(&g_tile_stamps[i], s_stamp[i]);  // (the consumers' entries)
}

}  // namespace kpop

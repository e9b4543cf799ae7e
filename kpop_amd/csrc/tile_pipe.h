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

namespace kpop {

constexpr uint32_t kPipeG = 64;                  // sequences a chunk
constexpr uint32_t kPipeXW = 193;                // dwords of a row of X: 768 one-byte counts + 4 (odd: 16 rows on 16 banks)
constexpr uint32_t kPipeAbsent = 0xFFFFFFFFu;    // a member's number when the twister has no row for it
constexpr uint32_t kPipeListCap = 8 * kTileS;    // entries of a producer wavefront's residual list: 8 sequences x 512 windows
constexpr size_t kPipeLdsBytes = (size_t)2 * kPipeG * kPipeXW * 4 + (size_t)kTileH * 8 + (size_t)2 * kTileSetCap * 4 + (size_t)kPipeG * kTileStageW * 4;

__device__ __forceinline__ uint32_t pipe_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// spin until the LDS counter has reached `target`
__device__ __forceinline__ void pipe_wait(const uint32_t *p, uint32_t target) {
  while ((int32_t)(pipe_ld(p) - target) < 0) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// a barrier among the eight wavefronts of a half: everybody adds one, everybody waits for eight more than last time
__device__ __forceinline__ void pipe_half_barrier(uint32_t *ctr, uint32_t &target, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  target += 8u;
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  pipe_wait(ctr, target);
}

__global__ __launch_bounds__(1024) void count_twist_tile_pipe_kernel(
    TwisterView tv, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets, int content,
    const uint32_t *__restrict__ nseg, const uint64_t *__restrict__ seg_off, double *__restrict__ partial,
    uint32_t *__restrict__ partial_cnt, const uint32_t *__restrict__ olong, const uint64_t *__restrict__ n_long_ptr,
    const uint32_t *__restrict__ gmax, const uint32_t *__restrict__ grel, uint32_t max_seg, uint32_t *__restrict__ slot_done,
    uint32_t *__restrict__ lists, int dbg) {
  constexpr uint32_t G = kPipeG, XW = kPipeXW;
  extern __shared__ __attribute__((aligned(16))) unsigned char pipe_lds[];
  uint32_t *Xw = reinterpret_cast<uint32_t *>(pipe_lds);          // [2][G][XW] four one-byte counts a word
  uint2 *ht = reinterpret_cast<uint2 *>(Xw + 2 * G * XW);         // [kTileH] {k-mer hash (kNoCol = empty), its number in the set}
  uint32_t *ucol = reinterpret_cast<uint32_t *>(ht + kTileH);     // [2][kTileSetCap] twister row of member u
  uint32_t *stage = ucol + 2 * kTileSetCap;                       // [G][kTileStageW] the stretch's bases
  __shared__ uint64_t s_slot[2][G];   // the group's (sequence, segment) slots, ~0: the sequence has no such segment
  __shared__ uint32_t s_rtot[2][8];   // entries of every producer wavefront's residual list (row | sequence of its eight << 29)
  __shared__ uint32_t s_U[2];         // members as multiplied (padded to 64)
  __shared__ uint32_t s_align[G];     // the stretch's address modulo 4, per sequence
  __shared__ uint32_t s_pbar, s_cbar, s_full, s_empty, s_done;
  __shared__ uint32_t s_new, s_samp, s_over, s_add[4], s_wbase[8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t n_long = (uint32_t)*n_long_ptr;
  if (n_long < kTileMinSeqs) return;  // (too few sequences to share anything: the streaming kernel's)
  if (threadIdx.x == 0) {
    s_pbar = 0;
    s_cbar = 0;
    s_full = 0;
    s_empty = 0;
    s_done = 0;
  }
  __syncthreads();  // the one hardware barrier: from here on the halves go their own ways
  const int k = tv.hk;
  const bool stamps = (dbg & 16) && lane == 0 && (wv == 0 || wv == 8);
  unsigned long long t_last = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  auto stamp = [&](int phase) {
    if (stamps) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      atomicAdd(&g_tile_stamps[phase], now - t_last);
      t_last = now;
    }
  };
  if (wv >= 8) {
    // =================================================================== PRODUCER
    const uint32_t pw = (uint32_t)wv - 8u, pt = pw * 64u + (uint32_t)lane;
    const uint32_t sq = pt >> 3, tq = pt & 7u;  // the thread's sequence of the group and eighth of the stretch (64 windows)
    const uint32_t myseed = pt >> 7;            // and the seed it hashes four windows of (sequences 0, 16, 32, 48)
    const uint32_t n_groups = (n_long + G - 1) / G;
    const uint64_t n_chunks = (uint64_t)n_groups * max_seg;
    const int shift = 2 * (k - 1);
    const uint32_t mask = (uint32_t)bits_mask(2 * k);
    uint32_t pbar_t = 0, n_pub = 0;
    int misses = 0;
    uint32_t skip = 0, backoff = 8;
    auto pbar = [&]() {  // (under the phase clocks the time spent waiting here is its own entry, not the phase's)
      const unsigned long long t0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
      pipe_half_barrier(&s_pbar, pbar_t, lane);
      if (stamps) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0;
        atomicAdd(&g_tile_stamps[6], dt);
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
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
      // chunks are dealt with the groups fastest: blocks running together work on one stretch of all sequences
      const uint32_t seg = (uint32_t)(chunk / n_groups), grp = (uint32_t)(chunk % n_groups);
      const uint32_t pg = (uint32_t)(((uint64_t)grp * G) / kTileProbeG);
      if (!grel[pg] || seg >= gmax[pg]) continue;  // (not one organism: tile_group_probe_kernel; none of the group's sequences is this long)
      if (skip) {
        --skip;
        continue;
      }
      const uint32_t buf = n_pub & 1u;
      const uint32_t li = grp * G + sq;
      uint32_t r = 0;
      uint64_t off = 0, len = 0;
      bool mine = false;
      if (li < n_long) {
        r = olong[li];
        off = offsets[r];
        len = offsets[r + 1] - off;
        mine = seg < nseg[r];
      }
      const uint64_t s_beg = (uint64_t)seg * kTileS;  // the stretch's first base (and window) in the sequence
      // ---- 0. the stretch's bases into LDS as aligned dwords (bytes past either end of the sequence are zeros: no window there),
      // the set cleared.  The barrier first: everybody is done with the last chunk's stage and set.
      pbar();
      {
        const uint8_t *ga = bases + off + s_beg;
        const uint32_t a = mine ? (uint32_t)(reinterpret_cast<uintptr_t>(ga) & 3u) : 0u;
        const int avail = mine ? (int)min<uint64_t>(len - s_beg, (uint64_t)(kTileS + k - 1)) : 0;
        uint32_t v[17];
#pragma unroll
        for (uint32_t j = 0; j < 17; ++j) {
          const int b0 = 4 * (int)(tq + 8u * j) - (int)a;
          v[j] = 0;
          if (b0 >= 0 && b0 + 4 <= avail)
            v[j] = *reinterpret_cast<const uint32_t *>(ga + b0);
          else
            for (int q = 0; q < 4; ++q)
              if (b0 + q >= 0 && b0 + q < avail) v[j] |= (uint32_t)ga[b0 + q] << (8 * q);
        }
#pragma unroll
        for (uint32_t j = 0; j < 17; ++j) stage[sq * kTileStageW + tq + 8u * j] = v[j];
        if (tq == 0) s_align[sq] = a;
      }
#pragma unroll
      for (uint32_t q = 0; q < kTileH / 512; ++q) ht[pt + 512u * q] = make_uint2(kNoCol, kPipeAbsent);
      if (pt == 0) {
        s_new = 0;
        s_samp = 0;
        s_over = 0;
      }
      if (pt < 4) s_add[pt] = 0;
      pbar();
      stamp(0);  // the bases staged, the set cleared
      // ---- 1. the seeds: four windows a thread, hashed out of LDS
      uint32_t sh4[4] = {kNoCol, kNoCol, kNoCol, kNoCol};
      {
        const uint32_t sseq = 16u * myseed, p0 = s_align[sseq] + 4u * (pt & 127u);
        const uint32_t *rowp = stage + sseq * kTileStageW + (p0 >> 2);
        uint32_t d[6], by[5];
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = rowp[i];
#pragma unroll
        for (int i = 0; i < 5; ++i) by[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], p0 & 3u);
        uint32_t fwd = 0, rc = 0;
        int run = 0;
#pragma unroll
        for (int j = 0; j < 18; ++j)
          if (j < k + 3) {  // (uniform)
            const uint32_t c = base_code((by[j >> 2] >> (8 * (j & 3))) & 0xFFu);
            fwd = ((fwd << 2) | (c & 3u)) & mask;
            rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
            run = c < 4u ? run + 1 : 0;
            const uint32_t hv = (run >= k) ? ((content == KPOP_DNA_DS && rc < fwd) ? rc : fwd) : kNoCol;
            const int i = j - (k - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (i == q) sh4[q] = hv;
          }
      }
      // ---- 2. the consensus set: a PRIMARY seed's k-mers whole; then the other seeds, admitted in order while the set stays
      // within kTileSetCap members, by what each would add at most.  The primary is seed 0 -- unless every other seed finds fewer
      // than half of its k-mers there (sequence 0 of the group is the odd one out): then the set is started again from seed 1.
#pragma unroll 1
      for (uint32_t primary = 0; primary < 2; ++primary) {
        if (myseed == primary) {
          uint32_t took = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (sh4[i] != kNoCol) took += insert(sh4[i]) ? 1u : 0u;
          if (took) atomicAdd(&s_new, took);
        }
        pbar();
        const uint32_t pos = (myseed + 4u - primary) & 3u;  // the seed's place in the order of admission (0: the primary)
        uint32_t em = 0;                                    // my k-mers the primary's set lacks
        if (pos) {
          uint32_t nadd = 0, nhas = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (sh4[i] != kNoCol) {
              ++nhas;
              if (find(sh4[i]).x == kNoCol) {
                em |= 1u << i;
                ++nadd;
              }
            }
          if (nhas) atomicAdd(&s_add[pos], nadd | (nhas << 16));  // k-mers it would add | k-mers it has
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
        if (pos && ((in >> pos) & 1u)) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if ((em >> i) & 1u) (void)insert(sh4[i]);
        }
        break;
      }
      pbar();
      stamp(1);  // the set built
      // ---- 3. the members' rows (four slots of the table a thread: the only look-ups in the twister's index besides the misses),
      // the members that HAVE a row numbered in table order; then this chunk's X cleared (once the consumers are done with it)
      uint32_t UP;
      {
        uint32_t kk[4], rows[4], occ = 0;
        uint4 q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) kk[i] = ht[4u * pt + i].x;
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const uint4 *>(tv.rsel + ((kk[i] != kNoCol ? kk[i] : 0u) >> 6));
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
        if (n_pub >= 2) {  // the consumers are done with this buffer's last chunk
          const unsigned long long t0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
          pipe_wait(&s_empty, 8u * (n_pub - 1u));
          if (stamps) {
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0;
            atomicAdd(&g_tile_stamps[7], dt);
            t_last += dt;
          }
        }
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
            ht[4u * pt + i].y = before;
            uc[before] = rows[i];
            ++before;
          }
        if (pt < UP - U) uc[U + pt] = 0;  // (padded with row 0 of the twister against zero counts)
        if (pt == 0) s_U[buf] = UP;
        uint4 *X4 = reinterpret_cast<uint4 *>(Xw + buf * G * XW);
        for (uint32_t qq = pt; qq < G * XW / 4; qq += 512) X4[qq] = make_uint4(0u, 0u, 0u, 0u);
      }
      pbar();
      stamp(2);  // the members' rows found and numbered, X cleared
      // ---- 4. every window against the set, eight at a time: a hit counts into X, a miss is remembered (a bit a window)
      uint64_t missm = 0;
      uint32_t found = 0;
      const uint32_t a = s_align[sq];
      const uint32_t *sg = stage + sq * kTileStageW + tq * 16u;
      const uint32_t sh = a + (uint32_t)(k - 1);
      const uint32_t *sm = sg + (sh >> 2);
      auto warm = [&](uint32_t &fwd, uint32_t &rc, int &run) {  // the k - 1 bases before the first window's last base
        uint32_t pw4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pw4[i] = __builtin_amdgcn_alignbyte(sg[i + 1], sg[i], a);
        fwd = 0;
        rc = 0;
        run = 0;
#pragma unroll
        for (int j = 0; j < 14; ++j)
          if (j < k - 1) {  // (uniform)
            const uint32_t c = base_code((pw4[j >> 2] >> (8 * (j & 3))) & 0xFFu);
            fwd = ((fwd << 2) | (c & 3u)) & mask;
            rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
            run = c < 4u ? run + 1 : 0;
          }
      };
      // the hashes of windows 8 b .. 8 b + 7 of the thread (rolled on from the state), and which of them are k-mers
      auto batch = [&](uint32_t b, uint32_t &fwd, uint32_t &rc, int &run, uint32_t (&h)[8]) -> uint32_t {
        const uint32_t m0 = __builtin_amdgcn_alignbyte(sm[2 * b + 1], sm[2 * b], sh & 3u), m1 = __builtin_amdgcn_alignbyte(sm[2 * b + 2], sm[2 * b + 1], sh & 3u);
        uint32_t vm = 0;
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) {
          const uint32_t c = base_code(((i < 4 ? m0 : m1) >> (8u * (i & 3u))) & 0xFFu);
          fwd = ((fwd << 2) | (c & 3u)) & mask;
          rc = (rc >> 2) | ((3u - (c & 3u)) << shift);
          run = c < 4u ? run + 1 : 0;
          h[i] = (content == KPOP_DNA_DS && rc < fwd) ? rc : fwd;
          vm |= run >= k ? (1u << i) : 0u;
        }
        return vm;
      };
      {
        uint32_t fwd, rc, valid = 0, hits = 0, over = 0;
        int run;
        warm(fwd, rc, run);
        uint32_t *Xs = Xw + buf * G * XW + sq * XW;
#pragma unroll 1
        for (uint32_t b = 0; b < 8; ++b) {
          uint32_t h[8];
          const uint32_t vm = batch(b, fwd, rc, run, h);
          uint2 e[8];
#pragma unroll
          for (uint32_t i = 0; i < 8; ++i) e[i] = ht[set_slot(h[i])];
#pragma unroll
          for (uint32_t i = 0; i < 8; ++i)
            if ((vm >> i) & 1u) {
              uint2 ee = e[i];
              if (ee.x != h[i] && ee.x != kNoCol) {  // (a first probe that met another k-mer: walk on)
                uint32_t slot = set_slot(h[i]);
#pragma unroll 1
                for (uint32_t t = 0; t < kTileH && ee.x != h[i] && ee.x != kNoCol; ++t) {
                  slot = (slot + 1) & (kTileH - 1);
                  ee = ht[slot];
                }
              }
              ++valid;
              if (ee.x == h[i]) {
                ++hits;
                if (ee.y != kPipeAbsent) {
                  const uint32_t sft = 8u * (ee.y & 3u);
                  const uint32_t old = atomicAdd(&Xs[ee.y >> 2], 1u << sft);
                  over |= ((old >> sft) & 0xFFu) == 0xFFu ? 1u : 0u;
                  ++found;
                }
              } else
                missm |= 1ull << (8u * b + i);
            }
        }
        const uint32_t vh = valid | (hits << 16);
        if (vh) atomicAdd(&s_samp, vh);
        if (over) s_over = 1u;
      }
      pbar();
      stamp(3);  // the windows counted
      {
        // sequences that share little with the seeds (fewer than half the windows are of the consensus), or a k-mer 256 times
        // in one sequence's stretch: the chunk is left to the streaming kernel
        const uint32_t fh = s_samp;
        if (s_over || (fh >> 16) * 2u < (fh & 0xFFFFu)) {
          if (++misses >= 4) {
            misses = 0;
            skip = backoff;
            backoff = min(backoff * 2u, 1u << 20);
          }
          continue;
        }
      }
      misses = 0;
      backoff = 8;
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
        uint32_t *wl = lists + (((uint64_t)blockIdx.x * 2 + buf) * 8 + pw) * kPipeListCap;
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) found += (uint32_t)__shfl_xor((int)found, o, 8);
        uint32_t wout = 0;  // entries that HAVE a row: the list as the consumers read it
        if (wtot) {  // (uniform)
          {
            uint32_t fwd, rc;
            int run;
            warm(fwd, rc, run);
            uint32_t pos = incl - rcnt;
#pragma unroll 1
            for (uint32_t b = 0; b < 8; ++b) {
              uint32_t h[8];
              (void)batch(b, fwd, rc, run, h);
              const uint32_t mb = (uint32_t)(missm >> (8u * b)) & 0xFFu;
#pragma unroll
              for (uint32_t i = 0; i < 8; ++i)
                if ((mb >> i) & 1u) wl[pos++] = h[i];
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          stamp(4);  // the misses' hashes listed
          // the entries' rows, 64 at a time: the ones the twister has stay (compacted in place: a batch is read before anything is
          // written at or after it), tagged with their sequence of the eight (bits 29..31: a row of this route is below 2^29) and
          // counted for it
          uint32_t sbeg[8];  // where every sequence's entries begin (scalars)
#pragma unroll
          for (int j = 0; j < 8; ++j) sbeg[j] = (uint32_t)__builtin_amdgcn_readlane((int)(incl - rcnt), 8 * j);
          uint32_t present = 0;  // of MY sequence (lanes of a sequence all count it)
          for (uint32_t p0 = 0; p0 < wtot; p0 += 64) {
            const uint32_t e = p0 + (uint32_t)lane;
            const uint32_t h = e < wtot ? __hip_atomic_load(wl + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
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
        if (tq == 0) {
          const uint64_t slot = mine ? seg_off[r] + seg : ~0ull;
          s_slot[buf][sq] = slot;
          if (mine) {
            slot_done[slot] = 1u;
            partial_cnt[slot] = found;
          }
        }
        if (lane == 0) s_rtot[buf][pw] = wout;
      }
      if ((dbg & 32) && pt == 0) {  // (bench.py's count of the matrix cores' work: chunks taken, members of their sets as multiplied)
        atomicAdd(&g_tile_stamps[14], 1ull);
        atomicAdd(&g_tile_stamps[15], (unsigned long long)UP);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      pbar();
      ++n_pub;
      if (pt == 0) __hip_atomic_store(&s_full, n_pub, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      stamp(5);  // the misses' rows found, the chunk handed over
    }
    pbar();
    if (pt == 0) __hip_atomic_store(&s_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }
  // ===================================================================== CONSUMER
  const uint32_t cw = (uint32_t)wv;
  const int ni = (int)(cw & 3u), mh = (int)(cw >> 2);  // 16 dimensions, 32 sequences (wavefronts cw and cw + 4 share a SIMD and their rows of T)
  const uint32_t g4 = (uint32_t)lane >> 4, c16 = (uint32_t)lane & 15u;
  const double *trow = tv.rows + min(16u * (uint32_t)ni + c16, tv.d_pad - 1);  // (columns past the twister's are not written below)
  const double *grow = tv.rows + min((uint32_t)lane, tv.n_dims - 1);
  uint32_t n_con = 0, cbar_t = 0;
  auto cbar = [&]() {
    const unsigned long long t0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    pipe_half_barrier(&s_cbar, cbar_t, lane);
    if (stamps) {
      const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0;
      atomicAdd(&g_tile_stamps[11], dt);
      t_last += dt;
    }
  };
  for (;;) {
    bool got = false;
    for (;;) {
      const uint32_t dn = __hip_atomic_load(&s_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t fl = __hip_atomic_load(&s_full, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if ((int32_t)(fl - n_con) > 0) {
        got = true;
        break;
      }
      if (dn) break;
      __builtin_amdgcn_s_sleep(2);
    }
    if (!got) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    stamp(8);  // waited for a chunk
    const uint32_t buf = n_con & 1u;
    const uint32_t UP = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_U[buf]), nb = UP / 16u;
    const uint32_t *uc = ucol + buf * kTileSetCap + 4u * g4;
    const uint32_t *xa = Xw + buf * G * XW + (32u * (uint32_t)mh + c16) * XW + g4;  // M tile 0 (tile 1: 16 rows on)
    // ---- the residual gather, lane = dimension: eight rows loaded before two blocks' MFMAs, added after them, in list order
    const uint32_t *wl = lists + (((uint64_t)blockIdx.x * 2 + buf) * 8 + cw) * kPipeListCap;
    const uint32_t wcnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_rtot[buf][cw]);
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
    auto load_rows = [&](double (&dst)[4], const uint4 u) {
      dst[0] = trow[(uint64_t)u.x * tv.d_pad];
      dst[1] = trow[(uint64_t)u.y * tv.d_pad];
      dst[2] = trow[(uint64_t)u.z * tv.d_pad];
      dst[3] = trow[(uint64_t)u.w * tv.d_pad];
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
    for (uint32_t b0 = 0; b0 < nb; b0 += 4) {  // (nb is a multiple of four)
#pragma unroll
      for (uint32_t s = 0; s < 4; ++s) {
        const uint32_t b = b0 + s;
        if (s == 0) gather_issue(gv, gj, entA);
        if (s == 2) gather_issue(gv, gj, entB);
        load_rows(bs[(s + 3u) & 3u], uqn);  // block b + 3's rows of T
        uqn = *reinterpret_cast<const uint4 *>(uc + 16u * min(b + 4u, nb - 1u));
        const uint32_t bx = min(b + 1u, nb - 1u);
        const uint32_t an0 = xa[4u * bx], an1 = xa[16u * XW + 4u * bx];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)((a0 >> (8u * j)) & 0xFFu), bs[s][j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)((a1 >> (8u * j)) & 0xFFu), bs[s][j], acc1, 0, 0, 0);
        }
        a0 = an0;
        a1 = an1;
        if ((s & 1u) == 1) gather_add(gv, gj);
      }
    }
    stamp(9);  // the matrix cores
    cbar();    // every consumer is done with X: its room takes the sums, [sequence][dimension]
    double *R = reinterpret_cast<double *>(Xw + buf * G * XW);
#pragma unroll
    for (uint32_t rr = 0; rr < 4; ++rr) {  // lane l holds rows (l >> 4) + 4 r of an M tile, column l & 15 of the wavefront's 16 dimensions
      R[(32u * (uint32_t)mh + g4 + 4u * rr) * 64u + 16u * (uint32_t)ni + c16] = acc0[rr];
      R[(32u * (uint32_t)mh + 16u + g4 + 4u * rr) * 64u + 16u * (uint32_t)ni + c16] = acc1[rr];
    }
    // the wavefront's list (all of it after the MFMAs' share), then its eight sequences' sums
    // (two batches in flight: sixteen rows a wavefront)
    if (gpos < wcnt) {
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
      const double v = __dadd_rn(R[sqn * 64u + (uint32_t)lane], rsum[j]);
      if (sl != ~0ull && (uint32_t)lane < tv.n_dims) partial[sl * tv.n_dims + lane] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&s_empty, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    ++n_con;
    stamp(10);  // the sums written
  }
}

}  // namespace kpop

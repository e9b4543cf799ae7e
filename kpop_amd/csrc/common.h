// common.h -- error plumbing, launch helpers and the per-process context.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kpop_hip.h"

namespace kpop {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
const char *get_error();

#define KPOP_FAIL(code, ...)        \
  do {                              \
    ::kpop::set_error(__VA_ARGS__); \
    return (code);                  \
  } while (0)

#define KPOP_HIP(expr)                                                                       \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      KPOP_FAIL(KPOP_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
  } while (0)

#define KPOP_TRY(expr)     \
  do {                     \
    int _rc = (expr);      \
    if (_rc != 0) return _rc; \
  } while (0)

#define KPOP_LAUNCH_CHECK() KPOP_HIP(hipGetLastError())

// grow-only device scratch owned by the library (long-sequence partial sums,
// segment tables).  ensure() may hipMalloc on the first call with a larger
// shape; steady-state calls only enqueue.
struct Workspace {
  void *p = nullptr;
  uint64_t bytes = 0;
  int ensure(uint64_t need, void **out);
  void release();
};

// Device scratch of the host-buffer entry points.  hipMalloc/hipFree cost milliseconds per call at the sizes a
// 100k-read batch needs, so scratch comes from grow-only chunks that are kept until kpop_shutdown(); an
// ArenaScope at the top of an entry point gives stack discipline (everything taken inside is handed back on exit).
struct Arena {
  struct Chunk {
    void *p;
    uint64_t bytes;
  };
  std::vector<Chunk> chunks;
  size_t cur = 0;
  uint64_t off = 0;
  int depth = 0;
  int take(uint64_t n, void **out);
  void release();
};

// One Context per DEVICE SLOT.  kpop_init(device) fills slot 0; kpop_init_devices(devices, n) fills slots 0..n-1 (two
// slots may name the same physical GPU -- that is how the multi-device paths are tested on a one-GPU box).  A host
// thread works on the slot it chose with kpop_use_device(slot) (slot 0 until it says otherwise); the library's own
// per-device worker threads choose theirs when they start.
struct Context {
  bool initialised = false;
  uint32_t generation = 1;  // bumped when the slot is released (PerSlotOnce)
  int slot = 0;
  int device = -1;
  int n_cus = 256;
  size_t lds_per_block = 65536;
  // The library-owned scratch of the device entry points is PER STREAM: two streams running kpop_dev_count_twist on
  // genomes, or kpop_dev_distance_summary against a large first operand, no longer share one block (round 2 did, and
  // raced).  The null stream's is the one kpop_dev_workspace_reserve() grows.
  std::mutex ws_mu;
  std::map<hipStream_t, Workspace> ws_by_stream;
  Workspace &ws_for(hipStream_t st) {
    std::lock_guard<std::mutex> g(ws_mu);
    return ws_by_stream[st];
  }
  // ... and a SECOND block a stream, for an entry point that prepares the input of another that takes the first (packed.hip: the
  // bases spread back to bytes for kpop_dev_count_twist)
  std::map<hipStream_t, Workspace> ws2_by_stream;
  Workspace &ws2_for(hipStream_t st) {
    std::lock_guard<std::mutex> g(ws_mu);
    return ws2_by_stream[st];
  }
  // ... and a stream of the library's own beside a caller's stream, for an entry point that runs two chains of kernels side by side
  // (the large-reference summary: one batch of query rows in its one-block-a-row kernels while the next is in the contraction).
  // Forked from and joined to the caller's stream with events: to the caller the work is still in order on the caller's stream.
  struct AuxLane {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, step[2] = {nullptr, nullptr};
  };
  std::map<hipStream_t, AuxLane> aux_by_stream;
  int aux_for(hipStream_t st, AuxLane **out);  // (runtime.hip; created on first use, destroyed with the slot)
  // host-buffer entry points (null stream + arena) of one slot run one at a time
  std::recursive_mutex host_mu;
  Arena arena;
  // tuning knobs (kpop_tune): gather depth, non-temporal row loads
  int tune_unroll = 8;
  int tune_nt = 2;       // row loads: 0 plain, 1 non-temporal, 2 by the size of the twister (count_twist.hip)
  int tune_seg = 0;      // windows per segment of the genome kernel, 0 = sized to the L2
  int tune_ldspad = 0;   // extra dynamic LDS per block of the fused reads kernel: fewer resident blocks (an occupancy probe)
  // development probes and A/B switches (kpop_tune("dbg", bits), KPOP_TUNE_DBG); results may be WRONG when the low ones are set:
  //   1, 2             count_wave_kernel: no look-back / no ticket          4 (the whole value)  summaries: the block kernel, not the wave kernel
  //   & 15             ca: inner sweeps of the blocked Jacobi step           32, 64, 2048         ca: plain steps / the looped kernel / no factor route
  //   256              counter: the median kernel's staging alone            4096, 8192, 16384    distance_rowwise: tile choices
  //   32768            wave summary: the doubled network instead of the tail   (>> 16) & 15       wave summary ablation: sort, chains, MAD, distances
  //   (>> 20) & 7      fused dense twist ablation                             (>> 24) & 15         count_twist_tile_kernel ablation: MFMA, X, set, lookups
  //   1024             distance_rowwise: the 256-row staging (two wavefronts a SIMD) where 128 rows would do
  //   1 << 28          CSR twist: one wavefront per spectrum even for a few very long spectra (same bits as the segmented launch)
  //   1 << 29          ca: stay on the Cholesky factor however many pivots were at the rounding floor
  //   1 << 30          summaries against <= 256 rows: round 3's kernel, one row of a wavefront at a time
  int tune_dbg = 0;
  int tune_dense = 2;    // the matrix-core routes of the twist: 2 (default) chosen by the batch -- assemblies through count_twist_tile_kernel (consensus on the matrix cores + residual gather), small-k assemblies through the dense image, dense spectra through the contraction --, 1 kpop_twist always dense, 0 never (the sparse mat-vec in the reference's order of additions everywhere)
  int tune_tileg = 64;   // sequences a chunk of count_twist_tile_kernel: 64 (one block of 1,024 threads a CU) or 32 (two of 512: measured slower, 1.96 against 1.64 ms on 5,000 mutants at 0.1 % -- every per-chunk step is paid twice as often)
  int tune_tilepipe = 1; // assemblies of one organism, up to 64 dimensions: count_twist_tile_pipe_kernel (producer and consumer wavefronts, tile_pipe.h); 0: round 4's count_twist_tile_kernel
  int tune_tilewide = 0; // 1: the three-stage form of that kernel (tile_pipe.h WIDE: what more than 64 dimensions get) at any number of dimensions
  int tune_tilecap_mb = 0; // MiB of per-slot tables a call of that route beyond 64 dimensions may take before the batch goes through in sub-batches (0: 4 GiB per 64 columns, a quarter of the device's memory at most)
  int tune_pipeprio = 1; // issue priority of the MFMA wavefronts of the tile kernel beyond 64 dimensions (producers: 2, gather wavefronts: 1); measured 0: 0.410 / 0.433, 1: 0.412 / 0.465, 2: 0.393 / 0.451, 3: 0.393 / 0.445 of the matrix peak at 256 / 1,635 dimensions; | 4: the gather with plain instead of non-temporal loads (no difference)
  int tune_blocksort = 1;  // -L on sequences of up to 32,768 windows: one block per sequence, sorted in LDS (0: device-wide sort)
  int tune_hist = 1;     // merged (-l) spectrum by atomic histogram when the hashes fit 26 bits (0: always sort)
  int tune_histguess = 1; // the merged count's partition path sizes its buckets from a sample of the items (one in 32) instead of a counting pass over all of them; a bucket that overflows sends the call back to the exact count.  0: always the exact count
  int tune_direct = 2; // a nearly complete twister of k 13..15 and <= 32 dimensions also keeps its rows at their hashes (twister.h): 2 by that rule, 1 whenever the table fits (any k <= 15), 0 never.  Read when a twister is loaded or synthesised
  int tune_summary_mfma_lists = 1; // the refinement reads the summary's candidate lists where its bands lie inside them; 0: it scans every distance row again
  int tune_distance_mfma = 1; // kpop_dev_distance_rowwise of 2^32 products and more (rows x rows x dimensions), euclidean / cosine: the contraction on the f64 matrix cores, pairs that cancel recomputed with the reference's chain (<= 1e-12 relative, not bit for bit); 0: the vector-pipe chain for every pair (the reference's bits)
  int tune_summary_mfma = 1; // (1, the default: approximate rows, then the summary's pass over them; 2: up to 128 dimensions the pass runs INSIDE the contraction and no approximate row is written -- same results, measured SLOWER: 256 x 1M x 64 2.9-3.2 against 2.7-2.8 ms, the classification's ~50 vector operations a pair serialise with the matrix pipe on a SIMD: profiles/r06_summary_select.txt) summaries against >= 65,536 rows, euclidean / cosine: the distances as f64 MFMAs + exact refinement (distance_mfma.hip); 0: the vector-pipe chain for every pair
  int tune_summary_lanes = 1; // 2: the matrix-core summary of 512 query rows and more in batches of 256 rows on TWO streams (the caller's and one of the library's), a batch's one-block-a-row kernels meant to run under the next batch's contraction and pass.  Measured level (1,024 x 1M x 64: 8.68 against 8.63 ms): a block of 1,024 threads and 100 KB of LDS does not get onto a CU while the other chain's grid holds it, the chains take turns anyway (profiles/r06_summary_lanes.txt).  Same results either way
  int tune_summary_sample = 1;  // the matrix-core summary's brackets and bands: 1 from the query rows' distances to a sample of the REFERENCE ROWS at even spacing (a small contraction of its own: whatever the layout of the database), 0 from 64 runs of 1,024 consecutive elements of the distance rows (a database laid out lineage by lineage makes those runs speak for a few lineages only: the brackets miss)
  int tune_summary_rawref = 1;  // the matrix-core summary takes the reference set as it is (norms and sums of squares from one pass, a dot product scaled where it comes out, the exact chains dividing as they go): no normalised copy of it; 0: the copy, as before
  int tune_summary_pass = 1;  // the matrix-core summary's pass over its approximate rows: 1 the pass for rows the library made (doubles against thresholds, a turn's candidates appended with one atomic a wavefront), 0 the general pass (keys, an append an element)
  int tune_summary_audit = 0;  // 1: count the rows the large-reference summaries leave to their exact fall-back (kpop_debug_summary_fallbacks reads and clears)
  uint64_t summary_fallback_rows = 0;
  int tune_summary2 = 1; // summaries against > 4,096 rows: 1 brackets and bands from a sample + ONE pass over distance rows, 3 the same in two passes (median, then MAD), 0 round 2's one block per row (8-10 passes), 2 the distances computed and reduced in one kernel, no distance rows (131,072 rows and more); 1, 2 and 3 are level at 256 x 1M
  int tune_histlds = 1;  // ... staged through LDS: private tables (k <= 7), sorted chunks of assemblies (0: direct atomics; 2: always sort the chunks)
};
constexpr int kMaxSlots = 16;
Context &ctx();              // the calling thread's slot
Context &ctx_of(int slot);
int current_slot();
int n_slots();
int use_slot(int slot);      // thread-local choice + hipSetDevice
int require_init();
void stop_slot_workers();  // multi.hip: the per-slot host threads end before their slots do
// once-per-device-slot latch for hipFuncSetAttribute and the like (function attributes are per device)
// (a slot that is released -- kpop_shutdown, kpop_init_devices with another GPU in it -- starts a new generation: the latches
// of the old one no longer hold, the new device needs its attributes set again)
struct PerSlotOnce {
  bool done[kMaxSlots] = {};
  uint32_t gen[kMaxSlots] = {};
  bool &operator()() {
    const int s = current_slot();
    const uint32_t g = ctx().generation;
    if (gen[s] != g) {
      gen[s] = g;
      done[s] = false;
    }
    return done[s];
  }
};

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// scopes open on the calling THREAD: a DevBuf takes arena memory only inside one of its own thread's scopes (which hold the slot's
// host_mu) -- a second host thread on the slot, in an entry point that opens none, gets a hipMalloc of its own instead of memory
// that somebody else's scope is about to rewind
int &arena_scopes_of_this_thread();

struct ArenaScope {
  Context &c;
  size_t cur;
  uint64_t off;
  ArenaScope() : c(ctx()) {
    c.host_mu.lock();
    Arena &a = c.arena;
    cur = a.cur;
    off = a.off;
    ++a.depth;
    ++arena_scopes_of_this_thread();
  }
  ~ArenaScope() {
    Arena &a = c.arena;
    a.cur = cur;
    a.off = off;
    --a.depth;
    --arena_scopes_of_this_thread();
    c.host_mu.unlock();
  }
  ArenaScope(const ArenaScope &) = delete;
  ArenaScope &operator=(const ArenaScope &) = delete;
};

// RAII device buffer for the host-side entry points: arena-backed inside an ArenaScope, hipMalloc'ed otherwise.
struct DevBuf {
  void *p = nullptr;
  uint64_t bytes = 0;
  bool owned = false;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() {
    if (p && owned) (void)hipFree(p);
  }
  int alloc(uint64_t n) {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    bytes = n;
    if (n == 0) n = 8;
    if (arena_scopes_of_this_thread() > 0) {
      owned = false;
      return ctx().arena.take(n, &p);
    }
    owned = true;
    KPOP_HIP(hipMalloc(&p, n));
    return 0;
  }
  template <class T>
  T *as() const {
    return reinterpret_cast<T *>(p);
  }
};

constexpr int kWave = 64;  // gfx950 wavefront

static inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
// HIP refuses launches with gridDim.x * blockDim.x >= 2^32; kernels given this grid stride over the rest
static inline uint32_t capped_grid(uint64_t blocks) { return (uint32_t)(blocks < (1u << 22) ? blocks : (1u << 22)); }

}  // namespace kpop

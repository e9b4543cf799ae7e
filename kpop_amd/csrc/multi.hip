// multi.hip -- several GPUs behind the C ABI, in ONE process (SURVEY.md 8b "multi-GPU sharding is internal", 8e).
//
// What it replaces: `-T` defaulting to every core (bin/KPopTwistDB.ml:103) and the fork()ed workers of
// Parallel.process_stream_chunkwise (lib/Twister.ml:90-196, lib/Matrix.ml:212-266, 712-766).  An OCaml host cannot call
// torch.distributed; it can call this.
//
// One host thread per device slot (kpop_init_devices), PERSISTENT: started at the first call that needs them, each chooses
// its slot once (thread-local, like hipSetDevice) and then takes job after job off a shared ticket until kpop_shutdown or
// the next kpop_init_devices.  (Round 3 started and joined a thread per slot per call and met at two condition-variable
// barriers per resident step: a visible share of a 2 ms step at eight GPUs.)  Every sequence is independent through count
// and twist, so
//
//   kpop_sharded_run              reads cut into contiguous shards (kpop_shard_bounds), one streaming pipeline
//                                 (pipeline.hip) per device, results written straight to the caller's rows: distances
//                                 against a reference set need NO exchange;
//   kpop_sharded_resident_step    the device-resident form of BASELINE config 4: every device twists its shard in
//                                 chunks and PUSHES each finished chunk into every peer's copy of the full matrix
//                                 (hipMemcpyPeerAsync, one stream per destination: xGMI is point-to-point, the seven
//                                 links of a GPU carry seven copies at once) while the next chunk is being twisted --
//                                 the all-gather of SURVEY.md 8e without a collective library in the way;
//   kpop_sharded_all_vs_all_summary  every device summarises (lib/Matrix.ml:691-766) rows of its shard against all N
//                                 gathered vectors; N x N is never formed;
//   kpop_sharded_distance_summary host matrices, second operand's rows cut over the devices, first operand replicated.
//
// Two slots may sit on the same physical GPU (that is how this file is tested on a one-GPU box): a twister replica is
// then an alias and a "peer" copy is a device-local one; the order of operations is the same.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "twister.h"

namespace {

using namespace kpop;

// a rendezvous of the slot threads inside a job: they are all running (the job was handed to every one of them), the
// wait is short, so it spins on an atomic (and yields now and then) instead of sleeping on a condition variable
struct SpinBarrier {
  std::atomic<int> waiting{0};
  std::atomic<uint64_t> phase{0};
  int n;
  explicit SpinBarrier(int n_) : n(n_) {}
  void wait() {
    const uint64_t ph = phase.load(std::memory_order_acquire);
    if (waiting.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
      waiting.store(0, std::memory_order_relaxed);
      phase.store(ph + 1, std::memory_order_release);
    } else {
      for (uint32_t spins = 0; phase.load(std::memory_order_acquire) == ph; ++spins)
        if ((spins & 1023u) == 1023u) std::this_thread::yield();
    }
  }
};

struct Block {  // grow-only device block of one slot
  void *p = nullptr;
  uint64_t bytes = 0;
  int ensure(uint64_t need) {
    if (need <= bytes) return 0;
    if (p) {
      KPOP_HIP(hipDeviceSynchronize());
      KPOP_HIP(hipFree(p));
      p = nullptr;
      bytes = 0;
    }
    KPOP_HIP(hipMalloc(&p, need + 256));
    bytes = need + 256;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <class T>
  T *as() const {
    return reinterpret_cast<T *>(p);
  }
};

struct SlotState {
  int slot = 0;
  kpop_twister *tw = nullptr;     // this slot's replica (owned unless it is the caller's own handle)
  bool owns_tw = false;
  kpop_pipeline *pl = nullptr;
  double *d_classes = nullptr, *d_metric = nullptr;
  hipStream_t s_compute = nullptr;
  std::vector<hipStream_t> s_comm;  // one per destination slot
  std::vector<hipEvent_t> chunk_done, chunk_t0, chunk_t1;  // per chunk of a resident step: twisted (for the pushes); around its kernels (timing)
  std::vector<float> chunk_ms;
  int chunks_timed = 0;
  Block full, dmat, work, stats, nn, idx, ndist, z;
  double ms_compute = 0.0, ms_exposed_comm = 0.0;
};

// The slot threads.  run(n, fn): fn(i) on slot i's thread for every i < n, at the same time; returns when all have returned;
// the first failure (status + message) is the caller's.  A thread that could not choose its slot makes every job fail up
// front -- no fn is entered anywhere, so nobody waits at a barrier for a partner that never came.
class SlotWorkers {
  std::mutex mu_call;  // one job at a time
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::vector<std::thread> th;
  std::vector<int> rc, slot_rc;
  std::vector<std::string> msg, slot_msg;
  std::function<int(int)> job;
  std::atomic<uint64_t> epoch{0};
  std::atomic<int> remaining{0};
  int n = 0;
  bool stop = false;

  void body(int i) {
    slot_rc[i] = use_slot(i);
    if (slot_rc[i] != 0) slot_msg[i] = get_error();
    uint64_t seen = epoch.load(std::memory_order_acquire);  // (the tickets of the threads before these are not theirs to take)
    {
      std::lock_guard<std::mutex> lk(mu);  // (started: run() waits for this before the first job)
      remaining.fetch_sub(1, std::memory_order_acq_rel);
      cv_done.notify_all();
    }
    for (;;) {
      // a job usually follows the last within microseconds (a step per call): spin a little, then sleep
      for (uint32_t spins = 0; spins < 20000 && epoch.load(std::memory_order_acquire) == seen; ++spins) {}
      if (epoch.load(std::memory_order_acquire) == seen) {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return stop || epoch.load(std::memory_order_acquire) != seen; });
      }
      if (stop) return;
      seen = epoch.load(std::memory_order_acquire);
      int r = job(i);
      rc[i] = r;
      if (r != 0) msg[i] = get_error();
      if (remaining.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> lk(mu);
        cv_done.notify_all();
      }
    }
  }
  void wait_done() {
    for (uint32_t spins = 0; spins < 200000 && remaining.load(std::memory_order_acquire) != 0; ++spins) {}
    if (remaining.load(std::memory_order_acquire) != 0) {
      std::unique_lock<std::mutex> lk(mu);
      cv_done.wait(lk, [&] { return remaining.load(std::memory_order_acquire) == 0; });
    }
  }

 public:
  ~SlotWorkers() { shutdown(); }
  void shutdown() {
    std::lock_guard<std::mutex> call(mu_call);
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
      cv_job.notify_all();
    }
    for (auto &t : th)
      if (t.joinable()) t.join();
    th.clear();
    n = 0;
    stop = false;
  }
  template <class F>
  int run(int want, F fn) {
    std::lock_guard<std::mutex> call(mu_call);
    if (want != n) {  // (first use, or another number of slots: kpop_init_devices stops the threads, so they never outlive their slots)
      {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
        cv_job.notify_all();
      }
      for (auto &t : th)
        if (t.joinable()) t.join();
      th.clear();
      stop = false;
      n = want;
      rc.assign(n, 0);
      slot_rc.assign(n, 0);
      msg.assign(n, std::string());
      slot_msg.assign(n, std::string());
      remaining.store(n, std::memory_order_release);
      for (int i = 0; i < n; ++i) th.emplace_back([this, i] { body(i); });
      wait_done();
    }
    for (int i = 0; i < n; ++i)
      if (slot_rc[i] != 0) {
        set_error("device slot %d: %s", i, slot_msg[i].c_str());
        return slot_rc[i];
      }
    job = fn;
    std::fill(rc.begin(), rc.end(), 0);
    remaining.store(n, std::memory_order_release);
    {
      std::lock_guard<std::mutex> lk(mu);
      epoch.fetch_add(1, std::memory_order_acq_rel);
      cv_job.notify_all();
    }
    wait_done();
    job = nullptr;
    for (int i = 0; i < n; ++i)
      if (rc[i] != 0) {
        set_error("device slot %d: %s", i, msg[i].c_str());
        return rc[i];
      }
    return 0;
  }
};
static SlotWorkers g_workers;

template <class F>
static int on_every_slot(int n, F fn) {
  return g_workers.run(n, fn);
}

static void bounds(uint64_t n_items, int rank, int world, uint64_t *lo, uint64_t *hi) {
  const uint64_t base = n_items / world, extra = n_items % world;
  *lo = (uint64_t)rank * base + std::min<uint64_t>(rank, extra);
  *hi = *lo + base + ((uint64_t)rank < extra ? 1 : 0);
}

}  // namespace

namespace kpop {
void stop_slot_workers() { g_workers.shutdown(); }
}  // namespace kpop

struct kpop_sharded {
  int n = 0;
  std::vector<SlotState> s;
  kpop_pipeline_config cfg{};
  uint32_t n_dims = 0, n_classes = 0;
  // the last resident step
  uint64_t n_total = 0;
  std::vector<uint64_t> lo, hi;
  bool gathered = false;
};

// Contiguous, balanced ranges: the first n_items % world ranks get one item more (kpop_amd/shard.py:shard_bounds; the
// reference deals chunks of elements_per_step to whichever worker is free, lib/Twister.ml:90-93 -- any partition gives
// the same rows).  Pure host arithmetic: no GPU needed.
extern "C" int kpop_shard_bounds(uint64_t n_items, int rank, int world, uint64_t *lo, uint64_t *hi) {
  if (world <= 0 || rank < 0 || rank >= world || !lo || !hi)
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_shard_bounds: rank %d outside world %d", rank, world);
  bounds(n_items, rank, world, lo, hi);
  return KPOP_OK;
}

extern "C" int kpop_twister_replicate(const kpop_twister *src, int slot, kpop_twister **out) {
  KPOP_TRY(require_init());
  if (!src || !out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_replicate: null argument");
  if (slot < 0 || slot >= n_slots()) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_twister_replicate: slot %d of %d", slot, n_slots());
  const int src_dev = ctx_of(src->slot).device, dst_dev = ctx_of(slot).device;
  kpop_twister *tw = new kpop_twister(*src);
  tw->slot = slot;
  if (src_dev == dst_dev) {  // the same HBM: a second handle, not a second copy
    tw->alias = true;
    *out = tw;
    return KPOP_OK;
  }
  tw->alias = false;
  tw->d_rows = nullptr;
  tw->d_rsel = nullptr;
  tw->d_rblk = nullptr;
  tw->d_sorted_hash = nullptr;
  tw->d_direct = nullptr;  // (the copy keeps the rows in rank order only: device_bytes below)
  if (src->d_direct) tw->device_bytes -= (1ull << (2 * src->k)) * src->d_pad * 8;
  const int prev = current_slot();
  int rc = use_slot(slot);
  if (rc == 0) do {
#define RP(expr)                                                                                   \
  {                                                                                                \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_));              \
      rc = KPOP_ERR_HIP;                                                                           \
      break;                                                                                       \
    }                                                                                              \
  }
      const uint64_t row_bytes = src->n_rows * (uint64_t)src->d_pad * 8;
      RP(hipMalloc((void **)&tw->d_rows, row_bytes ? row_bytes : 8));
      if (row_bytes) RP(hipMemcpyPeer(tw->d_rows, dst_dev, src->d_rows, src_dev, row_bytes));
      if (src->d_rsel) {
        const uint64_t b = (((1ull << (2 * src->k)) + 63) / 64) * sizeof(RankWord);
        RP(hipMalloc(&tw->d_rsel, b));
        RP(hipMemcpyPeer(tw->d_rsel, dst_dev, src->d_rsel, src_dev, b));
      }
      if (src->d_rblk) {
        const uint64_t b = rank_blocks(src->k) * 64;
        RP(hipMalloc(&tw->d_rblk, b));
        RP(hipMemcpyPeer(tw->d_rblk, dst_dev, src->d_rblk, src_dev, b));
      }
      if (src->d_sorted_hash) {
        RP(hipMalloc((void **)&tw->d_sorted_hash, src->n_rows * 8 + 8));
        if (src->n_rows) RP(hipMemcpyPeer(tw->d_sorted_hash, dst_dev, src->d_sorted_hash, src_dev, src->n_rows * 8));
      }
      RP(hipDeviceSynchronize());
#undef RP
    } while (0);
  (void)use_slot(prev);
  if (rc != 0) {
    kpop_twister_free(tw);
    return rc;
  }
  *out = tw;
  return KPOP_OK;
}

extern "C" int kpop_sharded_destroy(kpop_sharded *sh) {
  if (!sh) return KPOP_OK;
  const int prev = current_slot();
  for (SlotState &st : sh->s) {
    if (use_slot(st.slot) != 0) continue;
    if (st.s_compute) (void)hipStreamSynchronize(st.s_compute);
    for (hipStream_t c : st.s_comm)
      if (c) (void)hipStreamSynchronize(c);
    if (st.pl) kpop_pipeline_destroy(st.pl);
    if (st.owns_tw && st.tw) kpop_twister_free(st.tw);
    Block *all[] = {&st.full, &st.dmat, &st.work, &st.stats, &st.nn, &st.idx, &st.ndist, &st.z};
    for (Block *b : all) b->release();
    if (st.d_classes) (void)hipFree(st.d_classes);
    if (st.d_metric) (void)hipFree(st.d_metric);
    for (auto *v : {&st.chunk_done, &st.chunk_t0, &st.chunk_t1})
      for (hipEvent_t e : *v)
        if (e) (void)hipEventDestroy(e);
    if (st.s_compute) {
      Context &c = ctx();
      std::lock_guard<std::mutex> lk(c.ws_mu);
      auto it = c.ws_by_stream.find(st.s_compute);
      if (it != c.ws_by_stream.end()) {
        it->second.release();
        c.ws_by_stream.erase(it);
      }
      (void)hipStreamDestroy(st.s_compute);
    }
    for (hipStream_t c : st.s_comm)
      if (c) (void)hipStreamDestroy(c);
  }
  (void)use_slot(prev);
  delete sh;
  return KPOP_OK;
}

extern "C" int kpop_sharded_create(const kpop_twister *tw, const double *classes, uint32_t n_classes, const double *metric,
                                   const kpop_pipeline_config *cfg, kpop_sharded **out) {
  KPOP_TRY(require_init());
  if (!tw || !cfg || !out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_sharded_create: null argument");
  if (cfg->struct_size != sizeof(kpop_pipeline_config))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_sharded_create: config of %u bytes, this library's is %zu", cfg->struct_size, sizeof(kpop_pipeline_config));
  const int n = n_slots();
  kpop_sharded *sh = new kpop_sharded();
  sh->n = n;
  sh->s.resize(n);
  sh->cfg = *cfg;
  sh->n_dims = tw->n_dims;
  const bool need_classes = cfg->outputs & (KPOP_OUT_DISTANCES | KPOP_OUT_SUMMARY);
  sh->n_classes = (classes && metric) ? n_classes : 0;
  if (need_classes && !sh->n_classes) {
    delete sh;
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_sharded_create: distances / summary need the class vectors and the metric");
  }
  for (int i = 0; i < n; ++i) sh->s[i].slot = i;
  // replicas first, from the calling thread (the source device is busy with nothing else), then the per-slot state
  int rc = KPOP_OK;
  for (int i = 0; i < n && rc == KPOP_OK; ++i) {
    if (i == tw->slot) {
      sh->s[i].tw = const_cast<kpop_twister *>(tw);
      sh->s[i].owns_tw = false;
    } else {
      rc = kpop_twister_replicate(tw, i, &sh->s[i].tw);
      sh->s[i].owns_tw = rc == KPOP_OK;
    }
  }
  if (rc == KPOP_OK)
    rc = on_every_slot(n, [&](int i) -> int {
      SlotState &st = sh->s[i];
      KPOP_TRY(kpop_pipeline_create(st.tw, classes, sh->n_classes, metric, cfg, &st.pl));
      KPOP_HIP(hipStreamCreateWithFlags(&st.s_compute, hipStreamNonBlocking));
      st.s_comm.assign(n, nullptr);
      for (int j = 0; j < n; ++j)
        if (j != i) KPOP_HIP(hipStreamCreateWithFlags(&st.s_comm[j], hipStreamNonBlocking));
      if (sh->n_classes) {
        const uint64_t cb = (uint64_t)sh->n_classes * sh->n_dims * 8, mb = (uint64_t)sh->n_dims * 8;
        KPOP_HIP(hipMalloc((void **)&st.d_classes, cb));
        KPOP_HIP(hipMalloc((void **)&st.d_metric, mb));
        KPOP_HIP(hipMemcpy(st.d_classes, classes, cb, hipMemcpyHostToDevice));
        KPOP_HIP(hipMemcpy(st.d_metric, metric, mb, hipMemcpyHostToDevice));
      } else if (metric) {
        KPOP_HIP(hipMalloc((void **)&st.d_metric, (uint64_t)sh->n_dims * 8));
        KPOP_HIP(hipMemcpy(st.d_metric, metric, (uint64_t)sh->n_dims * 8, hipMemcpyHostToDevice));
      }
      return 0;
    });
  if (rc != KPOP_OK) {
    const std::string keep = get_error();
    kpop_sharded_destroy(sh);
    set_error("%s", keep.c_str());
    return rc;
  }
  *out = sh;
  return KPOP_OK;
}

extern "C" int kpop_sharded_slots(const kpop_sharded *sh) { return sh ? sh->n : 0; }

// host memory -> host memory, distances against the reference set: no exchange
extern "C" int kpop_sharded_run(kpop_sharded *sh, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                                const kpop_pipeline_outputs *o) {
  if (!sh || !o || (n_reads && !offsets)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_sharded_run: null argument");
  const uint32_t D = sh->n_dims, C = sh->n_classes, mn = sh->cfg.max_neighbours;
  return on_every_slot(sh->n, [&](int i) -> int {
    uint64_t lo, hi;
    bounds(n_reads, i, sh->n, &lo, &hi);
    if (hi == lo) return 0;
    kpop_pipeline_outputs mine = *o;
    if (mine.twisted) mine.twisted += lo * D;
    if (mine.distances) mine.distances += lo * C;
    if (mine.stats) mine.stats += lo * 4;
    if (mine.n_neighbours) mine.n_neighbours += lo;
    if (mine.nb_index) mine.nb_index += lo * mn;
    if (mine.nb_distance) mine.nb_distance += lo * mn;
    if (mine.nb_z) mine.nb_z += lo * mn;
    return kpop_pipeline_run(sh->s[i].pl, bases, offsets + lo, (uint32_t)(hi - lo), &mine);
  });
}

// kpop_spectra_twist over all devices (sequences of any length, the rows kpop_count_reads + kpop_twist would give bit for
// bit): what KPopTwistDB runs on a block of the reads stream that the fused kernel does not cover
extern "C" int kpop_sharded_spectra_twist(kpop_sharded *sh, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k,
                                          int content, int normalize, double *out) {
  if (!sh || (n_reads && (!offsets || !out))) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_sharded_spectra_twist: null argument");
  return on_every_slot(sh->n, [&](int i) -> int {
    uint64_t lo, hi;
    bounds(n_reads, i, sh->n, &lo, &hi);
    if (hi == lo) return 0;
    return kpop_spectra_twist(sh->s[i].tw, bases, offsets + lo, (uint32_t)(hi - lo), k, content, normalize, out + lo * sh->n_dims);
  });
}

// BASELINE config 4, device-resident: slot i holds its reads (d_bases[i], d_offsets[i]: n_reads[i] + 1 offsets into
// d_bases[i]) in its own HBM.  Twist in `chunks` pieces; with gather != 0 every finished piece is pushed to all peers
// while the next is twisted; then the distances of the slot's rows to the class vectors.  Returns when every device has
// finished and (gather) every copy of the full matrix is complete.
extern "C" int kpop_sharded_resident_step(kpop_sharded *sh, const uint8_t *const *d_bases, const uint64_t *const *d_offsets,
                                          const uint32_t *n_reads, const uint64_t *n_bases, uint32_t max_len, int chunks,
                                          int gather) {
  if (!sh || !d_bases || !d_offsets || !n_reads || !n_bases) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_sharded_resident_step: null argument");
  const int n = sh->n;
  const uint32_t D = sh->n_dims, C = sh->n_classes;
  sh->lo.assign(n, 0);
  sh->hi.assign(n, 0);
  uint64_t total = 0;
  for (int i = 0; i < n; ++i) {
    sh->lo[i] = total;
    total += n_reads[i];
    sh->hi[i] = total;
  }
  sh->n_total = total;
  sh->gathered = gather != 0;
  if (chunks < 1) chunks = 1;
  for (int i = 0; i < n; ++i)  // (a slot that could not start its thread's work would leave the others at the barrier)
    if (!ctx_of(i).initialised) KPOP_FAIL(KPOP_ERR_NOT_INIT, "kpop_sharded_resident_step: device slot %d is gone (kpop_shutdown?)", i);
  SpinBarrier bar(n);
  std::atomic<int> failed{0};  // a slot whose buffers could not be had: nobody pushes anything anywhere (its `full` may be gone)
  return on_every_slot(n, [&](int i) -> int {
    SlotState &st = sh->s[i];
    int rc = 0;
    // phase 1: buffers (every slot holds the full matrix; its own rows are written in place by the twist)
    do {
      if ((rc = st.full.ensure(std::max<uint64_t>(total, 1) * D * 8))) break;
      if (C && (rc = st.dmat.ensure(std::max<uint64_t>(n_reads[i], 1) * C * 8))) break;
      if (C && (rc = st.work.ensure(kpop_dev_distance_workspace_bytes(C, n_reads[i], D)))) break;
      while ((int)st.chunk_done.size() < chunks) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
          set_error("kpop_sharded_resident_step: hipEventCreate failed");
          rc = KPOP_ERR_HIP;
          break;
        }
        st.chunk_done.push_back(e);
      }
      while (rc == 0 && (int)st.chunk_t0.size() < chunks) {
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
          set_error("kpop_sharded_resident_step: hipEventCreate failed");
          rc = KPOP_ERR_HIP;
          break;
        }
        st.chunk_t0.push_back(e0);
        st.chunk_t1.push_back(e1);
      }
    } while (0);
    if (rc != 0) failed.store(1, std::memory_order_release);
    bar.wait();  // every slot's `full` pointer is final before anybody pushes into it -- the step's ONE rendezvous
    // (a failed slot still takes part in it, or the others would wait for ever; and then nobody goes on)
    if (rc == 0 && failed.load(std::memory_order_acquire)) {
      set_error("kpop_sharded_resident_step: another device slot failed to allocate its buffers");
      rc = KPOP_ERR_HIP;
    }
    st.chunks_timed = 0;
    const auto t0 = std::chrono::steady_clock::now();
    if (rc == 0) do {
        const uint32_t ni = n_reads[i];
        const uint32_t per = (ni + chunks - 1) / chunks;
        double *mine = st.full.as<double>() + sh->lo[i] * D;
        for (int c = 0; c < chunks && rc == 0; ++c) {
          const uint32_t a = std::min<uint64_t>((uint64_t)c * per, ni), b = std::min<uint64_t>((uint64_t)a + per, ni);
          if (b == a) continue;
          (void)hipEventRecord(st.chunk_t0[c], st.s_compute);
          if ((rc = kpop_dev_count_twist(st.tw, d_bases[i], d_offsets[i] + a, b - a, n_bases[i], max_len, sh->cfg.content,
                                         sh->cfg.normalize_counts, mine + (uint64_t)a * D, st.s_compute)))
            break;
          (void)hipEventRecord(st.chunk_t1[c], st.s_compute);
          st.chunks_timed = c + 1;
          if (!gather || n == 1) continue;
          if (hipEventRecord(st.chunk_done[c], st.s_compute) != hipSuccess) {
            set_error("hipEventRecord failed");
            rc = KPOP_ERR_HIP;
            break;
          }
          for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            SlotState &peer = sh->s[j];
            hipError_t e = hipStreamWaitEvent(st.s_comm[j], st.chunk_done[c], 0);
            if (e == hipSuccess)
              e = hipMemcpyPeerAsync(peer.full.as<double>() + (sh->lo[i] + a) * D, ctx_of(j).device, mine + (uint64_t)a * D,
                                     ctx_of(i).device, (uint64_t)(b - a) * D * 8, st.s_comm[j]);
            if (e != hipSuccess) {
              set_error("all-gather push %d -> %d: %s", i, j, hipGetErrorString(e));
              rc = KPOP_ERR_HIP;
              break;
            }
          }
        }
        if (rc == 0 && C && ni)
          rc = kpop_dev_distance_rowwise(st.d_classes, C, mine, ni, D, st.d_metric, sh->cfg.kind, sh->cfg.p, sh->cfg.normalize_distances,
                                         st.work.p, st.dmat.as<double>(), st.s_compute);
      } while (0);
    hipError_t e = hipStreamSynchronize(st.s_compute);
    const auto t1 = std::chrono::steady_clock::now();
    for (int j = 0; j < n && e == hipSuccess; ++j)
      if (st.s_comm[j]) e = hipStreamSynchronize(st.s_comm[j]);
    const auto t2 = std::chrono::steady_clock::now();
    st.ms_compute = std::chrono::duration<double, std::milli>(t1 - t0).count();
    st.ms_exposed_comm = std::chrono::duration<double, std::milli>(t2 - t1).count();
    if (rc == 0 && e != hipSuccess) {
      set_error("kpop_sharded_resident_step: %s", hipGetErrorString(e));
      rc = KPOP_ERR_HIP;
    }
    st.chunk_ms.assign(st.chunks_timed, 0.f);
    for (int c = 0; c < st.chunks_timed && rc == 0; ++c)
      if (hipEventElapsedTime(&st.chunk_ms[c], st.chunk_t0[c], st.chunk_t1[c]) != hipSuccess) st.chunk_ms[c] = 0.f;  // (a chunk without reads)
    // (no second rendezvous: this slot's pushes have landed -- its comm streams are drained -- and the call returns when every
    // slot's thread has come this far, so every copy of the full matrix is complete when the caller sees it)
    return rc;
  });
}

// device pointers of slot i after a resident step: the full matrix (n_total x n_dims, read order; rows of other slots
// valid only after a step with gather), this slot's first row in it, and its distances to the classes
extern "C" int kpop_sharded_resident_buffers(const kpop_sharded *sh, int slot, double **d_full, uint64_t *first_row, uint64_t *n_rows,
                                             double **d_distances) {
  if (!sh || slot < 0 || slot >= sh->n || sh->lo.empty()) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_sharded_resident_buffers: no resident step yet, or bad slot");
  if (d_full) *d_full = sh->s[slot].full.as<double>();
  if (first_row) *first_row = sh->lo[slot];
  if (n_rows) *n_rows = sh->hi[slot] - sh->lo[slot];
  if (d_distances) *d_distances = sh->s[slot].dmat.as<double>();
  return KPOP_OK;
}

// the fused count->twist launches of the last resident step on `slot`, chunk by chunk (HIP events on the slot's compute stream):
// what a roofline of the in-process path is computed from.  *n_chunks = entries written (at most max_chunks).
extern "C" int kpop_sharded_chunk_timings(const kpop_sharded *sh, int slot, double *ms, int max_chunks, int *n_chunks) {
  if (!sh || slot < 0 || slot >= sh->n || !n_chunks || (max_chunks > 0 && !ms)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_sharded_chunk_timings: bad argument");
  const SlotState &st = sh->s[slot];
  const int m = std::min<int>(max_chunks, (int)st.chunk_ms.size());
  for (int c = 0; c < m; ++c) ms[c] = st.chunk_ms[c];
  *n_chunks = m;
  return KPOP_OK;
}

extern "C" int kpop_sharded_timings(const kpop_sharded *sh, int slot, double *ms_compute, double *ms_exposed_comm) {
  if (!sh || slot < 0 || slot >= sh->n) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_sharded_timings: bad slot");
  if (ms_compute) *ms_compute = sh->s[slot].ms_compute;
  if (ms_exposed_comm) *ms_exposed_comm = sh->s[slot].ms_exposed_comm;
  return KPOP_OK;
}

// After a resident step with gather: slot i summarises the first min(queries_per_slot, n_i) rows of its shard (0 = all
// of them) against ALL n_total twisted vectors (Matrix.summarize_rowwise, lib/Matrix.ml:691-766).  Outputs are host
// arrays with one row per query, slots in order; out_query gets the global read number of each query; neighbour
// indices are global read numbers.  *n_queries_out = rows written.
extern "C" int kpop_sharded_all_vs_all_summary(kpop_sharded *sh, uint32_t queries_per_slot, uint32_t keep_at_most, uint32_t max_neighbours,
                                               uint64_t capacity, uint64_t *n_queries_out, uint64_t *out_query, double *out_stats,
                                               uint32_t *out_n, uint32_t *out_idx, double *out_dist, double *out_z) {
  if (!sh || !out_stats || !out_n) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_sharded_all_vs_all_summary: null argument");
  if (sh->lo.empty() || (!sh->gathered && sh->n > 1))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_sharded_all_vs_all_summary: needs a resident step with gather first");
  if (!sh->s[0].d_metric) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_sharded_all_vs_all_summary: the job was created without a metric");
  if (sh->n_total > 0xFFFFFFFFull) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_sharded_all_vs_all_summary: more than 2^32 vectors");
  const int n = sh->n;
  const uint32_t D = sh->n_dims, mn = max_neighbours;
  std::vector<uint64_t> q(n), q0(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    const uint64_t ni = sh->hi[i] - sh->lo[i];
    q[i] = queries_per_slot ? std::min<uint64_t>(queries_per_slot, ni) : ni;
    q0[i + 1] = q0[i] + q[i];
  }
  if (q0[n] > capacity) KPOP_FAIL(KPOP_ERR_CAPACITY, "kpop_sharded_all_vs_all_summary: %llu queries, capacity %llu", (unsigned long long)q0[n], (unsigned long long)capacity);
  if (n_queries_out) *n_queries_out = q0[n];
  return on_every_slot(n, [&](int i) -> int {
    SlotState &st = sh->s[i];
    const uint32_t qi = (uint32_t)q[i];
    if (!qi) return 0;
    const uint32_t N = (uint32_t)sh->n_total;
    KPOP_TRY(st.work.ensure(kpop_dev_distance_workspace_bytes(N, qi, D)));
    KPOP_TRY(st.stats.ensure((uint64_t)qi * 32));
    KPOP_TRY(st.nn.ensure((uint64_t)qi * 4));
    KPOP_TRY(st.idx.ensure((uint64_t)qi * std::max(mn, 1u) * 4));
    KPOP_TRY(st.ndist.ensure((uint64_t)qi * std::max(mn, 1u) * 8));
    KPOP_TRY(st.z.ensure((uint64_t)qi * std::max(mn, 1u) * 8));
    const double *full = st.full.as<double>();
    KPOP_TRY(kpop_dev_distance_summary(full, N, full + sh->lo[i] * D, qi, D, st.d_metric, sh->cfg.kind, sh->cfg.p, sh->cfg.normalize_distances,
                                       keep_at_most, mn, st.work.p, st.stats.as<double>(), st.nn.as<uint32_t>(), st.idx.as<uint32_t>(),
                                       st.ndist.as<double>(), st.z.as<double>(), st.s_compute));
    const uint64_t r = q0[i];
    KPOP_HIP(hipMemcpyAsync(out_stats + r * 4, st.stats.p, (uint64_t)qi * 32, hipMemcpyDeviceToHost, st.s_compute));
    KPOP_HIP(hipMemcpyAsync(out_n + r, st.nn.p, (uint64_t)qi * 4, hipMemcpyDeviceToHost, st.s_compute));
    if (mn && out_idx && out_dist && out_z) {
      KPOP_HIP(hipMemcpyAsync(out_idx + r * mn, st.idx.p, (uint64_t)qi * mn * 4, hipMemcpyDeviceToHost, st.s_compute));
      KPOP_HIP(hipMemcpyAsync(out_dist + r * mn, st.ndist.p, (uint64_t)qi * mn * 8, hipMemcpyDeviceToHost, st.s_compute));
      KPOP_HIP(hipMemcpyAsync(out_z + r * mn, st.z.p, (uint64_t)qi * mn * 8, hipMemcpyDeviceToHost, st.s_compute));
    }
    if (out_query)
      for (uint32_t t = 0; t < qi; ++t) out_query[r + t] = sh->lo[i] + t;
    KPOP_HIP(hipStreamSynchronize(st.s_compute));
    return 0;
  });
}

// kpop_distance_summary with the rows of the second operand cut over every device slot (first operand, metric
// replicated by each slot's upload): `KPopTwistDB -s` over several GPUs, lib/Matrix.ml:712-766's fork()ed workers.
extern "C" int kpop_sharded_distance_summary(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                                             const double *metric, int kind, double p, int normalize, uint32_t keep_at_most,
                                             uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx,
                                             double *out_dist, double *out_z) {
  KPOP_TRY(require_init());
  const int n = n_slots();
  const uint32_t mn = max_neighbours;
  return on_every_slot(n, [&](int i) -> int {
    uint64_t lo, hi;
    bounds(r2, i, n, &lo, &hi);
    if (hi == lo) return 0;
    return kpop_distance_summary(m1, r1, m2 + lo * n_dims, (uint32_t)(hi - lo), n_dims, metric, kind, p, normalize, keep_at_most, mn,
                                 out_stats + lo * 4, out_n + lo, out_idx ? out_idx + lo * mn : nullptr, out_dist ? out_dist + lo * mn : nullptr,
                                 out_z ? out_z + lo * mn : nullptr);
  });
}

// ... and kpop_distance_rowwise: rows of the result (= rows of the second operand) cut over the slots
extern "C" int kpop_sharded_distance_rowwise(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                                             const double *metric, int kind, double p, int normalize, double *out) {
  KPOP_TRY(require_init());
  const int n = n_slots();
  return on_every_slot(n, [&](int i) -> int {
    uint64_t lo, hi;
    bounds(r2, i, n, &lo, &hi);
    if (hi == lo) return 0;
    return kpop_distance_rowwise(m1, r1, m2 + lo * n_dims, (uint32_t)(hi - lo), n_dims, metric, kind, p, normalize, out + lo * r1);
  });
}

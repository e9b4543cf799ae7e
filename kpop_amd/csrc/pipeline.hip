// pipeline.hip -- the streaming host pipeline: reads in host memory -> twisted rows / distances / summary in host
// memory, with the bus and the kernels busy at the same time.
//
// What it replaces: the README.md:606 + :641/:656 chain,
//     KPopCount -L (bin/KPopCount.ml:36-50) | KPopTwistDB -k (lib/Twister.ml:58-206) ; KPopTwistDB -d / -s
//     (lib/Matrix.ml:191-266, 691-766)
// as ONE entry point for a host that holds reads in memory.  Round 2's host entry points ran H2D, kernels and D2H one
// after the other on the null stream from pageable memory, and a caller that wanted distances paid for the twisted rows
// crossing the bus twice (kpop_count_twist, then kpop_distance_rowwise).  Here:
//
//   * a batch is cut into chunks (reads are independent through count and twist);
//   * three streams: chunk c+1 goes up (H2D) while chunk c runs count->twist->distance and chunk c-1 comes down (D2H).
//     PCIe is full duplex, so both directions and the kernels overlap; the order inside a stream is the chunk order;
//   * a ring of device slots (inputs, outputs, distance workspace: per slot, so nothing is shared between chunks in
//     flight; the genome kernel's segment partials are per stream, common.h);
//   * `outputs` picks what comes back: a caller after distances (-d) or a summary (-s) never pulls the twisted rows;
//   * copies go straight from / to the caller's buffers.  When those are page-locked (kpop_host_alloc, or
//     kpop_host_register of memory the caller owns) the copy engines run asynchronously and submit() returns at once;
//     pageable buffers work too (HIP stages or pins them on the fly) with less overlap.
//
// Results are those of kpop_dev_count_twist + kpop_dev_distance_rowwise / kpop_dev_distance_summary on every chunk:
// every row depends on its own read only, so the chunking does not show in the output.
#include <algorithm>
#include <vector>

#include "common.h"
#include "twister.h"

namespace {

using namespace kpop;

struct DevMem {  // grow-only device block
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
    const uint64_t want = need + need / 8 + 256;
    KPOP_HIP(hipMalloc(&p, want));
    bytes = want;
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

struct RingSlot {
  DevMem bases, offsets, twisted, dist, work, stats, nn, idx, ndist, z;
  DevMem pcodes, pmask;  // a chunk's slice of a PACKED batch (kpop_pipeline_submit_packed), spread back into `bases` on the compute stream
  hipEvent_t h2d_done = nullptr, twist_done = nullptr, compute_done = nullptr, d2h_done = nullptr;
  bool in_use = false;  // d2h_done has been recorded at least once
};

constexpr int kTicketRing = 64;

}  // namespace

struct kpop_pipeline {
  int slot = 0;
  const kpop_twister *tw = nullptr;
  kpop_pipeline_config cfg{};
  uint32_t n_classes = 0, n_dims = 0;
  double *d_classes = nullptr, *d_metric = nullptr, *d_class_norms = nullptr;
  hipStream_t s_h2d = nullptr, s_compute = nullptr, s_d2h = nullptr;
  std::vector<RingSlot> ring;
  uint64_t next_chunk = 0;
  uint64_t next_ticket = 1;
  hipEvent_t ticket_done[kTicketRing] = {};
  uint64_t ticket_id[kTicketRing] = {};
  // statistics of the last submit (kpop_pipeline_stats)
  uint32_t last_chunks = 0;
  int last_pinned = 0;
  // record_timeline: six timing events per chunk of the last submit
  std::vector<hipEvent_t> tl_events;
  uint32_t tl_chunks = 0;
};

namespace {

struct SlotGuard {  // run on the pipeline's device slot, then go back to the caller's
  int prev;
  bool switched;
  explicit SlotGuard(int slot) : prev(current_slot()), switched(false) {
    if (slot != prev) switched = use_slot(slot) == 0;
  }
  ~SlotGuard() {
    if (switched) (void)use_slot(prev);
  }
};

static bool is_pinned(const void *p) {
  if (!p) return true;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // plain malloc'ed memory: "invalid value", not an error of ours
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

// Device -> host.  ROCm 7's runtime executes a LINEAR copy from device memory to page-locked host memory as a blit
// kernel of 16 workgroups: 36-46 GB/s beside a running kernel, on the CUs.  A RECTANGULAR copy of the same bytes goes to
// the SDMA engines: 56 GB/s whatever the CUs are doing, and nothing taken from the kernels (tools/probes/
// d2h_overlap_probe.cpp, profiles/r03_d_d2h_overlap.txt).  So a large copy to page-locked memory is issued as a
// rectangle; pageable destinations keep the linear copy (the runtime pins or stages them itself).
static int copy_down(void *dst, const void *src, uint64_t n_rows, uint64_t row_bytes, bool pinned, hipStream_t st) {
  const uint64_t bytes = n_rows * row_bytes;
  if (!bytes) return 0;
  uint64_t done = 0;
  if (pinned && bytes >= (512ull << 10)) {
    // rows of the rectangle = g matrix rows each, g the largest divisor of n_rows that keeps a rectangle row under
    // 512 KiB: the whole block goes in one rectangular copy with no linear tail (a tail is one more blit launch on
    // the download stream).  A row count with no useful divisor falls back to 256 KiB rows and a tail.
    uint64_t g = std::min<uint64_t>(n_rows, std::max<uint64_t>(1, (512ull << 10) / row_bytes));
    while (g > 1 && n_rows % g) --g;
    if (g * row_bytes >= (32ull << 10)) {
      KPOP_HIP(hipMemcpy2DAsync(dst, g * row_bytes, src, g * row_bytes, g * row_bytes, n_rows / g, hipMemcpyDeviceToHost, st));
      return 0;
    }
    constexpr uint64_t kRow = 256ull << 10;
    KPOP_HIP(hipMemcpy2DAsync(dst, kRow, src, kRow, kRow, bytes / kRow, hipMemcpyDeviceToHost, st));
    done = bytes / kRow * kRow;
  }
  if (done < bytes)
    KPOP_HIP(hipMemcpyAsync(reinterpret_cast<char *>(dst) + done, reinterpret_cast<const char *>(src) + done, bytes - done,
                            hipMemcpyDeviceToHost, st));
  return 0;
}

static void destroy(kpop_pipeline *pl) {
  if (!pl) return;
  SlotGuard g(pl->slot);
  if (pl->s_h2d) (void)hipStreamSynchronize(pl->s_h2d);
  if (pl->s_compute) (void)hipStreamSynchronize(pl->s_compute);
  if (pl->s_d2h) (void)hipStreamSynchronize(pl->s_d2h);
  for (RingSlot &s : pl->ring) {
    DevMem *all[] = {&s.bases, &s.offsets, &s.twisted, &s.dist, &s.work, &s.stats, &s.nn, &s.idx, &s.ndist, &s.z};
    for (DevMem *m : all) m->release();
    if (s.h2d_done) (void)hipEventDestroy(s.h2d_done);
    if (s.twist_done) (void)hipEventDestroy(s.twist_done);
    if (s.compute_done) (void)hipEventDestroy(s.compute_done);
    if (s.d2h_done) (void)hipEventDestroy(s.d2h_done);
  }
  for (int i = 0; i < kTicketRing; ++i)
    if (pl->ticket_done[i]) (void)hipEventDestroy(pl->ticket_done[i]);
  for (hipEvent_t e : pl->tl_events) (void)hipEventDestroy(e);
  if (pl->d_classes) (void)hipFree(pl->d_classes);
  if (pl->d_metric) (void)hipFree(pl->d_metric);
  if (pl->d_class_norms) (void)hipFree(pl->d_class_norms);
  if (pl->s_compute) {
    // the per-stream workspace of the genome kernel dies with its stream
    Context &c = ctx();
    std::lock_guard<std::mutex> lk(c.ws_mu);
    auto it = c.ws_by_stream.find(pl->s_compute);
    if (it != c.ws_by_stream.end()) {
      it->second.release();
      c.ws_by_stream.erase(it);
    }
  }
  if (pl->s_h2d) (void)hipStreamDestroy(pl->s_h2d);
  if (pl->s_compute) (void)hipStreamDestroy(pl->s_compute);
  if (pl->s_d2h) (void)hipStreamDestroy(pl->s_d2h);
  delete pl;
}

// chunk boundaries: at most chunk_reads reads and (unless one read alone is longer) chunk_bases bases
static uint32_t chunk_end(const uint64_t *offsets, uint32_t r0, uint32_t n, uint32_t chunk_reads, uint64_t chunk_bases) {
  const uint32_t hi = (uint32_t)std::min<uint64_t>(n, (uint64_t)r0 + chunk_reads);
  if (offsets[hi] - offsets[r0] <= chunk_bases) return hi;
  // first r with offsets[r] - offsets[r0] > chunk_bases; keep at least one read
  const uint64_t *e = std::upper_bound(offsets + r0, offsets + hi + 1, offsets[r0] + chunk_bases);
  const uint32_t r1 = (uint32_t)(e - offsets) - 1;
  return std::max(r1, r0 + 1);
}

}  // namespace

extern "C" int kpop_pipeline_create(const kpop_twister *tw, const double *classes, uint32_t n_classes, const double *metric,
                                    const kpop_pipeline_config *cfg, kpop_pipeline **out) {
  KPOP_TRY(require_init());
  if (!tw || !cfg || !out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_create: null argument");
  if (cfg->struct_size != sizeof(kpop_pipeline_config))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_create: config of %u bytes, this library's is %zu (set struct_size = sizeof)", cfg->struct_size,
              sizeof(kpop_pipeline_config));
  if (tw->slot != current_slot())
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_create: the twister lives on device slot %d, the calling thread works on %d", tw->slot, current_slot());
  const int outs = cfg->outputs;
  if (!outs || (outs & ~(KPOP_OUT_TWISTED | KPOP_OUT_DISTANCES | KPOP_OUT_SUMMARY)))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_create: outputs=%d selects nothing or an unknown output", outs);
  if (cfg->content != KPOP_DNA_DS && cfg->content != KPOP_DNA_SS && cfg->content != KPOP_PROTEIN)
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_create: Invalid_content(%d)", cfg->content);
  if (cfg->kind != KPOP_EUCLIDEAN && cfg->kind != KPOP_COSINE && cfg->kind != KPOP_MINKOWSKI)
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_create: unknown distance kind %d", cfg->kind);
  if (cfg->kind == KPOP_MINKOWSKI && !(cfg->p >= 0.0)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_create: negative Minkowski power");
  const bool need_classes = outs & (KPOP_OUT_DISTANCES | KPOP_OUT_SUMMARY);
  if (need_classes && (!classes || !metric || n_classes == 0))
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_create: distances / summary need the class vectors and the metric");
  kpop_pipeline *pl = new kpop_pipeline();
  pl->slot = current_slot();
  pl->tw = tw;
  pl->cfg = *cfg;
  pl->n_dims = tw->n_dims;
  pl->n_classes = need_classes ? n_classes : 0;
  if (pl->cfg.depth == 0) pl->cfg.depth = 4;
  if (pl->cfg.depth < 2) pl->cfg.depth = 2;
  if (pl->cfg.depth > 16) pl->cfg.depth = 16;
  if (pl->cfg.chunk_bases == 0) pl->cfg.chunk_bases = 256ull << 20;
  int rc = KPOP_OK;
  do {
#define PL_HIP(expr)                                                                                         \
  {                                                                                                          \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) {                                                                                  \
      set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_));                        \
      rc = KPOP_ERR_HIP;                                                                                     \
      break;                                                                                                 \
    }                                                                                                        \
  }
    PL_HIP(hipStreamCreateWithFlags(&pl->s_h2d, hipStreamNonBlocking));
    PL_HIP(hipStreamCreateWithFlags(&pl->s_compute, hipStreamNonBlocking));
    PL_HIP(hipStreamCreateWithFlags(&pl->s_d2h, hipStreamNonBlocking));
    pl->ring.resize(pl->cfg.depth);
    bool ok = true;
    for (RingSlot &s : pl->ring) {
      ok = ok && hipEventCreateWithFlags(&s.h2d_done, hipEventDisableTiming) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&s.twist_done, hipEventDisableTiming) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&s.compute_done, hipEventDisableTiming) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&s.d2h_done, hipEventDisableTiming) == hipSuccess;
    }
    for (int i = 0; i < kTicketRing; ++i) ok = ok && hipEventCreateWithFlags(&pl->ticket_done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      set_error("kpop_pipeline_create: hipEventCreate failed");
      rc = KPOP_ERR_HIP;
      break;
    }
    if (need_classes) {
      const uint64_t cb = (uint64_t)n_classes * tw->n_dims * 8, mb = (uint64_t)tw->n_dims * 8;
      PL_HIP(hipMalloc((void **)&pl->d_classes, cb));
      PL_HIP(hipMalloc((void **)&pl->d_metric, mb));
      PL_HIP(hipMemcpy(pl->d_classes, classes, cb, hipMemcpyHostToDevice));
      PL_HIP(hipMemcpy(pl->d_metric, metric, mb, hipMemcpyHostToDevice));
      // the norms of the class vectors (lib/Matrix.ml:42-76) once, not with every chunk
      if (pl->cfg.normalize_distances && n_classes < 128) {
        PL_HIP(hipMalloc((void **)&pl->d_class_norms, (uint64_t)n_classes * 8));
        if (kpop_dev_row_norms(pl->d_classes, n_classes, tw->n_dims, pl->d_metric, pl->cfg.kind, pl->cfg.p, pl->d_class_norms, nullptr) != KPOP_OK) {
          rc = KPOP_ERR_HIP;
          break;
        }
        PL_HIP(hipStreamSynchronize(nullptr));
      }
    }
#undef PL_HIP
  } while (0);
  if (rc != KPOP_OK) {
    destroy(pl);
    return rc;
  }
  *out = pl;
  return KPOP_OK;
}

extern "C" int kpop_pipeline_destroy(kpop_pipeline *pl) {
  destroy(pl);
  return KPOP_OK;
}

namespace kpop {
int launch_unpack_bases(const uint32_t *d_codes, uint32_t cshift, const uint32_t *d_invalid, uint32_t mshift, uint64_t n_bases, uint8_t *d_out, hipStream_t st);  // packed.hip
}
// bases (one byte a base) or codes + invalid (packed.hip: 2.25 bits a base, from base 0 of the batch): the same chunks either way
static int pipeline_submit_impl(kpop_pipeline *pl, const uint8_t *bases, const uint32_t *codes, const uint32_t *invalid, const uint64_t *offsets, uint32_t n_reads,
                                const kpop_pipeline_outputs *o, uint64_t *ticket) {
  if (!pl || !o || (n_reads && !offsets)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_submit: null argument");
  const bool packed = codes != nullptr;
  SlotGuard guard(pl->slot);
  const int outs = pl->cfg.outputs;
  const uint32_t D = pl->n_dims, C = pl->n_classes, mn = pl->cfg.max_neighbours;
  if (n_reads) {
    if ((outs & KPOP_OUT_TWISTED) && !o->twisted) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_submit: outputs has TWISTED but twisted is null");
    if ((outs & KPOP_OUT_DISTANCES) && !o->distances) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_submit: outputs has DISTANCES but distances is null");
    if ((outs & KPOP_OUT_SUMMARY) && (!o->stats || !o->n_neighbours || (mn && (!o->nb_index || !o->nb_distance || !o->nb_z))))
      KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_submit: outputs has SUMMARY but a summary buffer is null");
  }
  // the ticket's event slot: its previous holder (64 submits ago) must have completed
  const uint64_t tk = pl->next_ticket;
  const int ti = (int)(tk % kTicketRing);
  if (pl->ticket_id[ti]) KPOP_HIP(hipEventSynchronize(pl->ticket_done[ti]));
  uint64_t total_max_len = 0;
  for (uint32_t r = 0; r < n_reads; ++r) {
    if (offsets[r + 1] < offsets[r]) KPOP_FAIL(KPOP_ERR_INVALID, "offsets are not non-decreasing at read %u", r);
    total_max_len = std::max(total_max_len, offsets[r + 1] - offsets[r]);
  }
  if (total_max_len > 0xFFFFFFFFull) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "kpop_pipeline_submit: sequence longer than 2^32 bases");
  // chunk size: four chunks a batch.  More of them shorten the fill (first H2D + first kernel) and the drain (last D2H)
  // of a single call, but a launch of the one-wavefront-per-read kernel over fewer than ~25,000 reads no longer fills the
  // chip for long enough to reach its streaming rate (12,500 reads: 0.38 ms, against 0.15 ms as an eighth of a 100,000-read
  // launch; profiles/r03_c_pipeline_trace.txt)
  uint32_t chunk_reads = pl->cfg.chunk_reads;
  // ... and a SHORT FIRST chunk (a quarter of the others) when the size is the library's to choose: nothing comes down
  // before the first chunk has gone up and been twisted, and a download-bound batch ends that much later (0.50 ms of a
  // 2.46 ms batch with four equal chunks; profiles/r03_p_pipeline_timeline.txt)
  uint32_t first_chunk = 0, sched[4] = {0, 0, 0, 0}, n_sched = 0;
  if (chunk_reads == 0) {
    // ... unless the batch before this one is still in flight: then the batches overlap each other (this one goes up while
    // that one is twisted and comes down), and cutting this one up only costs -- every launch over a quarter of a batch
    // pays the kernel's ramp again (distances-only, ten batches of 100,000 reads in flight: 66 -> 7x M reads/s)
    const int prev = (int)((tk + kTicketRing - 1) % kTicketRing);
    const bool busy = tk > 1 && pl->ticket_id[prev] == tk - 1 && hipEventQuery(pl->ticket_done[prev]) != hipSuccess;
    if (busy) chunk_reads = std::min<uint32_t>(131072u, std::max<uint32_t>(16384u, (n_reads + 1023u) & ~1023u));
    else {
      chunk_reads = std::min<uint32_t>(131072u, std::max<uint32_t>(16384u, (div_up(n_reads, 4) + 1023u) & ~1023u));
      if (n_reads > chunk_reads) first_chunk = std::max<uint32_t>(4096u, (chunk_reads / 4) & ~1023u);
      // A lone batch of 64k-400k reads is download-bound, and what it loses is the time before the downloads start and the gaps
      // in which they wait for a chunk's kernels: chunks that GROW (11, 14, 18, 22, 35 % of the batch) keep every chunk's kernels
      // inside the download of the one before.  With kernels of 0.055 ms + 11.9 us per 1,000 reads and downloads of 0.02 ms +
      // 18.8 us per 1,000 (profiles/r03_q_pipeline_timeline.txt) the best five-chunk schedule ends at 2.25 ms for 100,000 reads;
      // a quarter-size chunk followed by four equal ones ends at 2.41 (measured: 2.44).
      if (n_reads >= 65536 && n_reads <= 400000) {
        static const uint32_t kPermille[5] = {110, 140, 180, 220, 350};
        for (int c = 0; c < 4; ++c) sched[c] = std::max<uint32_t>(4096u, (uint32_t)(((uint64_t)n_reads * kPermille[c] / 1000 + 1023u) & ~1023ull));
        n_sched = 4;  // (the fifth chunk is what is left -- all of it: 35 % of 400,000 reads is more than 131,072, and a sixth chunk of a
        chunk_reads = n_reads;  // few thousand reads would neither fill the chip nor shorten the drain)
        first_chunk = 0;
      }
    }
  }
  const bool pin_tw = is_pinned(o->twisted), pin_di = is_pinned(o->distances), pin_st = is_pinned(o->stats),
             pin_nn = is_pinned(o->n_neighbours), pin_ix = is_pinned(o->nb_index), pin_nd = is_pinned(o->nb_distance),
             pin_nz = is_pinned(o->nb_z);
  pl->last_pinned = ((packed ? is_pinned(codes) && is_pinned(invalid) : is_pinned(bases)) && is_pinned(offsets) && pin_tw && pin_di && pin_st && pin_nn && pin_ix && pin_nd && pin_nz) ? 1 : 0;
  uint32_t n_chunks = 0;
  const bool timeline = pl->cfg.record_timeline != 0;
  auto mark = [&](uint32_t chunk, int which, hipStream_t st) -> int {  // timing event `which` (0..5) of chunk
    if (!timeline || chunk >= 256) return 0;
    const size_t at = (size_t)chunk * 6 + which;
    while (pl->tl_events.size() <= at) {
      hipEvent_t e;
      KPOP_HIP(hipEventCreate(&e));
      pl->tl_events.push_back(e);
    }
    KPOP_HIP(hipEventRecord(pl->tl_events[at], st));
    return 0;
  };
  // (the chunks, as a function of its own: a failure half way -- an allocation, a launch -- must not leave the earlier chunks
  // copying into the caller's buffers behind a call that has reported an error and issued no ticket: see below)
  // the chunks' ends first: every slot is sized for the batch's LARGEST chunk, so that chunks of different sizes going round the ring
  // grow a slot once, not every time a larger chunk reaches it (growing synchronises)
  std::vector<uint32_t> ends;
  uint32_t cap_n = 0;
  uint64_t cap_nb = 0;
  for (uint32_t r0 = 0, c = 0; r0 < n_reads; ++c) {
    const uint32_t want = c < n_sched ? sched[c] : ((r0 == 0 && first_chunk) ? first_chunk : chunk_reads);
    const uint32_t r1 = chunk_end(offsets, r0, n_reads, want, pl->cfg.chunk_bases);
    ends.push_back(r1);
    cap_n = std::max(cap_n, r1 - r0);
    cap_nb = std::max(cap_nb, offsets[r1] - offsets[r0]);
    r0 = r1;
  }
  auto run_chunks = [&]() -> int {
  for (uint32_t r0 = 0; r0 < n_reads;) {
    const uint32_t r1 = ends[n_chunks];
    const uint32_t n = r1 - r0;
    const uint64_t b0 = offsets[r0], nb = offsets[r1] - b0;
    uint64_t max_len = 0;
    for (uint32_t r = r0; r < r1; ++r) max_len = std::max(max_len, offsets[r + 1] - offsets[r]);
    RingSlot &s = pl->ring[pl->next_chunk % pl->ring.size()];
    ++pl->next_chunk;
    // the slot's previous chunk must have left the device before its buffers are grown (ensure synchronises then
    // anyway) or overwritten (the stream waits below do that without stalling the host)
    KPOP_TRY(s.bases.ensure(cap_nb));
    KPOP_TRY(s.offsets.ensure((uint64_t)(cap_n + 1) * 8));
    KPOP_TRY(s.twisted.ensure((uint64_t)cap_n * D * 8));
    if (outs & KPOP_OUT_DISTANCES) KPOP_TRY(s.dist.ensure((uint64_t)cap_n * C * 8));
    if (outs & (KPOP_OUT_DISTANCES | KPOP_OUT_SUMMARY)) KPOP_TRY(s.work.ensure(kpop_dev_distance_workspace_bytes(C, cap_n, D)));
    if (outs & KPOP_OUT_SUMMARY) {
      KPOP_TRY(s.stats.ensure((uint64_t)cap_n * 4 * 8));
      KPOP_TRY(s.nn.ensure((uint64_t)cap_n * 4));
      KPOP_TRY(s.idx.ensure((uint64_t)cap_n * mn * 4));
      KPOP_TRY(s.ndist.ensure((uint64_t)cap_n * mn * 8));
      KPOP_TRY(s.z.ensure((uint64_t)cap_n * mn * 8));
    }
    // up
    // The slot's previous chunk must have come down before the slot is overwritten.  The HOST waits for that, not the
    // upload stream: an upload parked in its SDMA queue behind a stream-side wait keeps that engine from the downloads,
    // which the runtime then runs as blit kernels at two thirds of the rate, on the CUs (a batch of five chunks through a
    // ring of four: 2.95 ms instead of 2.05).  This is the pipeline's back-pressure: with `depth` chunks in flight, submit
    // waits for the oldest.
    if (s.in_use && hipEventQuery(s.d2h_done) != hipSuccess) KPOP_HIP(hipEventSynchronize(s.d2h_done));
    KPOP_TRY(mark(n_chunks, 0, pl->s_h2d));
    const uint64_t cw0 = b0 >> 4, mw0 = b0 >> 5;  // the chunk's first words of the packed batch
    if (packed && nb) {
      const uint64_t tot_c = (offsets[n_reads] + 15) >> 4, tot_m = (offsets[n_reads] + 31) >> 5;
      const uint64_t ncw = std::min(tot_c - cw0, ((b0 + nb + 15) >> 4) - cw0 + 1), nmw = std::min(tot_m - mw0, ((b0 + nb + 31) >> 5) - mw0 + 1);
      KPOP_TRY(s.pcodes.ensure(((cap_nb + 15) / 16 + 4) * 4));
      KPOP_TRY(s.pmask.ensure(((cap_nb + 31) / 32 + 4) * 4));
      KPOP_HIP(hipMemcpyAsync(s.pcodes.p, codes + cw0, ncw * 4, hipMemcpyHostToDevice, pl->s_h2d));
      KPOP_HIP(hipMemcpyAsync(s.pmask.p, invalid + mw0, nmw * 4, hipMemcpyHostToDevice, pl->s_h2d));
    } else if (nb)
      KPOP_HIP(hipMemcpyAsync(s.bases.p, bases + b0, nb, hipMemcpyHostToDevice, pl->s_h2d));
    KPOP_HIP(hipMemcpyAsync(s.offsets.p, offsets + r0, (uint64_t)(n + 1) * 8, hipMemcpyHostToDevice, pl->s_h2d));
    KPOP_TRY(mark(n_chunks, 1, pl->s_h2d));
    KPOP_HIP(hipEventRecord(s.h2d_done, pl->s_h2d));
    // count -> twist -> distance: the kernels see the caller's absolute offsets, so the bases pointer is moved back by b0
    KPOP_HIP(hipStreamWaitEvent(pl->s_compute, s.h2d_done, 0));
    KPOP_TRY(mark(n_chunks, 2, pl->s_compute));
    if (packed && nb)  // one byte a base again, at HBM's rate (the bus carried 2.25 bits of it)
      KPOP_TRY(launch_unpack_bases(s.pcodes.as<uint32_t>(), (uint32_t)(b0 & 15u), s.pmask.as<uint32_t>(), (uint32_t)(b0 & 31u), nb, s.bases.as<uint8_t>(), pl->s_compute));
    const uint8_t *d_bases = s.bases.as<uint8_t>() - b0;
    KPOP_TRY(kpop_dev_count_twist(pl->tw, d_bases, s.offsets.as<uint64_t>(), n, nb, (uint32_t)max_len, pl->cfg.content,
                                  pl->cfg.normalize_counts, s.twisted.as<double>(), pl->s_compute));
    // the twisted rows can go down while the distances are still being worked out: half of a chunk's bytes, and for the first
    // chunk of a lone batch that much less time before the downloads start
    const bool early = (outs & KPOP_OUT_TWISTED) && (outs & (KPOP_OUT_DISTANCES | KPOP_OUT_SUMMARY));
    if (early) KPOP_HIP(hipEventRecord(s.twist_done, pl->s_compute));
    if (outs & KPOP_OUT_DISTANCES)
      KPOP_TRY(kpop_dev_distance_rowwise_norms(pl->d_classes, C, pl->d_class_norms, s.twisted.as<double>(), n, D, pl->d_metric, pl->cfg.kind,
                                               pl->cfg.p, pl->cfg.normalize_distances, s.work.p, s.dist.as<double>(), pl->s_compute));
    if (outs & KPOP_OUT_SUMMARY)
      KPOP_TRY(kpop_dev_distance_summary(pl->d_classes, C, s.twisted.as<double>(), n, D, pl->d_metric, pl->cfg.kind, pl->cfg.p,
                                         pl->cfg.normalize_distances, pl->cfg.keep_at_most, mn, s.work.p, s.stats.as<double>(),
                                         s.nn.as<uint32_t>(), s.idx.as<uint32_t>(), s.ndist.as<double>(), s.z.as<double>(),
                                         pl->s_compute));
    KPOP_TRY(mark(n_chunks, 3, pl->s_compute));
    KPOP_HIP(hipEventRecord(s.compute_done, pl->s_compute));
    // down
    if (early) {
      KPOP_HIP(hipStreamWaitEvent(pl->s_d2h, s.twist_done, 0));
      KPOP_TRY(mark(n_chunks, 4, pl->s_d2h));
      KPOP_TRY(copy_down(o->twisted + (uint64_t)r0 * D, s.twisted.p, n, (uint64_t)D * 8, pin_tw, pl->s_d2h));
    }
    KPOP_HIP(hipStreamWaitEvent(pl->s_d2h, s.compute_done, 0));
    if (!early) KPOP_TRY(mark(n_chunks, 4, pl->s_d2h));
    if ((outs & KPOP_OUT_TWISTED) && !early)
      KPOP_TRY(copy_down(o->twisted + (uint64_t)r0 * D, s.twisted.p, n, (uint64_t)D * 8, pin_tw, pl->s_d2h));
    if (outs & KPOP_OUT_DISTANCES)
      KPOP_TRY(copy_down(o->distances + (uint64_t)r0 * C, s.dist.p, n, (uint64_t)C * 8, pin_di, pl->s_d2h));
    if (outs & KPOP_OUT_SUMMARY) {
      KPOP_TRY(copy_down(o->stats + (uint64_t)r0 * 4, s.stats.p, n, 32, pin_st, pl->s_d2h));
      KPOP_TRY(copy_down(o->n_neighbours + r0, s.nn.p, n, 4, pin_nn, pl->s_d2h));
      if (mn) {
        KPOP_TRY(copy_down(o->nb_index + (uint64_t)r0 * mn, s.idx.p, n, (uint64_t)mn * 4, pin_ix, pl->s_d2h));
        KPOP_TRY(copy_down(o->nb_distance + (uint64_t)r0 * mn, s.ndist.p, n, (uint64_t)mn * 8, pin_nd, pl->s_d2h));
        KPOP_TRY(copy_down(o->nb_z + (uint64_t)r0 * mn, s.z.p, n, (uint64_t)mn * 8, pin_nz, pl->s_d2h));
      }
    }
    KPOP_TRY(mark(n_chunks, 5, pl->s_d2h));
    KPOP_HIP(hipEventRecord(s.d2h_done, pl->s_d2h));
    s.in_use = true;
    ++n_chunks;
    r0 = r1;
  }
  return KPOP_OK;
  };
  const int rc = run_chunks();
  if (rc != KPOP_OK) {
    // nothing of this batch is in flight when the error goes back: the caller may free or reuse its buffers
    const std::string msg = get_error();
    (void)hipStreamSynchronize(pl->s_h2d);
    (void)hipStreamSynchronize(pl->s_compute);
    (void)hipStreamSynchronize(pl->s_d2h);
    set_error("%s", msg.c_str());
    return rc;
  }
  pl->last_chunks = n_chunks;
  pl->tl_chunks = timeline ? std::min<uint32_t>(n_chunks, 256) : 0;
  KPOP_HIP(hipEventRecord(pl->ticket_done[ti], pl->s_d2h));
  pl->ticket_id[ti] = tk;
  ++pl->next_ticket;
  if (ticket) *ticket = tk;
  return KPOP_OK;
}

extern "C" int kpop_pipeline_submit(kpop_pipeline *pl, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                                    const kpop_pipeline_outputs *o, uint64_t *ticket) {
  return pipeline_submit_impl(pl, bases, nullptr, nullptr, offsets, n_reads, o, ticket);
}

extern "C" int kpop_pipeline_submit_packed(kpop_pipeline *pl, const uint32_t *codes, const uint32_t *invalid, const uint64_t *offsets, uint32_t n_reads,
                                           const kpop_pipeline_outputs *o, uint64_t *ticket) {
  if (n_reads && offsets && offsets[n_reads] && (!codes || !invalid)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_submit_packed: null argument");
  static const uint32_t none = 0;
  return pipeline_submit_impl(pl, nullptr, codes ? codes : &none, invalid ? invalid : &none, offsets, n_reads, o, ticket);
}

extern "C" int kpop_pipeline_collect(kpop_pipeline *pl, uint64_t ticket) {
  if (!pl) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_collect: null pipeline");
  if (ticket == 0 || ticket >= pl->next_ticket) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_collect: ticket %llu was never issued", (unsigned long long)ticket);
  SlotGuard guard(pl->slot);
  const int ti = (int)(ticket % kTicketRing);
  // an older ticket whose event slot has been handed on completed before that hand-over (submit waits for it)
  if (pl->ticket_id[ti] == ticket) KPOP_HIP(hipEventSynchronize(pl->ticket_done[ti]));
  return KPOP_OK;
}

extern "C" int kpop_pipeline_run(kpop_pipeline *pl, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                                 const kpop_pipeline_outputs *o) {
  uint64_t tk = 0;
  KPOP_TRY(kpop_pipeline_submit(pl, bases, offsets, n_reads, o, &tk));
  return kpop_pipeline_collect(pl, tk);
}

extern "C" int kpop_pipeline_timeline(kpop_pipeline *pl, uint32_t max_chunks, double *ms, uint32_t *n_chunks) {
  if (!pl || !n_chunks || (max_chunks && !ms)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_timeline: null argument");
  if (!pl->cfg.record_timeline) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_timeline: the pipeline was created without record_timeline");
  SlotGuard guard(pl->slot);
  const uint32_t n = std::min(pl->tl_chunks, max_chunks);
  for (uint32_t c = 0; c < n; ++c)
    for (int w = 0; w < 6; ++w) {
      float t = 0.f;
      KPOP_HIP(hipEventSynchronize(pl->tl_events[(size_t)c * 6 + w]));
      KPOP_HIP(hipEventElapsedTime(&t, pl->tl_events[0], pl->tl_events[(size_t)c * 6 + w]));
      ms[(size_t)c * 6 + w] = (double)t;
    }
  *n_chunks = n;
  return KPOP_OK;
}

extern "C" int kpop_pipeline_stats(const kpop_pipeline *pl, uint32_t *chunks, int *pinned, uint32_t *depth) {
  if (!pl) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_pipeline_stats: null pipeline");
  if (chunks) *chunks = pl->last_chunks;
  if (pinned) *pinned = pl->last_pinned;
  if (depth) *depth = pl->cfg.depth;
  return KPOP_OK;
}

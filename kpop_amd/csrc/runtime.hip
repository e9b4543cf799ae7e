// runtime.hip -- process context, error reporting, device-memory helpers and
// the O(n_dims) metric computation.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <algorithm>

#include "common.h"

namespace kpop {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char *get_error() { return g_err; }

static Context g_ctx[kMaxSlots];
static int g_n_slots = 0;
static thread_local int tl_slot = 0;
static thread_local int tl_arena_scopes = 0;
int &arena_scopes_of_this_thread() { return tl_arena_scopes; }

Context &ctx() { return g_ctx[tl_slot]; }
Context &ctx_of(int slot) { return g_ctx[slot]; }
int current_slot() { return tl_slot; }
int n_slots() { return g_n_slots; }

int use_slot(int slot) {
  if (slot < 0 || slot >= g_n_slots || !g_ctx[slot].initialised)
    KPOP_FAIL(KPOP_ERR_INVALID, "kpop_use_device: slot %d is not one of the %d initialised", slot, g_n_slots);
  KPOP_HIP(hipSetDevice(g_ctx[slot].device));
  tl_slot = slot;
  return 0;
}

int Workspace::ensure(uint64_t need, void **out) {
  if (need > bytes) {
    if (p) {
      KPOP_HIP(hipDeviceSynchronize());  // earlier launches may still use the old block
      KPOP_HIP(hipFree(p));
      p = nullptr;
      bytes = 0;
    }
    uint64_t want = need + need / 4 + 4096;
    KPOP_HIP(hipMalloc(&p, want));
    bytes = want;
  }
  *out = p;
  return 0;
}

int Arena::take(uint64_t n, void **out) {
  n = (n + 255) & ~255ull;
  for (;;) {
    if (cur < chunks.size()) {
      if (off + n <= chunks[cur].bytes) {
        *out = reinterpret_cast<char *>(chunks[cur].p) + off;
        off += n;
        return 0;
      }
      ++cur;  // the rest of this chunk stays unused until the scope unwinds
      off = 0;
      continue;
    }
    Chunk c;
    c.bytes = std::max<uint64_t>(n, 64ull << 20);
    c.p = nullptr;
    KPOP_HIP(hipMalloc(&c.p, c.bytes));
    chunks.push_back(c);
  }
}

void Arena::release() {
  for (Chunk &c : chunks) (void)hipFree(c.p);
  chunks.clear();
  cur = 0;
  off = 0;
}

void Workspace::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  bytes = 0;
}

int require_init() {
  if (!ctx().initialised)
    KPOP_FAIL(KPOP_ERR_NOT_INIT, "libkpop_hip: kpop_init(device) has not been called (or failed: no usable GPU)");
  return 0;
}

}  // namespace kpop

using namespace kpop;

extern "C" const char *kpop_last_error(void) { return get_error(); }
extern "C" const char *kpop_version(void) { return "kpop_hip 0.1 (gfx950)"; }

extern "C" int kpop_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) KPOP_FAIL(KPOP_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

int Context::aux_for(hipStream_t st, AuxLane **out) {
  std::lock_guard<std::mutex> g(ws_mu);
  AuxLane &a = aux_by_stream[st];
  if (!a.stream) {
    KPOP_HIP(hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking));
    for (hipEvent_t *e : {&a.fork, &a.join, &a.step[0], &a.step[1]}) KPOP_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  *out = &a;
  return 0;
}

static void release_slot(Context &c) {
  if (!c.initialised) return;
  (void)hipSetDevice(c.device);
  {
    std::lock_guard<std::mutex> g(c.ws_mu);
    for (auto &kv : c.ws_by_stream) kv.second.release();
    c.ws_by_stream.clear();
    for (auto &kv : c.ws2_by_stream) kv.second.release();
    c.ws2_by_stream.clear();
    for (auto &kv : c.aux_by_stream) {
      Context::AuxLane &a = kv.second;
      if (a.stream) (void)hipStreamSynchronize(a.stream);
      for (hipEvent_t e : {a.fork, a.join, a.step[0], a.step[1]})
        if (e) (void)hipEventDestroy(e);
      if (a.stream) (void)hipStreamDestroy(a.stream);
    }
    c.aux_by_stream.clear();
  }
  c.arena.release();
  c.initialised = false;
  ++c.generation;
}

extern "C" int kpop_init_devices(const int *devices, int n) {
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0)
    KPOP_FAIL(KPOP_ERR_HIP, "kpop_init: no HIP device visible (%s); this library has no CPU path",
              e == hipSuccess ? "0 devices" : hipGetErrorString(e));
  if (n < 1 || n > kMaxSlots) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_init_devices: %d devices (1..%d)", n, kMaxSlots);
  if (!devices) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_init_devices: null device list");
  for (int i = 0; i < n; ++i)
    if (devices[i] < 0 || devices[i] >= n_dev)
      KPOP_FAIL(KPOP_ERR_INVALID, "kpop_init: device %d out of range 0..%d", devices[i], n_dev - 1);
  stop_slot_workers();
  for (int i = n; i < g_n_slots; ++i) release_slot(g_ctx[i]);
  for (int i = 0; i < n; ++i) {
    Context &c = g_ctx[i];
    if (c.initialised && c.device != devices[i]) release_slot(c);
    KPOP_HIP(hipSetDevice(devices[i]));
    hipDeviceProp_t prop;
    KPOP_HIP(hipGetDeviceProperties(&prop, devices[i]));
    c.slot = i;
    c.device = devices[i];
    c.n_cus = prop.multiProcessorCount;
    c.lds_per_block = prop.sharedMemPerBlock;
    c.initialised = true;
    if (const char *dbg = getenv("KPOP_TUNE_DBG")) c.tune_dbg = (uint32_t)atoi(dbg);  // (the A/B switches of kpop_tune("dbg"), for the command-line tools)
  }
  g_n_slots = n;
  // every pair of distinct devices may copy to each other directly (the in-library all-gather, multi.hip)
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      if (devices[i] == devices[j]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) {
        (void)hipSetDevice(devices[i]);
        hipError_t pe = hipDeviceEnablePeerAccess(devices[j], 0);
        if (pe != hipSuccess) (void)hipGetLastError();  // already enabled, or refused: copies then go through the host
      }
    }
  tl_slot = 0;
  KPOP_HIP(hipSetDevice(devices[0]));
  return KPOP_OK;
}

extern "C" int kpop_init(int device) { return kpop_init_devices(&device, 1); }

extern "C" int kpop_use_device(int slot) { return use_slot(slot); }
extern "C" int kpop_device_slots(void) { return g_n_slots; }

extern "C" int kpop_shutdown(void) {
  stop_slot_workers();
  for (int i = 0; i < g_n_slots; ++i) release_slot(g_ctx[i]);
  g_n_slots = 0;
  tl_slot = 0;
  return KPOP_OK;
}

extern "C" int kpop_tune(const char *key, int value) {
  if (!key) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_tune: null key");
  for (int i = 0; i < kMaxSlots; ++i) {  // a knob holds for every device slot
    Context &c = g_ctx[i];
    if (!strcmp(key, "unroll") && (value == 8 || value == 16)) c.tune_unroll = value;
    else if (!strcmp(key, "nt") && (value == 0 || value == 1 || value == 2)) c.tune_nt = value;
    else if (!strcmp(key, "dense") && value >= 0 && value <= 2) c.tune_dense = value;
    else if (!strcmp(key, "dbg")) c.tune_dbg = value;
    else if (!strcmp(key, "tileg") && (value == 32 || value == 64)) c.tune_tileg = value;
    else if (!strcmp(key, "tilepipe") && (value == 0 || value == 1)) c.tune_tilepipe = value;
    else if (!strcmp(key, "tilewide") && (value == 0 || value == 1)) c.tune_tilewide = value;
    else if (!strcmp(key, "tilecap_mb") && value >= 0) c.tune_tilecap_mb = value;
    else if (!strcmp(key, "pipeprio") && value >= 0 && value <= 7) c.tune_pipeprio = value;
    else if (!strcmp(key, "blocksort") && (value == 0 || value == 1)) c.tune_blocksort = value;
    else if (!strcmp(key, "ldspad") && value >= 0 && value <= 65536) c.tune_ldspad = value;
    else if (!strcmp(key, "hist") && (value == 0 || value == 1)) c.tune_hist = value;
    else if (!strcmp(key, "histlds") && value >= 0 && value <= 4) c.tune_histlds = value;
    else if (!strcmp(key, "histguess") && (value == 0 || value == 1)) c.tune_histguess = value;
    else if (!strcmp(key, "direct") && value >= 0 && value <= 2) c.tune_direct = value;
    else if (!strcmp(key, "summary_mfma_lists") && (value == 0 || value == 1)) c.tune_summary_mfma_lists = value;
    else if (!strcmp(key, "distance_mfma") && (value == 0 || value == 1)) c.tune_distance_mfma = value;
    else if (!strcmp(key, "summary_mfma") && value >= 0 && value <= 2) c.tune_summary_mfma = value;
    else if (!strcmp(key, "summary2") && value >= 0 && value <= 3) c.tune_summary2 = value;
    else if (!strcmp(key, "summary_lanes") && (value == 1 || value == 2)) c.tune_summary_lanes = value;
    else if (!strcmp(key, "summary_audit") && (value == 0 || value == 1)) c.tune_summary_audit = value;
    else if (!strcmp(key, "summary_pass") && (value == 0 || value == 1)) c.tune_summary_pass = value;
    else if (!strcmp(key, "summary_rawref") && (value == 0 || value == 1)) c.tune_summary_rawref = value;
    else if (!strcmp(key, "summary_sample") && (value == 0 || value == 1)) c.tune_summary_sample = value;
    else if (!strcmp(key, "seg") && (value == 0 || (value >= 64 && value <= 16384 && value % 64 == 0))) c.tune_seg = value;
    else KPOP_FAIL(KPOP_ERR_INVALID, "kpop_tune: unknown knob or value %s=%d", key, value);
  }
  return KPOP_OK;
}

extern "C" int kpop_synchronize(void *stream) {
  KPOP_TRY(require_init());
  KPOP_HIP(hipStreamSynchronize(as_stream(stream)));
  return KPOP_OK;
}

extern "C" int kpop_dev_workspace_reserve(uint64_t bytes) { return kpop_dev_workspace_reserve_stream(bytes, nullptr); }

extern "C" int kpop_dev_workspace_reserve_stream(uint64_t bytes, void *stream) {
  KPOP_TRY(require_init());
  void *p = nullptr;
  return ctx().ws_for(as_stream(stream)).ensure(bytes, &p);
}

// page-locked host memory: what the streaming pipeline (pipeline.hip) copies to and from at the bus rate, with the
// copy engines running beside the kernels
extern "C" int kpop_host_alloc(void **ptr, uint64_t bytes) {
  KPOP_TRY(require_init());
  if (!ptr) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_host_alloc: null ptr");
  KPOP_HIP(hipHostMalloc(ptr, bytes ? bytes : 8, hipHostMallocPortable));
  return KPOP_OK;
}

extern "C" int kpop_host_free(void *ptr) {
  if (ptr) KPOP_HIP(hipHostFree(ptr));
  return KPOP_OK;
}

extern "C" int kpop_host_register(void *ptr, uint64_t bytes) {
  KPOP_TRY(require_init());
  if (!ptr || !bytes) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_host_register: null or empty range");
  KPOP_HIP(hipHostRegister(ptr, bytes, hipHostRegisterPortable));
  return KPOP_OK;
}

extern "C" int kpop_host_unregister(void *ptr) {
  if (ptr) KPOP_HIP(hipHostUnregister(ptr));
  return KPOP_OK;
}

extern "C" int kpop_dev_malloc(void **ptr, uint64_t bytes) {
  KPOP_TRY(require_init());
  if (!ptr) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_malloc: null ptr");
  KPOP_HIP(hipMalloc(ptr, bytes ? bytes : 8));
  return KPOP_OK;
}

extern "C" int kpop_dev_free(void *ptr) {
  if (ptr) KPOP_HIP(hipFree(ptr));
  return KPOP_OK;
}

extern "C" int kpop_memcpy_h2d(void *dst, const void *src, uint64_t bytes) {
  KPOP_TRY(require_init());
  if (bytes) KPOP_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return KPOP_OK;
}

extern "C" int kpop_memcpy_d2h(void *dst, const void *src, uint64_t bytes) {
  KPOP_TRY(require_init());
  if (bytes) KPOP_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return KPOP_OK;
}

extern "C" int kpop_dev_memset(void *dst, int value, uint64_t bytes) {
  KPOP_TRY(require_init());
  if (bytes) KPOP_HIP(hipMemset(dst, value, bytes));
  return KPOP_OK;
}

// Space.Distance.Metric.compute, lib/Space.ml:88-105.  `Powers` delegates to
// BiOCamLib Numbers.Frequencies.Vector (absent from the reference checkout):
// pow_abs pi |> threshold_accum_abs thr |> pow_abs pe |> normalize_abs, with
// the semantics declared in DESIGN.md.  n_dims values: host arithmetic.
extern "C" int kpop_metric_compute(int metric_kind, const double *inertia, uint32_t n_dims, double power_int,
                                   double threshold, double power_ext, double *out) {
  if (!out && n_dims) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_metric_compute: null out");
  if (metric_kind == KPOP_METRIC_FLAT) {  // :89-95
    for (uint32_t i = 0; i < n_dims; ++i) out[i] = 1.0 / (double)n_dims;
    return KPOP_OK;
  }
  if (metric_kind != KPOP_METRIC_POWERS) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_metric_compute: unknown metric %d", metric_kind);
  if (!inertia && n_dims) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_metric_compute: null inertia");
  // lib/Space.ml:124-129 (of_string) rejects these
  if (power_int < 0.0 || power_ext < 0.0) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_metric_compute: negative power");
  if (threshold < 0.0 || threshold > 1.0) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_metric_compute: threshold outside [0,1]");
  double total = 0.0;
  for (uint32_t i = 0; i < n_dims; ++i) {
    out[i] = pow(fabs(inertia[i]), power_int);
    total += fabs(out[i]);
  }
  double run = 0.0;
  const double limit = threshold * total;
  for (uint32_t i = 0; i < n_dims; ++i) {
    const double a = fabs(out[i]);
    if (run >= limit) out[i] = 0.0;
    run += a;
  }
  double s = 0.0;
  for (uint32_t i = 0; i < n_dims; ++i) {
    out[i] = pow(fabs(out[i]), power_ext);
    s += fabs(out[i]);
  }
  if (s != 0.0)
    for (uint32_t i = 0; i < n_dims; ++i) out[i] = out[i] / s;
  return KPOP_OK;
}

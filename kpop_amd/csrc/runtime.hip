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

Context &ctx() {
  static Context c;
  return c;
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

extern "C" int kpop_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    KPOP_FAIL(KPOP_ERR_HIP, "kpop_init: no HIP device visible (%s); this library has no CPU path",
              e == hipSuccess ? "0 devices" : hipGetErrorString(e));
  if (device < 0 || device >= n) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_init: device %d out of range 0..%d", device, n - 1);
  KPOP_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  KPOP_HIP(hipGetDeviceProperties(&prop, device));
  Context &c = ctx();
  c.device = device;
  c.n_cus = prop.multiProcessorCount;
  c.lds_per_block = prop.sharedMemPerBlock;
  c.initialised = true;
  if (const char *dbg = getenv("KPOP_TUNE_DBG")) c.tune_dbg = (uint32_t)atoi(dbg);  // (the A/B switches of kpop_tune("dbg"), for the command-line tools)
  return KPOP_OK;
}

extern "C" int kpop_shutdown(void) {
  ctx().ws.release();
  ctx().arena.release();
  ctx().initialised = false;
  return KPOP_OK;
}

extern "C" int kpop_tune(const char *key, int value) {
  if (!key) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_tune: null key");
  Context &c = ctx();
  if (!strcmp(key, "unroll") && (value == 8 || value == 16)) c.tune_unroll = value;
  else if (!strcmp(key, "nt") && (value == 0 || value == 1 || value == 2)) c.tune_nt = value;
  else if (!strcmp(key, "dense") && value >= 0 && value <= 2) c.tune_dense = value;
  else if (!strcmp(key, "dbg")) c.tune_dbg = value;
  else if (!strcmp(key, "blocksort") && (value == 0 || value == 1)) c.tune_blocksort = value;
  else if (!strcmp(key, "ldspad") && value >= 0 && value <= 65536) c.tune_ldspad = value;
  else if (!strcmp(key, "hist") && (value == 0 || value == 1)) c.tune_hist = value;
  else if (!strcmp(key, "seg") && (value == 0 || (value >= 64 && value <= 16384 && value % 64 == 0))) c.tune_seg = value;
  else KPOP_FAIL(KPOP_ERR_INVALID, "kpop_tune: unknown knob or value %s=%d", key, value);
  return KPOP_OK;
}

extern "C" int kpop_synchronize(void *stream) {
  KPOP_TRY(require_init());
  KPOP_HIP(hipStreamSynchronize(as_stream(stream)));
  return KPOP_OK;
}

extern "C" int kpop_dev_workspace_reserve(uint64_t bytes) {
  KPOP_TRY(require_init());
  void *p = nullptr;
  return ctx().ws.ensure(bytes, &p);
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

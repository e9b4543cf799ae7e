// d2h_overlap_probe.cpp -- what does a device-to-host copy cost the HBM-bound count->twist kernel running beside it?
// Times kpop_dev_count_twist (100k x 150 bp, k=12, D=64) alone, then with a stream of 52 MB device-to-host copies
// running concurrently by several routes:
//   hip-pinned   hipMemcpyAsync to hipHostMalloc'ed memory   (ROCm 7's CLR runs this as a blit KERNEL, see the trace)
//   hip-pageable hipMemcpyAsync to malloc'ed memory
//   hip-2d       hipMemcpy2DAsync to pinned memory
//   own-N        this file's copy kernel with N workgroups storing straight to mapped pinned memory
//   hsa-sdma     hsa_amd_memory_async_copy (the SDMA engines), bypassing HIP
// Build: hipcc -O2 --offload-arch=gfx950 -I include tools/probes/d2h_overlap_probe.cpp -L kpop_amd -lkpop_hip -lhsa-runtime64 -Wl,-rpath,$PWD/kpop_amd -o d2h_probe
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "kpop_hip.h"

#define CK(e)                                                                         \
  do {                                                                                \
    hipError_t e_ = (e);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)
#define KP(e)                                                             \
  do {                                                                    \
    if ((e) != 0) {                                                       \
      fprintf(stderr, "%s -> %s\n", #e, kpop_last_error());              \
      exit(1);                                                            \
    }                                                                     \
  } while (0)

__global__ void own_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

static hsa_agent_t g_gpu, g_cpu;
static hsa_status_t find_agents(hsa_agent_t a, void *) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_gpu.handle) g_gpu = a;
  if (t == HSA_DEVICE_TYPE_CPU && !g_cpu.handle) g_cpu = a;
  return HSA_STATUS_SUCCESS;
}

int main(int argc, char **argv) {
  const uint32_t n = 100000, L = 150, D = 64;
  const int k = 12, reps = 20;
  KP(kpop_init(0));
  kpop_twister *tw = nullptr;
  KP(kpop_twister_synth(0x5EED, k, KPOP_DNA_DS, D, &tw));
  uint8_t *d_bases;
  uint64_t *d_off;
  double *d_out, *d_src;
  const size_t copy_bytes = 52u << 20;
  CK(hipMalloc(&d_bases, (size_t)n * L));
  CK(hipMalloc(&d_off, (n + 1) * 8));
  CK(hipMalloc(&d_out, (size_t)n * D * 8));
  CK(hipMalloc(&d_src, copy_bytes));
  CK(hipMemset(d_src, 1, copy_bytes));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  KP(kpop_dev_synth_reads(1, n, L, 0, d_bases, d_off, sa));
  void *h_pinned, *h_page = malloc(copy_bytes);
  CK(hipHostMalloc(&h_pinned, copy_bytes, hipHostMallocPortable | hipHostMallocMapped));
  memset(h_page, 0, copy_bytes);
  void *h_pinned_dev = nullptr;
  CK(hipHostGetDevicePointer(&h_pinned_dev, h_pinned, 0));
  hsa_init();
  hsa_iterate_agents(find_agents, nullptr);
  hsa_signal_t sig;
  hsa_signal_create(1, 0, nullptr, &sig);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto ct = [&]() { KP(kpop_dev_count_twist(tw, d_bases, d_off, n, (uint64_t)n * L, L, KPOP_DNA_DS, 1, d_out, sa)); };
  for (int i = 0; i < 5; ++i) ct();
  CK(hipDeviceSynchronize());
  const char *modes[] = {"none", "hip-pinned", "hip-pageable", "hip-2d", "own-8", "own-16", "own-64", "own-256", "hsa-sdma"};
  for (const char *mode : modes) {
    if (argc > 1 && strcmp(argv[1], mode)) continue;
    // the copies: issued from this thread, enough of them to outlast the timed kernels
    int n_copies = !strcmp(mode, "none") ? 0 : 40;
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    std::vector<hsa_signal_t> sigs;
    // kernels first (async), then copies (some routes block the host), kernels are timed by events
    CK(hipEventRecord(e0, sa));
    for (int i = 0; i < reps; ++i) ct();
    CK(hipEventRecord(e1, sa));
    for (int c = 0; c < n_copies; ++c) {
      if (!strcmp(mode, "hip-pinned")) CK(hipMemcpyAsync(h_pinned, d_src, copy_bytes, hipMemcpyDeviceToHost, sb));
      else if (!strcmp(mode, "hip-pageable")) CK(hipMemcpyAsync(h_page, d_src, copy_bytes, hipMemcpyDeviceToHost, sb));
      else if (!strcmp(mode, "hip-2d")) CK(hipMemcpy2DAsync(h_pinned, 1 << 20, d_src, 1 << 20, 1 << 20, copy_bytes >> 20, hipMemcpyDeviceToHost, sb));
      else if (!strncmp(mode, "own-", 4)) own_copy_kernel<<<atoi(mode + 4), 256, 0, sb>>>((const uint4 *)d_src, (uint4 *)h_pinned_dev, copy_bytes / 16);
      else if (!strcmp(mode, "hsa-sdma")) {
        hsa_signal_t s;
        hsa_signal_create(1, 0, nullptr, &s);
        hsa_status_t st = hsa_amd_memory_async_copy(h_pinned, g_cpu, d_src, g_gpu, copy_bytes, 0, nullptr, s);
        if (st != HSA_STATUS_SUCCESS) {
          fprintf(stderr, "hsa_amd_memory_async_copy -> %d\n", (int)st);
          exit(1);
        }
        sigs.push_back(s);
      }
    }
    CK(hipEventSynchronize(e1));
    auto t1 = std::chrono::steady_clock::now();
    // how many copies had completed by the time the kernels were done?  (approximate: ask now)
    CK(hipStreamSynchronize(sb));
    for (hsa_signal_t s : sigs) {
      hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
      hsa_signal_destroy(s);
    }
    auto t2 = std::chrono::steady_clock::now();
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double all = std::chrono::duration<double>(t2 - t0).count();
    printf("%-13s count_twist %.3f ms/launch (%d launches, host saw them done after %.2f ms); %d copies of %zu MiB all done after %.2f ms -> %.1f GB/s if the bus never idled\n",
           mode, ms / reps, reps, std::chrono::duration<double>(t1 - t0).count() * 1e3, n_copies, copy_bytes >> 20, all * 1e3,
           n_copies ? n_copies * (double)copy_bytes / all / 1e9 : 0.0);
    fflush(stdout);
  }
  return 0;
}

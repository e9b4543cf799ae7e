// Where do the wavefronts of a 1,024-thread block land?  (HW_REG_HW_ID: wave, SIMD, CU, SE of every wavefront; gfx9 register 4)
// hipcc --offload-arch=gfx950 -O2 -o tools/probes/simd_probe tools/probes/src/simd_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void probe(uint32_t *out, int lds_words) {
  extern __shared__ uint32_t pad[];
  if (lds_words && threadIdx.x == 0) pad[lds_words - 1] = 0;
  const uint32_t id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));  // HW_ID, all 32 bits
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
int main() {
  for (int threads : {1024, 768, 512}) {
    const int blocks = 8, waves = threads / 64;
    uint32_t *d;
    hipMalloc(&d, blocks * waves * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    probe<<<blocks, threads, 150 * 1024>>>(d, 150 * 256);
    std::vector<uint32_t> h(blocks * waves);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    printf("threads %d (one block a CU: 150 KB of LDS)\n", threads);
    for (int b = 0; b < blocks; ++b) {
      printf(" block %d:", b);
      for (int w = 0; w < waves; ++w) {
        const uint32_t v = h[b * waves + w];
        printf(" w%d:simd%u/slot%u/cu%u", w, (v >> 4) & 3, v & 15, (v >> 8) & 15);
      }
      printf("\n");
    }
    hipFree(d);
  }
  return 0;
}

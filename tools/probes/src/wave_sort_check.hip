// wave_sort_check.hip -- the one-wavefront networks of kpop_amd/csrc/wave_sort.h against std::sort: 64 R keys a wavefront
// (R = 1..16, 32- and 64-bit keys, many ties), and every lane exchange (lane ^ 1 ... lane ^ 32) against its definition.
// Built and run on the GPU box by tests/test_gpu_wave_sort.py:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -o check wave_sort_check.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../../../kpop_amd/csrc/wave_sort.h"
using namespace kpop;
template <int R, typename K> __global__ void k(const K *in, K *out) {
  const int lane = threadIdx.x;
  K key[R];
  for (int r = 0; r < R; ++r) key[r] = in[blockIdx.x * 64 * R + lane * R + r];
  wave_bitonic_sort<R, K>(key, lane);
  for (int r = 0; r < R; ++r) out[blockIdx.x * 64 * R + lane * R + r] = key[r];
}
template <int R, typename K> int run() {
  const int B = 64, N = 64 * R;
  std::vector<K> h(B * N), o(B * N);
  for (auto &x : h) x = (K)(((uint64_t)rand() << 20) ^ rand()) % (K)(rand() % 3 ? 1000 : ~(K)0);
  K *di, *dout; hipMalloc(&di, sizeof(K) * B * N); hipMalloc(&dout, sizeof(K) * B * N);
  hipMemcpy(di, h.data(), sizeof(K) * B * N, hipMemcpyHostToDevice);
  k<R, K><<<B, 64>>>(di, dout); hipMemcpy(o.data(), dout, sizeof(K) * B * N, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < B; ++b) { std::sort(h.begin() + b * N, h.begin() + (b + 1) * N); for (int i = 0; i < N; ++i) bad += h[b * N + i] != o[b * N + i]; }
  printf("R=%d K=%zu bad %d\n", R, sizeof(K), bad); return bad;
}
__global__ void xk(uint32_t *out) {
  const int lane = threadIdx.x;
  const uint32_t v = lane * 3 + 1;
#pragma unroll
  for (int m = 1; m <= 32; m <<= 1) out[(31 - __builtin_clz(m)) * 64 + lane] = wave_shfl_xor(v, m);
}
static int exchanges() {
  uint32_t *d, h[6 * 64];
  hipMalloc(&d, sizeof h); xk<<<1, 64>>>(d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int s = 0; s < 6; ++s) for (int l = 0; l < 64; ++l) bad += h[s * 64 + l] != (uint32_t)((l ^ (1 << s)) * 3 + 1);
  printf("exchanges bad %d\n", bad); return bad;
}
int main() { int bad = exchanges() + run<1, uint32_t>() + run<2, uint32_t>() + run<4, uint32_t>() + run<8, uint32_t>() + run<16, uint32_t>() + run<4, uint64_t>() + run<1, uint64_t>(); return bad != 0; }

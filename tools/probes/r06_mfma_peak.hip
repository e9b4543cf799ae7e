// round 6: what the f64 matrix pipe delivers when nothing else is asked of it -- every SIMD issuing independent v_mfma_f64_16x16x4_f64 back to
// back (W wavefronts a SIMD, 16 accumulator tiles a wavefront), no memory traffic; and the shader clock it ran at (cycle counter against
// the 100 MHz wall clock).  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/probes/r06_mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
using f64x4 = __attribute__((ext_vector_type(4))) double;
__global__ __launch_bounds__(256, 2) void peak_kernel(double *out, int iters, unsigned long long *clk) {
  f64x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  double s = 0.0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}
int main() {
  double *out;
  unsigned long long *clk, h[2];
  hipMalloc(&out, 8ull * 256 * 4096);
  hipMalloc(&clk, 16);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  for (int waves = 1; waves <= 2; ++waves) {  // wavefronts a SIMD (a block of 256 threads is one wavefront on each SIMD of a CU)
    const int blocks = cus * waves, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    peak_kernel<<<blocks, 256>>>(out, 1000, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    peak_kernel<<<blocks, 256>>>(out, iters, clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = 2.0 * 16 * 16 * 4 * 16.0 * iters * 4.0 * blocks;
    printf("%d CUs, %d wavefront(s) a SIMD: %.3f ms, %.2f TFLOP/s f64; shader clock %.0f MHz over the kernel (cycle counter / 100 MHz wall clock); %.1f cycles an MFMA a SIMD\n", cus, waves, ms,
           flops / ms / 1e9, (double)h[0] / (double)h[1] * 100.0, (double)h[0] / (16.0 * iters * waves));
  }
  return 0;
}

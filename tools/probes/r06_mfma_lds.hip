// round 6: the inner loop of the distance GEMM on its own -- per step of 4 dimensions 4 + 4 fragments read from LDS (ds_read_b64, the
// panels' layout: [dimension][row], stride 145 doubles) and 16 v_mfma_f64_16x16x4_f64 on them, the next step's fragments read while this
// step's MFMAs run; no global memory, no barrier.  What fraction of the pure-MFMA rate survives the LDS reads?
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_lds tools/probes/r06_mfma_lds.hip && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr int kDK = 16, kDS = 145;
template <int PF>
__global__ __launch_bounds__(256, 2) void loop_kernel(double *out, int chunks) {
  extern __shared__ double lds[];
  double (*Qs)[kDS] = reinterpret_cast<double (*)[kDS]>(lds);
  double (*Rs)[kDS] = reinterpret_cast<double (*)[kDS]>(lds + kDK * kDS);
  for (int i = threadIdx.x; i < 2 * kDK * kDS; i += 256) lds[i] = 1.0 + i * 1e-9;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
  f64x4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
  for (int c = 0; c < chunks; ++c) {
#pragma unroll
    for (int ks = 0; ks < kDK; ks += 4) {
      double fa[4], fb[4];
      const int kr = ks + (lane >> 4), cc = lane & 15;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = Qs[kr][wm + t * 16 + cc];
        fb[t] = Rs[kr][wn + t * 16 + cc];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (PF) __builtin_amdgcn_sched_barrier(0);
  }
  double s = 0.0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  double *out;
  hipMalloc(&out, 8ull * 256 * 4096);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, chunks = 4000;
  const size_t lds = 2 * kDK * kDS * 8;
  for (int waves = 1; waves <= 2; ++waves) {
    const int blocks = cus * waves;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    loop_kernel<0><<<blocks, 256, lds>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop_kernel<0><<<blocks, 256, lds>>>(out, chunks);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * 64.0 * chunks * 4.0 * blocks;
    printf("%d wavefront(s) a SIMD, LDS fragments + MFMAs: %.3f ms, %.2f TFLOP/s f64 = %.3f of 78.6\n", waves, ms, flops / ms / 1e9, flops / ms / 1e9 / 78.6);
  }
  return 0;
}

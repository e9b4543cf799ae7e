// gemm_f64.h -- f64 GEMM on the matrix cores (v_mfma_f64_16x16x4_f64), for the one genuinely dense
// contraction of KPop: the correspondence analysis behind KPopTwist (S'S over millions of k-mer rows, then
// S * W).  C[M x N] = op(A) * B with B stored [K][N]; op(A) is A' for A stored [K][M] (TRANS_A) or A for A
// stored [M][K].
//
// Block = 256 threads = 4 waves in a 2 x 2 arrangement, output tile 128 x 128, each wave 64 x 64 = 4 x 4 MFMA
// tiles (16 accumulators of 4 f64).  K is consumed 16 at a time: both panels are staged global -> registers ->
// LDS ([k][128+pad] so a fragment read is one conflict-free ds_read_b64 per lane), software-pipelined so the
// loads of chunk c+1 fly under the 64 MFMAs of chunk c.  MFMA operand maps (cdna_hip_programming.md section 3):
// lane l gives A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; it receives D[row = (l>>4) + 4*reg][col = l&15].
//
// Split-K: slabs of K (the slowest tile coordinate of a one-dimensional launch), each writing its own C slab; gemm_reduce_slabs_kernel adds the slabs in slab
// order (bitwise reproducible, no atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace kpop {

constexpr int kGT = 128;   // output tile edge
constexpr int kGK = 16;    // K per chunk (32 measured: slower, 41.8 / 69.0 ms against 36.6 / 59.7 at 524,800 x 1,636)
constexpr int kGS = kGT + 17;  // LDS row stride in doubles (odd: transposed staging writes spread over banks)

using f64x4 = __attribute__((ext_vector_type(4))) double;

template <bool TRANS_A>
__global__ __launch_bounds__(256, 2) void gemm_f64_mfma_kernel(const double *__restrict__ A, uint64_t lda,
                                                            const double *__restrict__ B, uint64_t ldb,
                                                            double *__restrict__ C, uint64_t ldc, uint32_t M, uint32_t N,
                                                            uint64_t K, uint64_t k_per_split, int upper_only,
                                                            uint32_t tiles_m, uint32_t tiles_n) {
  __shared__ double As[kGK][kGS];
  __shared__ double Bs[kGK][kGS];
  // Tile order (the launch is one-dimensional).  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.
  //   S W (!TRANS_A): XCD-aware.  The blocks that share id % 8 get a contiguous run of tile numbers (bijective remap,
  //       cdna_hip_programming.md T1) decoded with the N-tiles of one M-tile as neighbours, so the M-panel of S -- the
  //       big operand -- is fetched into one L2 once and reused by all its N-tiles: -5..-9 % measured.
  //   S'S (TRANS_A): plain order (M-tile fastest, K-slab slowest), K-slabs interleaved over the XCDs.  Giving each XCD
  //       whole K-slabs was measured and lost 20 % at 1,636 columns (91 live tiles per slab on 32 CUs: a 2.8-round
  //       tail per slab instead of one for the launch).
  uint32_t wgid = blockIdx.x;
  if (!TRANS_A) {
    const uint32_t nwg = gridDim.x, q8 = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8;
    wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + blockIdx.x / 8;
  }
  const uint32_t bz = wgid / (tiles_m * tiles_n), rem = wgid % (tiles_m * tiles_n);
  const uint32_t bx = TRANS_A ? rem % tiles_m : rem / tiles_n;
  const uint32_t by = TRANS_A ? rem / tiles_m : rem % tiles_n;
  const uint32_t m0 = bx * kGT, n0 = by * kGT;
  if (upper_only && bx > by) return;  // symmetric product: the mirror tile is filled by the reducer
  const uint64_t k_begin = (uint64_t)bz * k_per_split;
  const uint64_t k_end = min(K, k_begin + k_per_split);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
  f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  constexpr int kQ = kGK / 2;  // loads per thread, panel and chunk
  double ra[kQ], rb[kQ];
  // S W (!TRANS_A): the loads are unconditional, from addresses clamped into the operands, and what lies outside the tile is
  // zeroed afterwards -- predicated loads each waited for themselves: 65.0 -> 59.7 ms.  S'S keeps its predicated loads: the
  // same change cost it 3 ms (36.6 -> 40.0).
  auto prefetch = [&](uint64_t k0) {
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      const uint32_t nn = threadIdx.x & 127, kb = (threadIdx.x >> 7) + 2 * q;
      if (TRANS_A) {  // A[k][m]: lanes along m
        const uint32_t mm = threadIdx.x & 127, kk = (threadIdx.x >> 7) + 2 * q;
        ra[q] = (k0 + kk < k_end && m0 + mm < M) ? A[(k0 + kk) * lda + m0 + mm] : 0.0;
        rb[q] = (k0 + kb < k_end && n0 + nn < N) ? B[(k0 + kb) * ldb + n0 + nn] : 0.0;
      } else {  // A[m][k]: lanes along k (kGK consecutive doubles: whole 128-byte lines)
        const uint32_t kk = threadIdx.x & (kGK - 1), mm = threadIdx.x / kGK + (256 / kGK) * q;
        ra[q] = A[(uint64_t)min(m0 + mm, M - 1) * lda + min(k0 + kk, k_end - 1)];
        rb[q] = B[min(k0 + kb, k_end - 1) * ldb + min(n0 + nn, N - 1)];
      }
    }
  };
  if (k_begin < k_end) prefetch(k_begin);
  for (uint64_t k0 = k_begin; k0 < k_end; k0 += kGK) {
    __syncthreads();
    // (S W: what lies outside the operands is zeroed HERE, when the chunk is stored -- not right after its loads, which is before the
    // MFMAs of the chunk in front of it in program order and made them wait for these loads: distance_mfma.hip, profiles/r06_gemm_loads.txt)
    if (!TRANS_A && (k0 + kGK > k_end || m0 + kGT > M || n0 + kGT > N)) {  // (uniform) an edge
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        const uint32_t kk = threadIdx.x & (kGK - 1), mm = threadIdx.x / kGK + (256 / kGK) * q;
        const uint32_t nn = threadIdx.x & 127, kb = (threadIdx.x >> 7) + 2 * q;
        ra[q] = (k0 + kk < k_end && m0 + mm < M) ? ra[q] : 0.0;
        rb[q] = (k0 + kb < k_end && n0 + nn < N) ? rb[q] : 0.0;
      }
    }
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      if (TRANS_A) As[(threadIdx.x >> 7) + 2 * q][threadIdx.x & 127] = ra[q];
      else As[threadIdx.x & (kGK - 1)][threadIdx.x / kGK + (256 / kGK) * q] = ra[q];
      Bs[(threadIdx.x >> 7) + 2 * q][threadIdx.x & 127] = rb[q];
    }
    __syncthreads();
    if (k0 + kGK < k_end) prefetch(k0 + kGK);
#pragma unroll
    for (int ks = 0; ks < kGK; ks += 4) {
      double a[4], b[4];
      const int kr = ks + (lane >> 4), c = lane & 15;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = As[kr][wm + t * 16 + c];
        b[t] = Bs[kr][wn + t * 16 + c];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  double *Cs = C + (uint64_t)bz * M * ldc;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t row = m0 + wm + i * 16 + (lane >> 4) + 4 * r, col = n0 + wn + j * 16 + (lane & 15);
        if (row < M && col < N) Cs[(uint64_t)row * ldc + col] = acc[i][j][r];
      }
}

// out = sum over slabs, in slab order; symmetric: element (i,j) of the lower triangle takes (j,i)
template <int kDummy = 0>
__global__ void gemm_reduce_slabs_kernel(const double *__restrict__ slabs, uint32_t n_slabs, uint32_t M, uint32_t N,
                                         int symmetric, double *__restrict__ out) {
  const uint64_t total = (uint64_t)M * N, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    uint32_t i = (uint32_t)(e / N), j = (uint32_t)(e % N);
    if (symmetric && (i / kGT) > (j / kGT)) {  // tile below the diagonal was skipped
      const uint32_t t = i;
      i = j;
      j = t;
    }
    double s = 0.0;
    for (uint32_t z = 0; z < n_slabs; ++z) s += slabs[(uint64_t)z * total + (uint64_t)i * N + j];
    out[e] = s;
  }
}

// C = op(A) * B.  `slabs` needs n_splits * M * N doubles when n_splits > 1 (then C receives the ordered sum).
template <bool TRANS_A>
static int gemm_f64(const double *A, uint64_t lda, const double *B, uint64_t ldb, double *C, uint32_t M, uint32_t N, uint64_t K,
                    uint32_t n_splits, double *slabs, int symmetric, hipStream_t st) {
  if (M == 0 || N == 0) return 0;
  if (n_splits < 1) n_splits = 1;
  uint64_t kps = (K + n_splits - 1) / n_splits;
  kps = (kps + kGK - 1) / kGK * kGK;
  if (kps == 0) kps = kGK;
  n_splits = (uint32_t)std::max<uint64_t>(1, (K + kps - 1) / kps);
  const uint32_t tiles_m = div_up(M, kGT), tiles_n = div_up(N, kGT);
  const uint64_t n_blocks = (uint64_t)tiles_m * tiles_n * n_splits;
  if (n_blocks >= (1ull << 31)) KPOP_FAIL(KPOP_ERR_UNSUPPORTED, "gemm_f64: %u x %u x %u tiles", tiles_m, tiles_n, n_splits);
  double *dst = (n_splits > 1 || symmetric) ? slabs : C;
  gemm_f64_mfma_kernel<TRANS_A><<<dim3((uint32_t)n_blocks), dim3(256), 0, st>>>(A, lda, B, ldb, dst, N, M, N, K, kps, symmetric,
                                                                                  tiles_m, tiles_n);
  KPOP_LAUNCH_CHECK();
  if (dst != C) {
    gemm_reduce_slabs_kernel<0><<<dim3(std::min<uint32_t>(div_up((uint64_t)M * N, 256), 4096)), dim3(256), 0, st>>>(
        slabs, n_splits, M, N, symmetric, C);
    KPOP_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace kpop

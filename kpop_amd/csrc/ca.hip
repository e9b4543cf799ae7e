// ca.hip -- correspondence analysis: the twister generator (replaces the R stage of src/KPopTwist:93-116).
//
//   stuff  <- counts, each column (spectrum) divided by its sum                 src/KPopTwist:93-94
//   ca()   :  P = N / sum(N); masses r = P 1, c = P' 1;
//             S = D_r^-1/2 (P - r c') D_c^-1/2 = U diag(sv) V'                 library(ca)
//   twisted = D_c^-1/2 V diag(sv)   (principal column coordinates)               :98-100
//   inertia = sv^2 / sum(sv^2)                                                   :105
//   twister = (D_r^-1/2 U)'         (standard row coordinates, dims x k-mers)    :110-116
//
// I = number of k-mers (10^3 .. 10^7), J = number of spectra/classes (10 .. ~2000): tall and skinny.  The SVD is
// taken through the J x J Gram matrix G = S'S (MFMA f64 GEMM, split over K = I), a one-sided Jacobi
// eigen-decomposition of G (J^3, small; one launch per round-robin step), and U = S V diag(1/sv) as a second MFMA GEMM.
// The Gram route squares the condition number: dimensions whose singular value is below ~1e-7 of the largest
// lose accuracy -- they carry < 1e-14 of the inertia.
#include <math.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "gemm_f64.h"

namespace kpop {

// column sums of an I x J row-major matrix: blocks take row slabs, partial[J] per block, ordered final sum
constexpr uint32_t kCaSlab = 2048;

__global__ __launch_bounds__(256) void ca_col_partial_kernel(const double *__restrict__ N, uint64_t I, uint32_t J,
                                                             const double *__restrict__ col_scale,
                                                             double *__restrict__ partial) {
  const uint64_t i0 = (uint64_t)blockIdx.x * kCaSlab, i1 = min(I, i0 + kCaSlab);
  for (uint32_t j = threadIdx.x; j < J; j += 256) {
    const double w = col_scale ? col_scale[j] : 1.0;
    double s = 0.0;
    for (uint64_t i = i0; i < i1; ++i) s += N[i * J + j] * w;
    partial[(uint64_t)blockIdx.x * J + j] = s;
  }
}

__global__ void ca_col_final_kernel(const double *__restrict__ partial, uint32_t n_blocks, uint32_t J,
                                    double *__restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J) return;
  double s = 0.0;
  for (uint32_t b = 0; b < n_blocks; ++b) s += partial[(uint64_t)b * J + j];
  out[j] = s;
}

// r_i = sum_j N_ij w_j ; one wave per row, fixed-order lane tree
__global__ __launch_bounds__(256) void ca_row_mass_kernel(const double *__restrict__ N, uint64_t I, uint32_t J,
                                                          const double *__restrict__ w, double *__restrict__ r) {
  const int lane = threadIdx.x & 63;
  const uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= I) return;
  double s = 0.0;
  for (uint32_t j = lane; j < J; j += 64) s += N[i * J + j] * w[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) r[i] = s;
}

// S_ij = (N_ij w_j - r_i c_j) / sqrt(r_i c_j); rows without mass are zero
__global__ void ca_standardise_kernel(const double *__restrict__ N, uint64_t I, uint32_t J, const double *__restrict__ w,
                                      const double *__restrict__ r, const double *__restrict__ c, double *__restrict__ S) {
  const uint64_t total = I * J, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const uint64_t i = e / J;
    const uint32_t j = (uint32_t)(e % J);
    const double ri = r[i], cj = c[j];
    S[e] = (ri > 0.0 && cj > 0.0) ? (N[e] * w[j] - ri * cj) / sqrt(ri * cj) : 0.0;
  }
}

// twister[d][i] = U_id / sqrt(r_i), written dims-major from the I x nd product
__global__ __launch_bounds__(256) void ca_row_coords_kernel(const double *__restrict__ U, uint64_t I, uint32_t nd,
                                                            const double *__restrict__ r, double *__restrict__ twister) {
  __shared__ double tile[32][33];
  const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const uint64_t i_base = (uint64_t)blockIdx.x * 32;
  const uint32_t d_base = blockIdx.y * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint64_t i = i_base + ty + 8 * q;
    const uint32_t d = d_base + tx;
    double v = 0.0;
    if (i < I && d < nd) {
      const double ri = r[i];
      v = ri > 0.0 ? U[i * nd + d] / sqrt(ri) : 0.0;
    }
    tile[ty + 8 * q][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t d = d_base + ty + 8 * q;
    const uint64_t i = i_base + tx;
    if (d < nd && i < I) twister[(uint64_t)d * I + i] = tile[tx][ty + 8 * q];
  }
}

// One-sided Jacobi (Hestenes) on the columns of the symmetric PSD matrix G (n x n): on exit the column norms
// are the eigenvalues and V holds the eigenvectors.  Round-robin (circle method) pairing: the n/2 rotations of a
// step touch disjoint columns, so a step is one launch with one 256-thread block per pair; 3 ordered block
// reductions (|a_p|^2, |a_q|^2, a_p.a_q) then the rotation of both columns of A and of V.  The largest
// |a_p.a_q| / (|a_p||a_q|) of a sweep is kept in `worst` (non-negative doubles order like their bit patterns).
__device__ __forceinline__ double block_sum_256(double v, double *s_w) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}

__global__ __launch_bounds__(256) void jacobi_step_kernel(double *__restrict__ A, double *__restrict__ V, uint32_t n,
                                                          uint32_t m, uint32_t step, unsigned long long *worst) {
  __shared__ double s_w[4];
  const uint32_t t = blockIdx.x;
  uint32_t p, q;
  if (t == 0) {
    p = m - 1;
    q = step;
  } else {
    p = (step + t) % (m - 1);
    q = (step + (m - 1) - t) % (m - 1);
  }
  if (p >= n || q >= n) return;  // the bye of an odd n
  if (p > q) {
    const uint32_t x = p;
    p = q;
    q = x;
  }
  double *ap = A + (uint64_t)p * n, *aq = A + (uint64_t)q * n;
  double alpha = 0.0, beta = 0.0, gamma = 0.0;
  for (uint32_t i = threadIdx.x; i < n; i += 256) {
    const double x = ap[i], y = aq[i];
    alpha += x * x;
    beta += y * y;
    gamma += x * y;
  }
  alpha = block_sum_256(alpha, s_w);
  beta = block_sum_256(beta, s_w);
  gamma = block_sum_256(gamma, s_w);
  if (gamma == 0.0) return;
  const double denom = sqrt(alpha * beta);
  const double off = denom > 0.0 ? fabs(gamma) / denom : 0.0;
  if (threadIdx.x == 0) atomicMax(worst, (unsigned long long)__double_as_longlong(off));
  if (off < 1e-15) return;
  const double zeta = (beta - alpha) / (2.0 * gamma);
  const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
  double *vp = V + (uint64_t)p * n, *vq = V + (uint64_t)q * n;
  for (uint32_t i = threadIdx.x; i < n; i += 256) {
    const double x = ap[i], y = aq[i];
    ap[i] = cs * x - sn * y;
    aq[i] = sn * x + cs * y;
    const double vx = vp[i], vy = vq[i];
    vp[i] = cs * vx - sn * vy;
    vq[i] = sn * vx + cs * vy;
  }
}

// ---------------------------------------------------------------------------
// Blocked one-sided Jacobi.  The plain step above rotates ONE column pair per workgroup and moves all of A and V through
// the caches for it: n - 1 such steps per sweep (22,890 launches of 22 us at n = 1,636).  Here a workgroup takes two
// blocks of kJB = 4 columns, forms their 8 x 8 Gram matrix in one pass, diagonalises it in registers -- one wavefront, a
// lane per matrix element, the four disjoint rotations of a round-robin round at a time -- and applies the accumulated
// 8 x 8 rotation to the eight columns of A and of V in a second pass: 28 column pairs orthogonalised per trip through
// memory instead of one, and n / 4 - 1 steps per sweep.
// ---------------------------------------------------------------------------
constexpr int kJB = 4, kJC = 2 * kJB;  // columns per block, per workgroup

__device__ __forceinline__ int jb_partner(int i, int round) {  // circle method on 8 indices, index 7 fixed
  if (i == 7) return round;
  if (i == round) return 7;
  return (2 * round - i + 14) % 7;
}

template <int NT>
__global__ __launch_bounds__(NT) void jacobi_block_step_kernel(double *__restrict__ A, double *__restrict__ V, uint32_t n,
                                                                uint32_t m_blk, uint32_t step, unsigned long long *worst, int inner_sweeps) {
  __shared__ double s_part[NT / 64][kJC * kJC];
  __shared__ double s_R[kJC][kJC];
  __shared__ int s_skip;
  const uint32_t t = blockIdx.x;
  uint32_t P, Q;
  if (t == 0) {
    P = m_blk - 1;
    Q = step;
  } else {
    P = (step + t) % (m_blk - 1);
    Q = (step + (m_blk - 1) - t) % (m_blk - 1);
  }
  if (P > Q) {
    const uint32_t x = P;
    P = Q;
    Q = x;
  }
  uint32_t col[kJC];
  bool ok[kJC];
#pragma unroll
  for (int c = 0; c < kJC; ++c) {
    col[c] = (c < kJB ? P * kJB + c : Q * kJB + (c - kJB));
    ok[c] = col[c] < n;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // 8 x 8 Gram matrix of the columns (upper triangle, 36 sums per thread)
  double g[kJC * (kJC + 1) / 2];
#pragma unroll
  for (int e = 0; e < kJC * (kJC + 1) / 2; ++e) g[e] = 0.0;
  for (uint32_t i = threadIdx.x; i < n; i += NT) {
    double x[kJC];
#pragma unroll
    for (int c = 0; c < kJC; ++c) x[c] = ok[c] ? A[(uint64_t)col[c] * n + i] : 0.0;
    int e = 0;
#pragma unroll
    for (int a = 0; a < kJC; ++a)
#pragma unroll
      for (int b = a; b < kJC; ++b) g[e++] += x[a] * x[b];
  }
  {
    int e = 0;
#pragma unroll
    for (int a = 0; a < kJC; ++a)
#pragma unroll
      for (int b = a; b < kJC; ++b) {
        double v = g[e++];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) {
          s_part[wv][a * kJC + b] = v;
          s_part[wv][b * kJC + a] = v;
        }
      }
  }
  __syncthreads();
  if (wv == 0) {
    const int r = lane >> 3, c = lane & 7;
    double M = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) M += s_part[w][lane];
    double R = r == c ? 1.0 : 0.0;
    // how far from orthogonal the eight columns are (the sweep's convergence measure, as in the plain step)
    const double drr = __shfl(M, 9 * r, 64), dcc = __shfl(M, 9 * c, 64);
    double off = (r < c && drr > 0.0 && dcc > 0.0) ? fabs(M) / sqrt(drr * dcc) : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) off = fmax(off, __shfl_xor(off, o, 64));
    if (lane == 0) {
      atomicMax(worst, (unsigned long long)__double_as_longlong(off));
      s_skip = off < 1e-15;
    }
    if (!(off < 1e-15)) {
      for (int sweep = 0; sweep < inner_sweeps; ++sweep) {
        for (int round = 0; round < 7; ++round) {
          const int pr = jb_partner(r, round), pc = jb_partner(c, round);
          // rotation of the plane (a, b), a < b, that holds index i: from M[a][a], M[b][b], M[a][b]
          double cs[2], sn[2];
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const int i = w ? c : r, pi = w ? pc : pr;
            const int a = min(i, pi), b = max(i, pi);
            const double app = __shfl(M, 9 * a, 64), aqq = __shfl(M, 9 * b, 64), apq = __shfl(M, 8 * a + b, 64);
            double cc = 1.0, ss = 0.0;
            if (apq != 0.0) {
              const double zeta = (aqq - app) / (2.0 * apq);
              const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
              cc = 1.0 / sqrt(1.0 + tt * tt);
              ss = cc * tt;
            }
            cs[w] = cc;
            sn[w] = ss;
          }
          // J[a][a] = J[b][b] = cs, J[a][b] = sn, J[b][a] = -sn;  M <- J' M J,  R <- R J
          const double jr_other = (r < pr) ? -sn[0] : sn[0];  // J[pr][r]
          const double jc_other = (c < pc) ? -sn[1] : sn[1];  // J[pc][c]
          const double m_rpc = __shfl(M, 8 * r + pc, 64), m_prc = __shfl(M, 8 * pr + c, 64), m_prpc = __shfl(M, 8 * pr + pc, 64);
          const double r_rpc = __shfl(R, 8 * r + pc, 64);
          M = cs[0] * (M * cs[1] + m_rpc * jc_other) + jr_other * (m_prc * cs[1] + m_prpc * jc_other);
          R = R * cs[1] + r_rpc * jc_other;
        }
        double od = (r != c) ? fabs(M) : 0.0, dg = (r == c) ? fabs(M) : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          od = fmax(od, __shfl_xor(od, o, 64));
          dg = fmax(dg, __shfl_xor(dg, o, 64));
        }
        if (od <= 1e-16 * dg) break;
      }
    }
    s_R[r][c] = R;
  }
  __syncthreads();
  if (s_skip) return;
  double Rl[kJC][kJC];
#pragma unroll
  for (int a = 0; a < kJC; ++a)
#pragma unroll
    for (int b = 0; b < kJC; ++b) Rl[a][b] = s_R[a][b];
  for (uint32_t i = threadIdx.x; i < n; i += NT) {
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      double *X = which ? V : A;
      double x[kJC], y[kJC];
#pragma unroll
      for (int c = 0; c < kJC; ++c) x[c] = ok[c] ? X[(uint64_t)col[c] * n + i] : 0.0;
#pragma unroll
      for (int c = 0; c < kJC; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < kJC; ++a) acc += x[a] * Rl[a][c];
        y[c] = acc;
      }
#pragma unroll
      for (int c = 0; c < kJC; ++c)
        if (ok[c]) X[(uint64_t)col[c] * n + i] = y[c];
    }
  }
}

__global__ void jacobi_identity_kernel(double *__restrict__ V, uint32_t n) {
  const uint64_t total = (uint64_t)n * n, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) V[e] = (e / n == e % n) ? 1.0 : 0.0;
}

__global__ __launch_bounds__(256) void jacobi_colnorm_kernel(const double *__restrict__ A, uint32_t n, double *__restrict__ lambda) {
  __shared__ double s_w[4];
  const double *a = A + (uint64_t)blockIdx.x * n;
  double s = 0.0;
  for (uint32_t i = threadIdx.x; i < n; i += 256) s += a[i] * a[i];
  s = block_sum_256(s, s_w);
  if (threadIdx.x == 0) lambda[blockIdx.x] = sqrt(s);
}

// d_G is overwritten (columns rotated); d_V receives the eigenvectors as columns (V[j*n + i] = component i of vector j)
static int jacobi_eigen_psd_device(double *d_G, double *d_V, uint32_t n, double *d_lambda, hipStream_t st) {
  DevBuf worst;
  KPOP_TRY(worst.alloc(8));
  jacobi_identity_kernel<<<dim3(std::min<uint32_t>(div_up((uint64_t)n * n, 256), 4096)), dim3(256), 0, st>>>(d_V, n);
  KPOP_LAUNCH_CHECK();
  const uint32_t m = (n + 1) & ~1u;
  const uint32_t n_blk = div_up(n, kJB), m_blk = (n_blk + 1) & ~1u;
  const bool blocked = n >= 4 * kJC && !(ctx().tune_dbg & 32);  // (32: the plain steps, for A/B)
  for (int sweep = 0; sweep < 60; ++sweep) {
    KPOP_HIP(hipMemsetAsync(worst.p, 0, 8, st));
    if (blocked) {
      for (uint32_t step = 0; step + 1 < m_blk; ++step) {
        jacobi_block_step_kernel<256><<<dim3(m_blk / 2), dim3(256), 0, st>>>(d_G, d_V, n, m_blk, step, worst.as<unsigned long long>(), (ctx().tune_dbg & 15) ? (ctx().tune_dbg & 15) : 1);
        KPOP_LAUNCH_CHECK();
      }
    } else
    for (uint32_t step = 0; step + 1 < m; ++step) {
      jacobi_step_kernel<<<dim3(m / 2), dim3(256), 0, st>>>(d_G, d_V, n, m, step, worst.as<unsigned long long>());
      KPOP_LAUNCH_CHECK();
    }
    double w = 0.0;
    KPOP_HIP(hipMemcpyAsync(&w, worst.p, 8, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipStreamSynchronize(st));
    if (w < 1e-15) break;
  }
  jacobi_colnorm_kernel<<<dim3(n), dim3(256), 0, st>>>(d_G, n, d_lambda);
  KPOP_LAUNCH_CHECK();
  return 0;
}

}  // namespace kpop

using namespace kpop;

extern "C" int kpop_ca(const double *counts, uint64_t n_kmers, uint32_t n_spectra, int normalize, uint32_t *n_dims_out,
                       double *twisted, double *inertia, double *twister) {
  KPOP_TRY(require_init());
  const uint64_t I = n_kmers;
  const uint32_t J = n_spectra;
  if (!counts || !n_dims_out || !twisted || !inertia || !twister) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_ca: null argument");
  if (I < 2 || J < 2) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_ca: need at least 2 k-mers and 2 spectra");
  const uint32_t nd = (uint32_t)std::min<uint64_t>(I, J) - 1;
  *n_dims_out = nd;
  hipStream_t st = nullptr;
  const uint32_t n_slabs = div_up(I, kCaSlab);
  DevBuf dN, dS, dW, dR, dC, dPart, dG, dGslabs, dWm, dU, dT;
  KPOP_TRY(dN.alloc(I * J * 8));
  KPOP_TRY(dS.alloc(I * J * 8));
  KPOP_TRY(dW.alloc((uint64_t)J * 8));
  KPOP_TRY(dR.alloc(I * 8));
  KPOP_TRY(dC.alloc((uint64_t)J * 8));
  KPOP_TRY(dPart.alloc((uint64_t)n_slabs * J * 8));
  KPOP_HIP(hipMemcpyAsync(dN.p, counts, I * J * 8, hipMemcpyHostToDevice, st));
  // column sums -> weights w_j (P_ij = N_ij w_j)
  ca_col_partial_kernel<<<dim3(n_slabs), dim3(256), 0, st>>>(dN.as<double>(), I, J, nullptr, dPart.as<double>());
  KPOP_LAUNCH_CHECK();
  ca_col_final_kernel<<<dim3(div_up(J, 256)), dim3(256), 0, st>>>(dPart.as<double>(), n_slabs, J, dC.as<double>());
  KPOP_LAUNCH_CHECK();
  std::vector<double> colsum(J), w(J), c(J);
  KPOP_HIP(hipMemcpyAsync(colsum.data(), dC.p, (uint64_t)J * 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  double total = 0.0;
  for (uint32_t j = 0; j < J; ++j) {
    if (!(colsum[j] > 0.0)) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_ca: spectrum %u has no counts", j);
    total += normalize ? 1.0 : colsum[j];
  }
  for (uint32_t j = 0; j < J; ++j) {
    w[j] = normalize ? 1.0 / colsum[j] / total : 1.0 / total;  // col/sum(col), then /sum(N)
    c[j] = colsum[j] * w[j];
  }
  KPOP_HIP(hipMemcpyAsync(dW.p, w.data(), (uint64_t)J * 8, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(dC.p, c.data(), (uint64_t)J * 8, hipMemcpyHostToDevice, st));
  ca_row_mass_kernel<<<dim3(div_up(I, 4)), dim3(256), 0, st>>>(dN.as<double>(), I, J, dW.as<double>(), dR.as<double>());
  KPOP_LAUNCH_CHECK();
  ca_standardise_kernel<<<dim3(std::min<uint32_t>(div_up(I * J, 256), 1u << 20)), dim3(256), 0, st>>>(
      dN.as<double>(), I, J, dW.as<double>(), dR.as<double>(), dC.as<double>(), dS.as<double>());
  KPOP_LAUNCH_CHECK();
  // G = S'S  (J x J), K = I split over up to 64 slabs
  const uint32_t tiles = div_up(J, kGT) * div_up(J, kGT);
  uint32_t splits = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, 2048 / std::max(1u, tiles)), std::max<uint64_t>(1, I / 4096));
  splits = std::min(splits, 256u);
  KPOP_TRY(dG.alloc((uint64_t)J * J * 8));
  KPOP_TRY(dGslabs.alloc((uint64_t)splits * J * J * 8));
  KPOP_TRY(gemm_f64<true>(dS.as<double>(), J, dS.as<double>(), J, dG.as<double>(), J, J, I, splits, dGslabs.as<double>(), 1, st));
  DevBuf dV, dLambda;
  KPOP_TRY(dV.alloc((uint64_t)J * J * 8));
  KPOP_TRY(dLambda.alloc((uint64_t)J * 8));
  KPOP_TRY(jacobi_eigen_psd_device(dG.as<double>(), dV.as<double>(), J, dLambda.as<double>(), st));
  std::vector<double> V((size_t)J * J), lambda(J);
  KPOP_HIP(hipMemcpyAsync(V.data(), dV.p, (uint64_t)J * J * 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipMemcpyAsync(lambda.data(), dLambda.p, (uint64_t)J * 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  // order by decreasing eigenvalue; sv = sqrt(lambda)
  std::vector<uint32_t> order(J);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return lambda[a] > lambda[b]; });
  std::vector<double> sv(nd), Wm((size_t)J * nd);
  double sum_sq = 0.0;
  for (uint32_t d = 0; d < nd; ++d) {
    sv[d] = sqrt(std::max(lambda[order[d]], 0.0));
    sum_sq += sv[d] * sv[d];
  }
  for (uint32_t d = 0; d < nd; ++d) {
    inertia[d] = sum_sq > 0.0 ? sv[d] * sv[d] / sum_sq : 0.0;
    const double *v = &V[(size_t)order[d] * J];
    for (uint32_t j = 0; j < J; ++j) {
      twisted[(size_t)j * nd + d] = v[j] / sqrt(c[j]) * sv[d];   // D_c^-1/2 V diag(sv)
      Wm[(size_t)j * nd + d] = sv[d] > 0.0 ? v[j] / sv[d] : 0.0;  // V diag(1/sv)
    }
  }
  // U = S W (I x nd), then the twister = (D_r^-1/2 U)'
  KPOP_TRY(dWm.alloc((uint64_t)J * nd * 8));
  KPOP_TRY(dU.alloc(I * nd * 8));
  KPOP_TRY(dT.alloc(I * nd * 8));
  KPOP_HIP(hipMemcpyAsync(dWm.p, Wm.data(), (uint64_t)J * nd * 8, hipMemcpyHostToDevice, st));
  KPOP_TRY(gemm_f64<false>(dS.as<double>(), J, dWm.as<double>(), nd, dU.as<double>(), (uint32_t)std::min<uint64_t>(I, 0xFFFFFFFFull), nd, J,
                           1, nullptr, 0, st));
  ca_row_coords_kernel<<<dim3(div_up(I, 32), div_up(nd, 32)), dim3(256), 0, st>>>(dU.as<double>(), I, nd, dR.as<double>(),
                                                                                 dT.as<double>());
  KPOP_LAUNCH_CHECK();
  KPOP_HIP(hipMemcpyAsync(twister, dT.p, I * nd * 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  return KPOP_OK;
}

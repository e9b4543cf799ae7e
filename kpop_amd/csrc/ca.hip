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

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
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
    uint64_t i = i0;
    for (; i + 8 <= i1; i += 8) {  // eight loads in flight, added in row order (one at a time ran at 1.2 TB/s)
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = N[(i + u) * J + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u] * w;
    }
    for (; i < i1; ++i) s += N[i * J + j] * w;
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
  if (w) {
    for (uint32_t j = lane; j < J; j += 64) s += N[i * J + j] * w[j];
  } else {  // plain row sums
    for (uint32_t j = lane; j < J; j += 64) s += N[i * J + j];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) r[i] = s;
}

// S_ij = (N_ij w_j - r_i c_j) / sqrt(r_i c_j); rows without mass are zero
__global__ void ca_standardise_kernel(const double *N, uint64_t I, uint32_t J, const double *__restrict__ w,
                                      const double *__restrict__ r, const double *__restrict__ c, double *S) {  // (S may be N)
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

// One wavefront, a lane per element of the 8 x 8 Gram matrix summed from the waves' partials: the sweep's convergence
// measure goes to *worst, the accumulated rotation to s_R (the identity, and *s_skip set, when the eight columns are
// orthogonal already).
template <int WAVES>
__device__ __forceinline__ void jacobi_solve_8x8(double (*s_part)[kJC * kJC], double (*s_R)[kJC], int *s_skip, unsigned long long *worst,
                                                 int inner_sweeps) {
  const int lane = threadIdx.x & 63;
  const int r = lane >> 3, c = lane & 7;
  double M = 0.0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) M += s_part[w][lane];
  double R = r == c ? 1.0 : 0.0;
  // how far from orthogonal the eight columns are (the sweep's convergence measure, as in the plain step)
  const double drr = __shfl(M, 9 * r, 64), dcc = __shfl(M, 9 * c, 64);
  double off = (r < c && drr > 0.0 && dcc > 0.0) ? fabs(M) / sqrt(drr * dcc) : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) off = fmax(off, __shfl_xor(off, o, 64));
  if (lane == 0) {
    atomicMax(worst, (unsigned long long)__double_as_longlong(off));
    *s_skip = off < 1e-15;
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
  if (wv == 0) jacobi_solve_8x8<NT / 64>(s_part, s_R, &s_skip, worst, inner_sweeps);
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

// v[0 .. N) summed over the 64 lanes, spread over them: after the stage at distance OFF the lanes with that bit clear
// hold (the pair sums of) the lower half of the values, the others the upper half.
template <int N, int OFF>
__device__ __forceinline__ void jacobi_halve(double *v, int lane) {
  constexpr int H = (N + 1) / 2;
  const bool upper = lane & OFF;
#pragma unroll
  for (int j = 0; j < H; ++j) {
    const double lo = v[j], hi = (j + H < N) ? v[j + H] : 0.0;
    const double keep = upper ? hi : lo, send = upper ? lo : hi;
    v[j] = keep + __shfl_xor(send, OFF, 64);
  }
}
__device__ __forceinline__ void jacobi_wave_totals(double *v, int lane) {  // 36 -> 18 -> 9 -> 5 -> 3 -> 2 -> 1
  jacobi_halve<36, 32>(v, lane);
  jacobi_halve<18, 16>(v, lane);
  jacobi_halve<9, 8>(v, lane);
  jacobi_halve<5, 4>(v, lane);
  jacobi_halve<3, 2>(v, lane);
  jacobi_halve<2, 1>(v, lane);
}
static_assert(kJC * (kJC + 1) / 2 == 36, "jacobi_wave_totals is written for 36 sums");

// The same step with the workgroup's rows of its eight columns held in registers (n <= 256 * ROWS): every load of the Gram
// pass is in flight at once instead of one row of eight per round trip, the eight columns of V are requested before the
// 8 x 8 problem is solved (their latency hides behind it), and A is rotated from registers without a second read.  The
// looped kernel above spends 17 us of its 50 us per step in the Gram pass alone at n = 1,636: one wavefront per SIMD and
// seven dependent trips to the L2 / MALL.
template <int ROWS, bool WITH_V>
__global__ __launch_bounds__(256) void jacobi_block_step_reg_kernel(double *__restrict__ A, double *__restrict__ V, uint32_t n,
                                                                    uint32_t m_blk, uint32_t step, unsigned long long *worst, int inner_sweeps) {
  __shared__ double s_part[4][kJC * kJC];
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
  uint64_t base[kJC];
  bool ok[kJC];
#pragma unroll
  for (int c = 0; c < kJC; ++c) {
    const uint32_t col = (c < kJB ? P * kJB + c : Q * kJB + (c - kJB));
    ok[c] = col < n;
    base[c] = (uint64_t)(ok[c] ? col : 0u) * n;  // (loads are unconditional -- a predicated load costs its own round trip -- and masked after)
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double x[ROWS][kJC];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const uint32_t i = threadIdx.x + 256u * r;
    const uint32_t ii = i < n ? i : 0u;
#pragma unroll
    for (int c = 0; c < kJC; ++c) x[r][c] = A[base[c] + ii];
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const bool in = threadIdx.x + 256u * r < n;
#pragma unroll
    for (int c = 0; c < kJC; ++c) x[r][c] = (in && ok[c]) ? x[r][c] : 0.0;
  }
  {
    double g[kJC * (kJC + 1) / 2];
#pragma unroll
    for (int e = 0; e < kJC * (kJC + 1) / 2; ++e) g[e] = 0.0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      int e = 0;
#pragma unroll
      for (int a = 0; a < kJC; ++a)
#pragma unroll
        for (int b = a; b < kJC; ++b) g[e++] += x[r][a] * x[r][b];
    }
    // sum over the wavefront: a halving exchange (the lane pair at distance OFF splits the values between them), 38
    // exchanges instead of 36 x 6, none waiting on the one before; lane l ends with the total of entry `mine`
    jacobi_wave_totals(g, lane);
    int mine = 0, real = kJC * (kJC + 1) / 2, slots = real;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int h = (slots + 1) / 2;
      if (lane & off) {
        mine += h;
        real = max(real - h, 0);
      } else
        real = min(real, h);
      slots = h;
    }
    if (real > 0) {  // (36 entries: the uneven halvings leave 28 lanes holding padding)
      int a = 0, rest = mine;
      while (rest >= kJC - a) {
        rest -= kJC - a;
        ++a;
      }
      const int b = a + rest;
      s_part[wv][a * kJC + b] = g[0];
      s_part[wv][b * kJC + a] = g[0];
    }
  }
  double y[WITH_V ? ROWS : 1][kJC];  // the rows of V, asked for now and used after the 8 x 8 problem
  if (WITH_V) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const uint32_t i = threadIdx.x + 256u * r;
      const uint32_t ii = i < n ? i : 0u;
#pragma unroll
      for (int c = 0; c < kJC; ++c) y[WITH_V ? r : 0][c] = V[base[c] + ii];
    }
  }
  __syncthreads();
  if (wv == 0) jacobi_solve_8x8<4>(s_part, s_R, &s_skip, worst, inner_sweeps);
  __syncthreads();
  if (s_skip) return;
  double Rl[kJC][kJC];
#pragma unroll
  for (int a = 0; a < kJC; ++a)
#pragma unroll
    for (int b = 0; b < kJC; ++b) Rl[a][b] = s_R[a][b];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const bool in = threadIdx.x + 256u * r < n;
#pragma unroll
    for (int c = 0; c < kJC; ++c) {
      double acc = 0.0, acv = 0.0;
#pragma unroll
      for (int a = 0; a < kJC; ++a) {
        acc += x[r][a] * Rl[a][c];
        if (WITH_V) acv += y[WITH_V ? r : 0][a] * Rl[a][c];
      }
      if (in && ok[c]) {
        A[base[c] + threadIdx.x + 256u * r] = acc;
        if (WITH_V) V[base[c] + threadIdx.x + 256u * r] = acv;
      }
    }
  }
}

__global__ void jacobi_identity_kernel(double *__restrict__ V, uint32_t n) {
  const uint64_t total = (uint64_t)n * n, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) V[e] = (e / n == e % n) ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------
// Cholesky factor of the Gram matrix, in place (G = L L'; G is symmetric, so its storage is L's column-major one: column j
// at A + j n).  Right-looking, panels of kCholW columns: the diagonal block is factored in LDS by one workgroup, the rows
// below it are solved against it a thread each, and the trailing matrix loses the panel's outer product (the f64 MFMA GEMM,
// then a subtraction).  G is positive SEMI-definite -- the analysis' trivial direction, or fewer k-mers than spectra: a
// pivot not above `tol` closes its column with zeros (in exact arithmetic the whole row and column of the Schur complement
// are zero there).  Why a Cholesky factor at all: the one-sided Jacobi iteration on L needs no accumulated V (the
// eigenvectors of G are L's normalised columns once they are orthogonal) and starts from columns that are far closer to
// orthogonal than G's own.
// ---------------------------------------------------------------------------
constexpr int kCholW = 64;

__global__ __launch_bounds__(256) void chol_diag_kernel(double *__restrict__ A, uint32_t n, uint32_t j0, uint32_t w, double tol,
                                                        double *__restrict__ D, uint32_t *__restrict__ n_dead) {
  __shared__ double s[kCholW][kCholW + 1];  // s[c][r]: row r of column c
  for (uint32_t e = threadIdx.x; e < (uint32_t)kCholW * kCholW; e += 256) {
    const uint32_t c = e / kCholW, r = e % kCholW;
    s[c][r] = (c < w && r < w) ? A[(uint64_t)(j0 + c) * n + j0 + r] : (c == r ? 1.0 : 0.0);  // (beyond w: the identity)
  }
  __syncthreads();
  for (uint32_t k = 0; k < w; ++k) {
    const double d = s[k][k];
    const bool dead = !(d > tol);
    if (dead && threadIdx.x == 0) atomicAdd(n_dead, 1u);
    const double root = dead ? 0.0 : sqrt(d);
    __syncthreads();  // everyone has the pivot
    for (uint32_t r = k + threadIdx.x; r < w; r += 256) s[k][r] = (r == k) ? root : (dead ? 0.0 : s[k][r] / root);
    __syncthreads();
    const uint32_t t = w - k - 1;
    for (uint32_t e = threadIdx.x; e < t * t; e += 256) {
      const uint32_t c = k + 1 + e / t, r = k + 1 + e % t;
      if (r >= c) s[c][r] -= s[k][r] * s[k][c];
    }
    __syncthreads();
  }
  for (uint32_t e = threadIdx.x; e < (uint32_t)kCholW * kCholW; e += 256) {
    const uint32_t c = e / kCholW, r = e % kCholW;
    const double v = r >= c ? s[c][r] : 0.0;
    D[e] = v;
    if (c < w && r < w) A[(uint64_t)(j0 + c) * n + j0 + r] = v;  // (the upper triangle of the block becomes the zeros L has there)
  }
}

// Rows below the diagonal block: x L11' = a, a thread per row, the row in registers.  Right-looking, a column of L11 at a
// time, and the row SHIFTS as it goes: the entry being solved is always x[0], it is stored at once, and the update of the
// entries after it moves them one place down (x[j-1] = x[j] - x0 L11[c+j][c]) -- so every register index is a constant
// while the column index c stays a loop variable.  (Unrolled over c with the row in place, the compiler fetched all 2,016
// entries of L11 ahead of the arithmetic and spilled 15 KB a lane: 463 us per panel.)
__global__ __launch_bounds__(256) void chol_panel_kernel(double *__restrict__ A, uint32_t n, uint32_t j0, uint32_t w, const double *__restrict__ D) {
  __shared__ double L[kCholW][2 * kCholW];  // L[c][r]: row r of column c of L11 (identity beyond w); zeros from row kCholW on
  for (uint32_t e = threadIdx.x; e < (uint32_t)kCholW * 2 * kCholW; e += 256) {
    const uint32_t c = e / (2 * kCholW), r = e % (2 * kCholW);
    L[c][r] = r < (uint32_t)kCholW ? D[c * kCholW + r] : 0.0;
  }
  __syncthreads();
  const uint32_t i = j0 + w + blockIdx.x * 256 + threadIdx.x;
  const uint32_t ii = i < n ? i : n - 1;
  double x[kCholW];
  {
    const double *src = A + (uint64_t)j0 * n + ii;
#pragma unroll
    for (int c = 0; c < kCholW; ++c) {
      x[c] = (uint32_t)c < w ? *src : 0.0;
      if ((uint32_t)c + 1 < w) src += n;
    }
  }
  double *dst = A + (uint64_t)j0 * n + ii;
#pragma unroll 1
  for (uint32_t c = 0; c < w; ++c) {
    const double *col = &L[c][c];
    const double d = col[0];
    const double x0 = d > 0.0 ? x[0] / d : 0.0;
    if (i < n) *dst = x0;
    dst += n;
#pragma unroll
    for (int j = 1; j < kCholW; ++j) x[j - 1] = __builtin_fma(-x0, col[j], x[j]);
    x[kCholW - 1] = 0.0;
  }
}

// the trailing m x m block at (off, off) loses C (m x m, its own leading dimension)
__global__ void chol_subtract_kernel(double *__restrict__ A, uint32_t n, uint32_t off, uint32_t m, const double *__restrict__ C) {
  const uint64_t total = (uint64_t)m * m, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const uint32_t c = (uint32_t)(e / m), i = (uint32_t)(e % m);
    A[(uint64_t)(off + c) * n + off + i] -= C[e];
  }
}

// what is above the diagonal of L (rows i < j of column j) is zero; the panels left the Gram matrix's entries there
__global__ void chol_zero_upper_kernel(double *__restrict__ A, uint32_t n) {
  const uint64_t total = (uint64_t)n * n, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride)
    if (e % n < e / n) A[e] = 0.0;
}

// columns orthogonal: eigenvalue = squared norm, eigenvector = the column over its norm (zero for a null direction)
__global__ __launch_bounds__(256) void jacobi_unit_columns_kernel(const double *__restrict__ A, uint32_t n, double *__restrict__ V,
                                                                  double *__restrict__ lambda) {
  __shared__ double s_w[4];
  const double *a = A + (uint64_t)blockIdx.x * n;
  double s = 0.0;
  for (uint32_t i = threadIdx.x; i < n; i += 256) s += a[i] * a[i];
  s = block_sum_256(s, s_w);
  if (threadIdx.x == 0) lambda[blockIdx.x] = s;
  const double norm = sqrt(s);
  for (uint32_t i = threadIdx.x; i < n; i += 256) V[(uint64_t)blockIdx.x * n + i] = norm > 0.0 ? a[i] / norm : 0.0;
}

// *dead_out: the pivots at or below the tolerance (their columns are zeroed, not pivoted around)
static int cholesky_in_place(double *d_A, uint32_t n, double *d_scratch /* n x n */, hipStream_t st, uint32_t *dead_out) {
  // the tolerance of a pivot: the largest diagonal entry of G times a few thousand roundings
  std::vector<double> diag(n);
  KPOP_HIP(hipMemcpy2DAsync(diag.data(), 8, d_A, (size_t)(n + 1) * 8, 8, n, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  double mx = 0.0;
  for (double v : diag) mx = std::max(mx, v);
  const double tol = mx * 1e-13;
  DevBuf dD;
  KPOP_TRY(dD.alloc((uint64_t)kCholW * kCholW * 8 + 64));
  uint32_t *d_dead = reinterpret_cast<uint32_t *>(dD.as<double>() + (uint64_t)kCholW * kCholW);
  KPOP_HIP(hipMemsetAsync(d_dead, 0, 4, st));
  for (uint32_t j0 = 0; j0 < n; j0 += kCholW) {
    const uint32_t w = std::min<uint32_t>(kCholW, n - j0), m = n - j0 - w;
    chol_diag_kernel<<<dim3(1), dim3(256), 0, st>>>(d_A, n, j0, w, tol, dD.as<double>(), d_dead);
    KPOP_LAUNCH_CHECK();
    if (m == 0) break;
    chol_panel_kernel<<<dim3(div_up(m, 256)), dim3(256), 0, st>>>(d_A, n, j0, w, dD.as<double>());
    KPOP_LAUNCH_CHECK();
    // C = P P' with P the m x w panel: its columns are w contiguous runs of memory, i.e. the w x m row-major matrix P'
    const double *P = d_A + (uint64_t)j0 * n + j0 + w;
    KPOP_TRY(gemm_f64<true>(P, n, P, n, d_scratch, m, m, w, 1, nullptr, 0, st));
    chol_subtract_kernel<<<dim3(std::min<uint32_t>(div_up((uint64_t)m * m, 256), 4096)), dim3(256), 0, st>>>(d_A, n, j0 + w, m, d_scratch);
    KPOP_LAUNCH_CHECK();
  }
  chol_zero_upper_kernel<<<dim3(std::min<uint32_t>(div_up((uint64_t)n * n, 256), 4096)), dim3(256), 0, st>>>(d_A, n);
  KPOP_LAUNCH_CHECK();
  KPOP_HIP(hipMemcpyAsync(dead_out, d_dead, 4, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  return 0;
}

__global__ __launch_bounds__(256) void jacobi_colnorm_kernel(const double *__restrict__ A, uint32_t n, double *__restrict__ lambda) {
  __shared__ double s_w[4];
  const double *a = A + (uint64_t)blockIdx.x * n;
  double s = 0.0;
  for (uint32_t i = threadIdx.x; i < n; i += 256) s += a[i] * a[i];
  s = block_sum_256(s, s_w);
  if (threadIdx.x == 0) lambda[blockIdx.x] = sqrt(s);
}

// d_G is overwritten; d_V receives the eigenvectors as columns (V[j*n + i] = component i of vector j), d_lambda the eigenvalues
static int jacobi_eigen_psd_device(double *d_G, double *d_V, uint32_t n, double *d_lambda, unsigned long long *d_worst, hipStream_t st) {
  const uint32_t m = (n + 1) & ~1u;
  const uint32_t n_blk = div_up(n, kJB), m_blk = (n_blk + 1) & ~1u;
  const bool blocked = n >= 4 * kJC && !(ctx().tune_dbg & 32);  // (32: the plain steps, for A/B)
  // From 32 columns up to what the register kernel holds: Jacobi on the Cholesky factor of G, no V to rotate.  (2048: on G
  // itself with V accumulated, as below that size and above it, for A/B.)
  bool on_factor = blocked && n <= 2048 && !(ctx().tune_dbg & (64 | 2048));
  if (on_factor) {
    // The factorisation does not pivot: a pivot at the rounding floor is declared dead and its column zeroed.  A Gram matrix
    // of standardised residuals has ONE such direction by construction (the trivial axis); MORE of them mean near-dependent
    // columns (near-duplicate classes), where the couplings dropped with the dead columns would show in the eigenvectors at
    // ~1e-7 relative, not 1e-13.  Then G is put back and the iteration runs on G itself, as it does above 2,048.
    DevBuf keep;
    KPOP_TRY(keep.alloc((uint64_t)n * n * 8));
    KPOP_HIP(hipMemcpyAsync(keep.p, d_G, (uint64_t)n * n * 8, hipMemcpyDeviceToDevice, st));
    uint32_t dead = 0;
    KPOP_TRY(cholesky_in_place(d_G, n, d_V, st, &dead));
    if (dead > 1 && !(ctx().tune_dbg & (1 << 29))) {  // (1 << 29: stay on the factor, for A/B)
      if (getenv("KPOP_JACOBI_TRACE")) fprintf(stderr, "[jacobi] %u pivots at the rounding floor: the iteration runs on G, not on its factor\n", dead);
      KPOP_HIP(hipMemcpyAsync(d_G, keep.p, (uint64_t)n * n * 8, hipMemcpyDeviceToDevice, st));
      KPOP_HIP(hipStreamSynchronize(st));
      on_factor = false;
    }
  }
  if (!on_factor) {
    jacobi_identity_kernel<<<dim3(std::min<uint32_t>(div_up((uint64_t)n * n, 256), 4096)), dim3(256), 0, st>>>(d_V, n);
    KPOP_LAUNCH_CHECK();
  }
  double w_before = 1.0;
  bool converged = false;
  for (int sweep = 0; sweep < 60; ++sweep) {
    KPOP_HIP(hipMemsetAsync(d_worst, 0, 8, st));
    if (blocked) {
      for (uint32_t step = 0; step + 1 < m_blk; ++step) {
        const int inner = (ctx().tune_dbg & 15) ? (int)(ctx().tune_dbg & 15) : 1;
        const dim3 grid(m_blk / 2), block(256);
        unsigned long long *w = d_worst;
        switch ((ctx().tune_dbg & 64) ? 0u : div_up(n, 256)) {  // (64: the looped kernel, for A/B)
#define KPOP_JACOBI_ROWS(R)                                                                                           \
  case R:                                                                                                             \
    if (on_factor) jacobi_block_step_reg_kernel<R, false><<<grid, block, 0, st>>>(d_G, d_V, n, m_blk, step, w, inner); \
    else jacobi_block_step_reg_kernel<R, true><<<grid, block, 0, st>>>(d_G, d_V, n, m_blk, step, w, inner);           \
    break;
          KPOP_JACOBI_ROWS(1) KPOP_JACOBI_ROWS(2) KPOP_JACOBI_ROWS(3) KPOP_JACOBI_ROWS(4)
          KPOP_JACOBI_ROWS(5) KPOP_JACOBI_ROWS(6) KPOP_JACOBI_ROWS(7) KPOP_JACOBI_ROWS(8)
#undef KPOP_JACOBI_ROWS
          default: jacobi_block_step_kernel<256><<<grid, block, 0, st>>>(d_G, d_V, n, m_blk, step, w, inner);
        }
        KPOP_LAUNCH_CHECK();
      }
    } else
    for (uint32_t step = 0; step + 1 < m; ++step) {
      jacobi_step_kernel<<<dim3(m / 2), dim3(256), 0, st>>>(d_G, d_V, n, m, step, d_worst);
      KPOP_LAUNCH_CHECK();
    }
    double w = 0.0;
    KPOP_HIP(hipMemcpyAsync(&w, d_worst, 8, hipMemcpyDeviceToHost, st));
    KPOP_HIP(hipStreamSynchronize(st));
    if (getenv("KPOP_JACOBI_TRACE")) fprintf(stderr, "[jacobi] sweep %d: largest |cos| between two columns %.3e\n", sweep, w);
    // done at 1e-15 -- or at the rounding floor, which for some matrices sits a little above that: below 1e-13 a sweep that
    // does not halve the largest cosine any more is rotating noise (the quadratic phase squares it every sweep)
    if (w < 1e-15 || (w < 1e-13 && w > 0.5 * w_before)) {
      converged = true;
      break;
    }
    w_before = w;
  }
  // (sixty sweeps have always been enough -- eight to twelve are typical --; a matrix that is not done by then is reported,
  // not passed on in silence: the caller's results are then only as orthogonal as the figure says)
  if (!converged)
    fprintf(stderr, "[kpop] warning: the one-sided Jacobi iteration stopped after 60 sweeps with |cos| = %.3e between two columns\n", w_before);
  if (on_factor) jacobi_unit_columns_kernel<<<dim3(n), dim3(256), 0, st>>>(d_G, n, d_V, d_lambda);
  else jacobi_colnorm_kernel<<<dim3(n), dim3(256), 0, st>>>(d_G, n, d_lambda);
  KPOP_LAUNCH_CHECK();
  return 0;
}

// Class positions and the projection W from the eigenvectors, on the device: twisted[j][d] = v_d[j] / sqrt(c_j) * sv_d and
// W[j][d] = v_d[j] / sv_d for the dimensions in decreasing order of eigenvalue (order[d] = the column of V).  Tiles of 32 x 32
// through LDS: V is read along j (its columns are contiguous), the outputs are written along d.
__global__ __launch_bounds__(256) void ca_positions_kernel(const double *__restrict__ V, uint32_t J, uint32_t nd, const uint32_t *__restrict__ order,
                                                           const double *__restrict__ sv, const double *__restrict__ c,
                                                           double *__restrict__ twisted, double *__restrict__ W) {
  __shared__ double tile[32][33];
  const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const uint32_t j0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t d = d0 + ty + 8 * q, j = j0 + tx;
    tile[ty + 8 * q][tx] = (d < nd && j < J) ? V[(uint64_t)order[d] * J + j] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t j = j0 + ty + 8 * q, d = d0 + tx;
    if (j < J && d < nd) {
      const double v = tile[tx][ty + 8 * q], s = sv[d];
      twisted[(uint64_t)j * nd + d] = v / sqrt(c[j]) * s;  // D_c^-1/2 V diag(sv)
      W[(uint64_t)j * nd + d] = s > 0.0 ? v / s : 0.0;     // V diag(1/sv)
    }
  }
}

// ---------------------------------------------------------------------------
// The analysis on device-resident data.  d_N: the counts (I x J, row-major); d_S: where the standardised table goes (may
// be d_N).  The small per-column and per-dimension work (J weights; the order of J eigenvalues) is host code between
// launches, so the stream is synchronised a few times on the way.
//
// The twister (nd x I, dims-major) is produced in slabs of kCaDimSlab dimensions -- U = S W[:, slab] on the matrix cores,
// then the scaled transpose -- so that the table never exists twice on the device and, for a host destination, the copy
// of one slab runs under the product of the next (a second stream; the rows of a dims-major slab are one contiguous run
// of the destination).
// ---------------------------------------------------------------------------
constexpr uint32_t kCaDimSlab = 256;

struct CaTimer {
  bool on = getenv("KPOP_TIMING") != nullptr;
  hipStream_t st = nullptr;
  std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
  void mark(const char *what) {
    if (!on) return;
    (void)hipStreamSynchronize(st);
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[kpop_ca] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - last).count());
    last = t;
  }
};

// A large pageable host buffer to the device: the runtime pins, copies and unpins what one call is given, one after the
// other, so pieces handed over by a few threads (a stream each) put the pinning of one under the transfer of another.
static int host_to_device_threads(void *dst, const void *src, uint64_t bytes) {
  const uint64_t piece = 256ull << 20;
  const char *e = getenv("KPOP_CA_COPY_THREADS");
  const unsigned n_threads = (unsigned)std::max(1, std::min(16, e ? atoi(e) : 4));
  if (bytes < 2 * piece || n_threads == 1) {
    KPOP_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return KPOP_OK;
  }
  int device = 0;
  KPOP_HIP(hipGetDevice(&device));
  std::atomic<uint64_t> next{0};
  std::atomic<int> failed{0};
  std::vector<std::thread> threads;
  for (unsigned t = 0; t < n_threads; ++t)
    threads.emplace_back([&, device] {
      hipStream_t s = nullptr;
      if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
        failed = 1;
        return;
      }
      for (;;) {
        const uint64_t off = next.fetch_add(piece);
        if (off >= bytes) break;
        const uint64_t len = std::min(piece, bytes - off);
        if (hipMemcpyAsync((char *)dst + off, (const char *)src + off, len, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
          failed = 1;
          break;
        }
      }
      (void)hipStreamDestroy(s);
    });
  for (auto &t : threads) t.join();
  if (failed) KPOP_FAIL(KPOP_ERR_HIP, "kpop_ca: copying the table to the device failed");
  return KPOP_OK;
}

struct CaOutput {          // where the results go: host pointers (kpop_ca) or device pointers (kpop_dev_ca)
  bool on_device = false;
  double *twisted = nullptr, *inertia = nullptr, *twister = nullptr;
};

static int ca_on_device(const double *d_N, double *d_S, uint64_t I, uint32_t J, int normalize, const CaOutput &out, hipStream_t st,
                        CaTimer &tm) {
  const uint32_t nd = (uint32_t)std::min<uint64_t>(I, J) - 1;
  const uint32_t n_slabs = div_up(I, kCaSlab);
  const uint32_t tiles = div_up(J, kGT) * div_up(J, kGT);
  uint32_t splits = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, 2048 / std::max(1u, tiles)), std::max<uint64_t>(1, I / 4096));
  splits = std::min(splits, 256u);  // of K = I in G = S'S
  const uint32_t n_dim_slabs = div_up(nd, kCaDimSlab);
  const uint32_t slab_w = std::min(nd, kCaDimSlab);
  DevBuf dW, dR, dC, dPart, dG, dGslabs, dWm, dV, dLambda, dWorst, dOrder, dSv, dTwisted, dU[1], dT[2];
  KPOP_TRY(dW.alloc((uint64_t)J * 8));
  KPOP_TRY(dR.alloc(I * 8));
  KPOP_TRY(dC.alloc((uint64_t)J * 8));
  KPOP_TRY(dPart.alloc((uint64_t)n_slabs * J * 8));
  KPOP_TRY(dG.alloc((uint64_t)J * J * 8));
  KPOP_TRY(dGslabs.alloc((uint64_t)splits * J * J * 8));
  KPOP_TRY(dV.alloc((uint64_t)J * J * 8));
  KPOP_TRY(dLambda.alloc((uint64_t)J * 8));
  KPOP_TRY(dWorst.alloc(8));
  KPOP_TRY(dWm.alloc((uint64_t)J * nd * 8));
  KPOP_TRY(dOrder.alloc((uint64_t)J * 4));
  KPOP_TRY(dSv.alloc((uint64_t)nd * 8));
  if (!out.on_device) KPOP_TRY(dTwisted.alloc((uint64_t)J * nd * 8));
  struct Events {
    hipEvent_t ready[2] = {nullptr, nullptr};
    ~Events() {
      for (int b = 0; b < 2; ++b)
        if (ready[b]) (void)hipEventDestroy(ready[b]);
    }
  } ss;
  KPOP_TRY(dU[0].alloc(I * slab_w * 8));
  if (!out.on_device) {
    for (int b = 0; b < 2; ++b) {
      KPOP_HIP(hipEventCreateWithFlags(&ss.ready[b], hipEventDisableTiming));
      if ((uint32_t)b < n_dim_slabs) KPOP_TRY(dT[b].alloc(I * slab_w * 8));
    }
  }
  // column sums -> weights w_j (P_ij = N_ij w_j)
  ca_col_partial_kernel<<<dim3(n_slabs), dim3(256), 0, st>>>(d_N, I, J, nullptr, dPart.as<double>());
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
  ca_row_mass_kernel<<<dim3(div_up(I, 4)), dim3(256), 0, st>>>(d_N, I, J, dW.as<double>(), dR.as<double>());
  KPOP_LAUNCH_CHECK();
  ca_standardise_kernel<<<dim3(std::min<uint32_t>(div_up(I * J, 256), 1u << 20)), dim3(256), 0, st>>>(
      d_N, I, J, dW.as<double>(), dR.as<double>(), dC.as<double>(), d_S);
  KPOP_LAUNCH_CHECK();
  tm.mark("masses, standardise");
  // G = S'S  (J x J), K = I split over up to 256 slabs
  KPOP_TRY(gemm_f64<true>(d_S, J, d_S, J, dG.as<double>(), J, J, I, splits, dGslabs.as<double>(), 1, st));
  tm.mark("G = S'S");
  KPOP_TRY(jacobi_eigen_psd_device(dG.as<double>(), dV.as<double>(), J, dLambda.as<double>(), dWorst.as<unsigned long long>(), st));
  tm.mark("Jacobi");
  std::vector<double> lambda(J);
  KPOP_HIP(hipMemcpyAsync(lambda.data(), dLambda.p, (uint64_t)J * 8, hipMemcpyDeviceToHost, st));
  KPOP_HIP(hipStreamSynchronize(st));
  // order by decreasing eigenvalue; sv = sqrt(lambda): J numbers, host work.  The J x nd class positions and the projection
  // are built from the eigenvectors where they are.
  std::vector<uint32_t> order(J);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return lambda[a] > lambda[b]; });
  std::vector<double> sv(nd), h_inertia(nd);
  double sum_sq = 0.0;
  for (uint32_t d = 0; d < nd; ++d) {
    sv[d] = sqrt(std::max(lambda[order[d]], 0.0));
    sum_sq += sv[d] * sv[d];
  }
  for (uint32_t d = 0; d < nd; ++d) h_inertia[d] = sum_sq > 0.0 ? sv[d] * sv[d] / sum_sq : 0.0;
  KPOP_HIP(hipMemcpyAsync(dOrder.p, order.data(), (uint64_t)J * 4, hipMemcpyHostToDevice, st));
  KPOP_HIP(hipMemcpyAsync(dSv.p, sv.data(), (uint64_t)nd * 8, hipMemcpyHostToDevice, st));
  double *d_twisted = out.on_device ? out.twisted : dTwisted.as<double>();
  ca_positions_kernel<<<dim3(div_up(J, 32), div_up(nd, 32)), dim3(256), 0, st>>>(dV.as<double>(), J, nd, dOrder.as<uint32_t>(), dSv.as<double>(),
                                                                                 dC.as<double>(), d_twisted, dWm.as<double>());
  KPOP_LAUNCH_CHECK();
  if (out.on_device) KPOP_HIP(hipMemcpyAsync(out.inertia, h_inertia.data(), (uint64_t)nd * 8, hipMemcpyHostToDevice, st));
  else {
    std::copy(h_inertia.begin(), h_inertia.end(), out.inertia);
    KPOP_HIP(hipMemcpyAsync(out.twisted, d_twisted, (uint64_t)J * nd * 8, hipMemcpyDeviceToHost, st));
  }
  KPOP_HIP(hipStreamSynchronize(st));  // (order, sv and h_inertia are the sources of copies in flight)
  tm.mark("V back, order, W");
  // U = S W (I x nd) slab by slab, then the twister = (D_r^-1/2 U)'
  const uint32_t M = (uint32_t)std::min<uint64_t>(I, 0xFFFFFFFFull);
  if (out.on_device) {
    for (uint32_t s = 0; s < n_dim_slabs; ++s) {
      const uint32_t d0 = s * kCaDimSlab, wd = std::min(kCaDimSlab, nd - d0);
      KPOP_TRY(gemm_f64<false>(d_S, J, dWm.as<double>() + d0, nd, dU[0].as<double>(), M, wd, J, 1, nullptr, 0, st));
      ca_row_coords_kernel<<<dim3(div_up(I, 32), div_up(wd, 32)), dim3(256), 0, st>>>(dU[0].as<double>(), I, wd, dR.as<double>(),
                                                                                     out.twister + (uint64_t)d0 * I);
      KPOP_LAUNCH_CHECK();
    }
    KPOP_HIP(hipStreamSynchronize(st));  // (the host vectors above are the source of copies in flight)
    tm.mark("U = S W, row coordinates");
    return KPOP_OK;
  }
  KPOP_HIP(hipStreamSynchronize(st));  // W is on the device; the copiers' streams are not ordered after `st`
  // The destination is usually memory the caller has just allocated: whoever writes a page first takes its fault, and one
  // thread copying takes them at 17 GB/s.  Two threads, a stream each, copy half of every slab: 6.9 GB in 0.35 s instead
  // of 0.43 s (more threads: no better; 0.14 s when the pages are present already).  Touching the pages ahead of time from
  // other threads, under the analysis, was measured first: the device sat idle until they were done -- 228-245 ms at
  // 6.9 GB, whatever their number -- and nothing was gained.
  struct Progress {
    std::mutex m;
    std::condition_variable cv;
    uint32_t produced = 0;            // slabs whose kernels are enqueued and whose `ready` event is recorded
    uint32_t parts_done[2] = {0, 0};  // copiers finished with the slab in buffer b
    uint32_t copied = 0;              // slabs wholly on the host
    int failed = 0;
  } pg;
  int device = 0;
  KPOP_HIP(hipGetDevice(&device));
  const char *e_cp = getenv("KPOP_CA_COPY_THREADS");
  const unsigned n_copiers = (unsigned)std::max(1, std::min(16, e_cp ? atoi(e_cp) : 2));
  auto produce = [&](uint32_t s) -> int {
    const int b = s & 1;
    const uint32_t d0 = s * kCaDimSlab, wd = std::min(kCaDimSlab, nd - d0);
    KPOP_TRY(gemm_f64<false>(d_S, J, dWm.as<double>() + d0, nd, dU[0].as<double>(), M, wd, J, 1, nullptr, 0, st));
    ca_row_coords_kernel<<<dim3(div_up(I, 32), div_up(wd, 32)), dim3(256), 0, st>>>(dU[0].as<double>(), I, wd, dR.as<double>(),
                                                                                   dT[b].as<double>());
    KPOP_LAUNCH_CHECK();
    KPOP_HIP(hipEventRecord(ss.ready[b], st));
    {
      std::lock_guard<std::mutex> lk(pg.m);
      pg.produced = s + 1;
    }
    pg.cv.notify_all();
    return 0;
  };
  std::vector<std::thread> copiers;
  for (unsigned t = 0; t < n_copiers; ++t)
    copiers.emplace_back([&, t, device] {
      hipStream_t cs = nullptr;
      bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
      for (uint32_t s = 0; s < n_dim_slabs; ++s) {
        const int b = s & 1;
        const uint32_t d0 = s * kCaDimSlab, wd = std::min(kCaDimSlab, nd - d0);
        {
          std::unique_lock<std::mutex> lk(pg.m);
          pg.cv.wait(lk, [&] { return pg.produced > s || pg.failed; });
          if (pg.failed) break;
        }
        const uint64_t bytes = (uint64_t)wd * I * 8, share = ((bytes / n_copiers) + 4095) & ~4095ull;
        const uint64_t lo = std::min(bytes, share * t), hi = (t + 1 == n_copiers) ? bytes : std::min(bytes, share * (t + 1));
        if (ok && hi > lo)
          ok = hipStreamWaitEvent(cs, ss.ready[b], 0) == hipSuccess &&
               hipMemcpyAsync((char *)(out.twister + (uint64_t)d0 * I) + lo, (const char *)dT[b].p + lo, hi - lo, hipMemcpyDeviceToHost, cs) == hipSuccess &&
               hipStreamSynchronize(cs) == hipSuccess;
        {
          std::lock_guard<std::mutex> lk(pg.m);
          if (!ok) pg.failed = 1;
          if (++pg.parts_done[b] == n_copiers) {
            pg.parts_done[b] = 0;
            pg.copied = s + 1;
          }
        }
        pg.cv.notify_all();
        if (!ok) break;
      }
      if (cs) (void)hipStreamDestroy(cs);
    });
  int rc = produce(0);
  if (rc == 0 && n_dim_slabs > 1) rc = produce(1);
  for (uint32_t s = 0; rc == 0 && s + 2 < n_dim_slabs; ++s) {
    {
      std::unique_lock<std::mutex> lk(pg.m);
      pg.cv.wait(lk, [&] { return pg.copied > s || pg.failed; });  // buffer s & 1 is free again
      if (pg.failed) break;
    }
    rc = produce(s + 2);
  }
  if (rc != 0) {
    {
      std::lock_guard<std::mutex> lk(pg.m);
      pg.failed = 1;
    }
    pg.cv.notify_all();
  }
  for (auto &t : copiers) t.join();
  if (rc != 0) return rc;
  if (pg.failed) KPOP_FAIL(KPOP_ERR_HIP, "kpop_ca: copying the twister to the host failed");
  KPOP_HIP(hipStreamSynchronize(st));
  tm.mark("U = S W, twister to the host");
  return KPOP_OK;
}

// rows of a row-major table gathered into another (one wavefront per row)
__global__ __launch_bounds__(256) void table_gather_rows_kernel(const double *__restrict__ table, uint32_t J, const uint64_t *__restrict__ rows,
                                                                uint64_t n_sel, double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint64_t stride = (uint64_t)gridDim.x * 4;
  for (uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n_sel; i += stride) {
    const double *src = table + rows[i] * J;
    double *dst = out + i * J;
    for (uint32_t j = lane; j < J; j += 64) dst[j] = src[j];
  }
}

}  // namespace kpop

using namespace kpop;

static int ca_check(const void *counts, uint64_t I, uint32_t J, const void *n_dims_out, const void *twisted, const void *inertia,
                    const void *twister, const char *who) {
  if (!counts || !n_dims_out || !twisted || !inertia || !twister) KPOP_FAIL(KPOP_ERR_INVALID, "%s: null argument", who);
  if (I < 2 || J < 2) KPOP_FAIL(KPOP_ERR_INVALID, "%s: need at least 2 k-mers and 2 spectra", who);
  return KPOP_OK;
}

extern "C" int kpop_ca(const double *counts, uint64_t n_kmers, uint32_t n_spectra, int normalize, uint32_t *n_dims_out,
                       double *twisted, double *inertia, double *twister) {
  KPOP_TRY(require_init());
  const uint64_t I = n_kmers;
  const uint32_t J = n_spectra;
  KPOP_TRY(ca_check(counts, I, J, n_dims_out, twisted, inertia, twister, "kpop_ca"));
  const uint32_t nd = (uint32_t)std::min<uint64_t>(I, J) - 1;
  *n_dims_out = nd;
  hipStream_t st = nullptr;
  CaTimer tm;
  CaOutput out;
  out.twisted = twisted;
  out.inertia = inertia;
  out.twister = twister;
  int rc;
  {
    DevBuf dS;  // the counts, standardised in place
    KPOP_TRY(dS.alloc(I * J * 8));
    tm.mark("alloc");
    KPOP_TRY(host_to_device_threads(dS.p, counts, I * J * 8));
    tm.mark("counts to the device");
    rc = ca_on_device(dS.as<double>(), dS.as<double>(), I, J, normalize, out, st, tm);
    tm.mark("release of the work buffers");
  }
  tm.mark("release of the table");
  return rc;
}

extern "C" uint64_t kpop_dev_ca_workspace_bytes(uint64_t n_kmers, uint32_t n_spectra) {
  return n_kmers * n_spectra * 8 + 256;  // the standardised table (the counts are not modified)
}

extern "C" int kpop_dev_ca(const double *d_counts, uint64_t n_kmers, uint32_t n_spectra, int normalize, void *d_work,
                           uint32_t *n_dims_out, double *d_twisted, double *d_inertia, double *d_twister, void *stream) {
  KPOP_TRY(require_init());
  KPOP_TRY(ca_check(d_counts, n_kmers, n_spectra, n_dims_out, d_twisted, d_inertia, d_twister, "kpop_dev_ca"));
  if (!d_work) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_ca: null workspace");
  *n_dims_out = (uint32_t)std::min<uint64_t>(n_kmers, n_spectra) - 1;
  CaTimer tm;
  tm.st = (hipStream_t)stream;
  CaOutput out;
  out.on_device = true;
  out.twisted = d_twisted;
  out.inertia = d_inertia;
  out.twister = d_twister;
  // d_work == d_counts: standardise the table where it stands (it is overwritten)
  double *d_S = d_work == (const void *)d_counts ? const_cast<double *>(d_counts)
                                                 : reinterpret_cast<double *>(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
  return ca_on_device(d_counts, d_S, n_kmers, n_spectra, normalize, out, (hipStream_t)stream, tm);
}

// ---------------------------------------------------------------- a k-mers x spectra table that stays on the device
// (the pieces KPopTwist needs between kpop_dev_counter_transform and kpop_dev_ca: src/KPopTwist:76-91 selects k-mers by a
// list, a sample and the row sums before the analysis)
extern "C" int kpop_dev_table_row_sums(const double *d_table, uint64_t n_rows, uint32_t n_cols, double *d_out, void *stream) {
  KPOP_TRY(require_init());
  if (!d_table || !d_out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_table_row_sums: null argument");
  if (n_rows == 0) return KPOP_OK;
  ca_row_mass_kernel<<<dim3(div_up(n_rows, 4)), dim3(256), 0, (hipStream_t)stream>>>(d_table, n_rows, n_cols, nullptr, d_out);
  KPOP_LAUNCH_CHECK();
  return KPOP_OK;
}

extern "C" int kpop_dev_table_col_sums(const double *d_table, uint64_t n_rows, uint32_t n_cols, double *d_out, void *stream) {
  KPOP_TRY(require_init());
  if (!d_table || !d_out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_table_col_sums: null argument");
  if (n_cols == 0) return KPOP_OK;
  const uint32_t n_slabs = std::max<uint32_t>(1, div_up(n_rows, kCaSlab));
  DevBuf part;
  KPOP_TRY(part.alloc((uint64_t)n_slabs * n_cols * 8));
  hipStream_t st = (hipStream_t)stream;
  ca_col_partial_kernel<<<dim3(n_slabs), dim3(256), 0, st>>>(d_table, n_rows, n_cols, nullptr, part.as<double>());
  KPOP_LAUNCH_CHECK();
  ca_col_final_kernel<<<dim3(div_up(n_cols, 256)), dim3(256), 0, st>>>(part.as<double>(), n_slabs, n_cols, d_out);
  KPOP_LAUNCH_CHECK();
  KPOP_HIP(hipStreamSynchronize(st));  // (the partial sums are freed on return)
  return KPOP_OK;
}

extern "C" int kpop_dev_table_gather_rows(const double *d_table, uint32_t n_cols, const uint64_t *d_rows, uint64_t n_sel, double *d_out,
                                          void *stream) {
  KPOP_TRY(require_init());
  if (n_sel == 0) return KPOP_OK;
  if (!d_table || !d_rows || !d_out) KPOP_FAIL(KPOP_ERR_INVALID, "kpop_dev_table_gather_rows: null argument");
  table_gather_rows_kernel<<<dim3((uint32_t)std::min<uint64_t>(div_up(n_sel, 4), 1u << 20)), dim3(256), 0, (hipStream_t)stream>>>(
      d_table, n_cols, d_rows, n_sel, d_out);
  KPOP_LAUNCH_CHECK();
  return KPOP_OK;
}

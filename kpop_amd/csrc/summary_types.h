// summary_types.h -- what the large-reference summary's kernels hand one another per query row (summary_large.hip), and what
// the matrix-core path's refinement reads of it (distance_mfma.hip).
#pragma once
#include <cstdint>

namespace kpop {

constexpr uint32_t kNbCap = 16384;  // room of a row's neighbour candidates

struct RowInfo {   // written by the sample / finish kernels, read by the passes
  double m_hat, median, mean, sd;
  uint64_t klo, khi, kcut;   // brackets of the median (keys of d), neighbour threshold
  uint64_t mlo, mhi;         // bracket of the MAD (keys of |d - median|)
  uint32_t sample_n, sample_stride;
};
struct RowCounts {
  uint32_t lt_lo, eq_lo, n_cand, eq_hi, n_nb;       // pass 1
  uint32_t m_lt, m_eqlo, m_cand, m_eqhi;            // pass 2 (the one-pass path: m_lt = inside the median's bracket, m_eqlo = in the inner region)
  uint32_t fail, pad0, pad1;
};
struct FusedThr {  // one query row's thresholds, as distances (64 bytes)
  double lo, hi;            // the median's bracket (-inf / +inf: none)
  double Llo, Lin, Uin, Uhi;  // lower band [Llo, Lin], inner region (Lin, Uin), upper band [Uin, Uhi]
  double cut, mhat;
};

// the one-kernel paths' bookkeeping per (query row, STRIPE of kStripe reference rows): the vector-pipe kernel of summary_large.hip and the
// matrix-core kernel of distance_mfma.hip write it, fused_finish_kernel<true> reads it
constexpr uint32_t kStripe = 2048, kMaxStripes = 8192;
struct StripeRec {
  uint32_t lt_eqlo, eqhi_nmed, inner, c_cnt;  // (16 bits each where paired: a stripe has 2,048 elements)
};

// the one-pass path's per-row lists, for a caller that wants to look things up in them (nullptr members: that path did not run)
struct SummaryLists {
  const RowInfo *info = nullptr;
  const FusedThr *thr = nullptr;
  const RowCounts *cnt = nullptr;
  const double *cand = nullptr;     // [rows][cap]: the values inside the median's bracket or in the MAD's bands
  const uint32_t *cand_i = nullptr;  // ... and their columns
  uint32_t cap = 0;
  const uint32_t *nb_idx = nullptr;  // [rows][kNbCap]: the columns and values at or below the neighbours' threshold
  const double *nb_d = nullptr;
  const uint32_t *n_failed = nullptr;  // the device word that counts the rows whose brackets or bands missed (they went through the ten-pass kernel)
};

}  // namespace kpop

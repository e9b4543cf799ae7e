// space_ops.h -- the per-dimension term and the final scale of a distance (lib/Space.ml:150-165), shared by the distance
// kernels (distance.hip) and the summary that computes its distances as it goes (summary_large.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/kpop_hip.h"

namespace kpop {

// unscaled component and scale (lib/Space.ml:150-165)
template <int KIND>
__device__ __forceinline__ double component(double diff, double m, double p) {
  if (KIND == KPOP_MINKOWSKI) return __dmul_rn(pow(fabs(diff), p), m);
  return __dmul_rn(__dmul_rn(diff, diff), m);
}

template <int KIND>
__device__ __forceinline__ double scale_distance(double x, double p) {
  if (KIND == KPOP_EUCLIDEAN) return sqrt(x);
  if (KIND == KPOP_COSINE) return x / 2.0;
  return pow(x, 1.0 / p);
}

}  // namespace kpop

// select_probe.cpp -- block_select_ranks (kpop_amd/csrc/summary_large.hip) against std::nth_element on rows with tie
// groups of thousands: hipcc -O2 --offload-arch=gfx950 -I kpop_amd/csrc -I include tools/probes/select_probe.cpp -o select_probe
#include "../../kpop_amd/csrc/summary_large.hip"

#include <algorithm>
#include <cmath>
#include <random>
#include <vector>

using namespace kpop;

__global__ __launch_bounds__(kLT) void probe_kernel(const double *row, uint32_t n, double centre, uint32_t rank, double far, double *out, uint32_t *out_cnt) {
  __shared__ uint32_t s_hist[kSel * kBins];
  __shared__ uint64_t s_cand[kSel * kCand];
  __shared__ uint32_t s_misc[64];
  Sel sm[1] = {Sel{rank, f64_key(0.0), f64_key(far), 0, 0, 0, 0, 0}};
  block_select_ranks<1>(PlainRow{row}, n, centre, sm, 1, s_hist, s_cand, s_misc);
  if (threadIdx.x == 0) {
    out[0] = key_f64(sm[0].value);
    out_cnt[0] = sm[0].n_less;
    out_cnt[1] = sm[0].n_equal;
    out_cnt[2] = sm[0].done;
  }
}

int main() {
  std::mt19937_64 g(6);
  std::normal_distribution<double> N(1.0, 0.2);
  for (uint32_t n : {5000u, 20000u, 66000u, 200003u}) {
    std::vector<double> d(n);
    for (auto &x : d) x = std::round(std::fabs(N(g)) * 100.0) / 100.0;
    const double centre = 1.0;
    std::vector<double> t(n);
    double far = 0;
    for (uint32_t i = 0; i < n; ++i) {
      t[i] = std::fabs(d[i] - centre);
      far = std::max(far, t[i]);
    }
    std::vector<double> s = t;
    std::nth_element(s.begin(), s.begin() + n / 2, s.end());
    double *dd, *dout;
    uint32_t *dc;
    hipMalloc(&dd, n * 8);
    hipMalloc(&dout, 8);
    hipMalloc(&dc, 16);
    hipMemcpy(dd, d.data(), n * 8, hipMemcpyHostToDevice);
    probe_kernel<<<1, kLT>>>(dd, n, centre, n / 2, far, dout, dc);
    double got;
    uint32_t c[3];
    hipMemcpy(&got, dout, 8, hipMemcpyDeviceToHost);
    hipMemcpy(c, dc, 12, hipMemcpyDeviceToHost);
    printf("n %u: want %.17g got %.17g (n_less %u n_equal %u done %u) %s\n", n, s[n / 2], got, c[0], c[1], c[2], got == s[n / 2] ? "ok" : "WRONG");
  }
  return 0;
}

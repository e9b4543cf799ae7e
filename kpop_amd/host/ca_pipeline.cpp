// ca_pipeline.cpp -- see ca_pipeline.h
#include "ca_pipeline.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <thread>
#include <tuple>
#include <unordered_map>

#include "../../include/kpop_hip.h"

namespace kpop_host {

namespace {

uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

}  // namespace

std::vector<std::string> read_lines(const std::string &path) {
  std::vector<std::string> out;
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) throw Error("cannot open '" + path + "'");
  char *buf = nullptr;
  size_t cap = 0;
  ssize_t n;
  while ((n = getline(&buf, &cap, f)) >= 0) {
    while (n > 0 && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) --n;
    out.emplace_back(buf, (size_t)n);
  }
  free(buf);
  fclose(f);
  return out;
}

// [4/16] keep list, [5/16] resampling, [6/16] thresholding (src/KPopTwist:76-91): the rows of the table that go into the
// analysis, in order.  row_sum(r) = the sum of row r of the table.
static std::vector<size_t> select_kmers(const std::vector<std::string> &kmers, const CaParams &P, const std::function<double(size_t)> &row_sum) {
  const size_t I = kmers.size();
  std::vector<size_t> sel(I);
  for (size_t i = 0; i < I; ++i) sel[i] = i;
  if (!P.keep_path.empty()) {
    std::unordered_map<std::string, size_t> idx;
    for (size_t i = 0; i < I; ++i) idx[kmers[i]] = i;
    sel.clear();
    for (const std::string &nm : read_lines(P.keep_path)) {
      auto it = idx.find(strip_external_quotes_and_check(nm));
      if (it == idx.end()) throw Error("k-mer '" + nm + "' of the keep list is not in the table");
      sel.push_back(it->second);
    }
  }
  if (P.fraction < 1.0) {
    const size_t want = (size_t)((double)sel.size() * P.fraction);
    std::vector<std::pair<uint64_t, size_t>> keyed;
    for (size_t i = 0; i < sel.size(); ++i) keyed.push_back({mix64(0x4B506F70ull + i), i});
    std::sort(keyed.begin(), keyed.end());
    std::vector<size_t> pick;
    for (size_t i = 0; i < want; ++i) pick.push_back(keyed[i].second);
    std::sort(pick.begin(), pick.end());
    std::vector<size_t> ns;
    for (size_t i : pick) ns.push_back(sel[i]);
    sel.swap(ns);
  }
  {
    std::vector<double> rsum(sel.size(), 0.0);
    parallel_for(sel.size(), 4096, [&](size_t lo, size_t hi) {
      for (size_t r = lo; r < hi; ++r) rsum[r] = row_sum(sel[r]);
    });
    double mx = 0.0;
    for (size_t r = 0; r < sel.size(); ++r) mx = std::max(mx, rsum[r]);
    std::vector<size_t> ns;
    for (size_t r = 0; r < sel.size(); ++r)
      if (rsum[r] >= mx * P.threshold) ns.push_back(sel[r]);
    sel.swap(ns);
  }
  return sel;
}

// principal row coordinates = standard ones x sv.  sv_d is not recoverable from the normalised inertia, so it is recomputed
// from the class positions: for principal column coordinates sum_j c_j G_jd^2 = sv_d^2, c_j the column masses (1/J once the
// columns are normalised).
static void fill_kmer_coords(CaResult &R, const std::vector<std::string> &knames, const std::vector<double> &colsum, bool normalize) {
  const size_t J = colsum.size(), nd = R.inertia.data.size(), I = knames.size();
  double total = 0.0;
  for (size_t j = 0; j < J; ++j) total += normalize ? 1.0 : colsum[j];
  R.kmer_coords.col_names = R.inertia.col_names;
  R.kmer_coords.row_names = knames;
  R.kmer_coords.data.resize(I * nd);
  for (size_t d = 0; d < nd; ++d) {
    double sv2 = 0.0;
    for (size_t j = 0; j < J; ++j) sv2 += (normalize ? 1.0 : colsum[j]) / total * R.twisted.data[j * nd + d] * R.twisted.data[j * nd + d];
    const double sv = sqrt(sv2);
    for (size_t r = 0; r < I; ++r) R.kmer_coords.data[r * nd + d] = R.twister.data[d * I + r] * sv;
  }
}

static void name_results(CaResult &R, const std::vector<std::string> &spectra, size_t I, size_t J) {
  const size_t nd = std::min(I, J) - 1;
  std::vector<std::string> dims(nd);
  for (size_t d = 0; d < nd; ++d) dims[d] = "Dim" + std::to_string(d + 1);
  R.twisted.col_names = dims;
  R.twisted.row_names = spectra;
  R.twisted.data.resize(J * nd);
  R.inertia.col_names = dims;
  R.inertia.row_names = {"inertia"};
  R.inertia.data.resize(nd);
  R.twister.row_names = dims;
  R.twister.data.resize(nd * I);
}

CaResult run_ca(std::vector<std::string> kmers, const std::vector<std::string> &spectra, DVec N, const CaParams &P) {
  const size_t J = spectra.size();
  size_t I = kmers.size();
  if (N.size() != I * J) throw Error("count table is not k-mers x spectra");
  std::vector<size_t> sel = select_kmers(kmers, P, [&](size_t r) {
    double t = 0.0;
    for (size_t j = 0; j < J; ++j) t += N[r * J + j];
    return t;
  });
  stage_mark("KPopTwist", "  k-mers selected");
  // the selected rows, in order: the table itself when every row is selected (the usual case: nothing to copy)
  bool all_rows = sel.size() == I;
  for (size_t r = 0; all_rows && r < I; ++r) all_rows = sel[r] == r;
  DVec M;
  std::vector<std::string> knames;
  if (all_rows) {
    M.swap(N);
    knames.swap(kmers);
  } else {
    M.resize(sel.size() * J);
    knames.resize(sel.size());
    parallel_for(sel.size(), 4096, [&](size_t lo, size_t hi) {
      for (size_t r = lo; r < hi; ++r) {
        memcpy(&M[r * J], &N[sel[r] * J], J * sizeof(double));
        knames[r] = kmers[sel[r]];
      }
    });
    DVec().swap(N);
  }
  I = sel.size();
  if (I < 2 || J < 2) throw Error("correspondence analysis needs at least 2 k-mers and 2 spectra");
  if (P.verbose) fprintf(stderr, "[8/16] Twisting counts (%zu k-mers x %zu spectra) on the GPU...\n", I, J);
  CaResult R;
  name_results(R, spectra, I, J);
  if (P.want_kmer_coords) R.twister.col_names = knames;  // (the names are needed once more below)
  else R.twister.col_names = std::move(knames);
  stage_mark("KPopTwist", "  table and result buffers laid out");
  uint32_t nd_out = 0;
  if (kpop_ca(M.data(), I, (uint32_t)J, P.normalize ? 1 : 0, &nd_out, R.twisted.data.data(), R.inertia.data.data(),
              R.twister.data.data()) != 0)
    throw Error(std::string("libkpop_hip: ") + kpop_last_error());
  stage_mark("KPopTwist", "  kpop_ca");
  if (P.want_kmer_coords) {
    std::vector<double> colsum(J, 0.0);
    for (size_t r = 0; r < I; ++r)
      for (size_t j = 0; j < J; ++j) colsum[j] += M[r * J + j];
    fill_kmer_coords(R, knames, colsum, P.normalize);
  }
  // the table is garbage now (gigabytes of it): a thread of its own gives it back while the results are being written
  auto *garbage = new std::tuple<DVec, DVec, std::vector<std::string>, std::vector<std::string>>(std::move(M), std::move(N), std::move(knames),
                                                                                              std::move(kmers));
  std::thread([garbage] { delete garbage; }).detach();
  return R;
}

namespace {
void chk(int rc) {
  if (rc != 0) throw Error(std::string("libkpop_hip: ") + kpop_last_error());
}
struct DeviceBlock {  // device memory through the C ABI, freed on scope exit
  void *p = nullptr;
  DeviceBlock() = default;
  DeviceBlock(const DeviceBlock &) = delete;
  DeviceBlock &operator=(const DeviceBlock &) = delete;
  ~DeviceBlock() { release(); }
  void alloc(uint64_t bytes) {
    release();
    chk(kpop_dev_malloc(&p, bytes ? bytes : 8));
  }
  void release() {
    if (p) (void)kpop_dev_free(p);
    p = nullptr;
  }
  void adopt(void *q) {
    release();
    p = q;
  }
  double *f64() const { return static_cast<double *>(p); }
};
}  // namespace

CaResult run_ca_device(std::vector<std::string> kmers, const std::vector<std::string> &spectra, void *d_table_owned, const CaParams &P) {
  DeviceBlock table;
  table.adopt(d_table_owned);
  const size_t J = spectra.size();
  size_t I = kmers.size();
  std::vector<double> sums(I);
  {
    DeviceBlock d_sums;
    d_sums.alloc(I * 8);
    chk(kpop_dev_table_row_sums(table.f64(), I, (uint32_t)J, d_sums.f64(), nullptr));
    chk(kpop_memcpy_d2h(sums.data(), d_sums.p, I * 8));
  }
  std::vector<size_t> sel = select_kmers(kmers, P, [&](size_t r) { return sums[r]; });
  stage_mark("KPopTwist", "  k-mers selected");
  bool all_rows = sel.size() == I;
  for (size_t r = 0; all_rows && r < I; ++r) all_rows = sel[r] == r;
  std::vector<std::string> knames;
  if (all_rows) knames.swap(kmers);
  else {
    knames.resize(sel.size());
    for (size_t r = 0; r < sel.size(); ++r) knames[r] = kmers[sel[r]];
    std::vector<uint64_t> rows(sel.begin(), sel.end());
    DeviceBlock d_rows, picked;
    d_rows.alloc(rows.size() * 8);
    chk(kpop_memcpy_h2d(d_rows.p, rows.data(), rows.size() * 8));
    picked.alloc((uint64_t)sel.size() * J * 8);
    chk(kpop_dev_table_gather_rows(table.f64(), (uint32_t)J, static_cast<const uint64_t *>(d_rows.p), sel.size(), picked.f64(), nullptr));
    chk(kpop_synchronize(nullptr));
    table.release();
    table.adopt(picked.p);
    picked.p = nullptr;
  }
  I = sel.size();
  if (I < 2 || J < 2) throw Error("correspondence analysis needs at least 2 k-mers and 2 spectra");
  if (P.verbose) fprintf(stderr, "[8/16] Twisting counts (%zu k-mers x %zu spectra) on the GPU...\n", I, J);
  const size_t nd = std::min(I, J) - 1;
  CaResult R;
  name_results(R, spectra, I, J);
  std::vector<double> colsum;
  if (P.want_kmer_coords) {  // (before the analysis: it standardises the table where it stands)
    colsum.resize(J);
    DeviceBlock d_cs;
    d_cs.alloc(J * 8);
    chk(kpop_dev_table_col_sums(table.f64(), I, (uint32_t)J, d_cs.f64(), nullptr));
    chk(kpop_memcpy_d2h(colsum.data(), d_cs.p, J * 8));
  }
  stage_mark("KPopTwist", "  table and result buffers laid out");
  {
    DeviceBlock d_twisted, d_inertia, d_twister;
    d_twisted.alloc((uint64_t)J * nd * 8);
    d_inertia.alloc((uint64_t)nd * 8);
    d_twister.alloc((uint64_t)nd * I * 8);
    uint32_t nd_out = 0;
    chk(kpop_dev_ca(table.f64(), I, (uint32_t)J, P.normalize ? 1 : 0, table.p, &nd_out, d_twisted.f64(), d_inertia.f64(), d_twister.f64(), nullptr));
    table.release();
    stage_mark("KPopTwist", "  kpop_dev_ca");
    chk(kpop_memcpy_d2h(R.twisted.data.data(), d_twisted.p, (uint64_t)J * nd * 8));
    chk(kpop_memcpy_d2h(R.inertia.data.data(), d_inertia.p, (uint64_t)nd * 8));
    chk(kpop_memcpy_d2h(R.twister.data.data(), d_twister.p, (uint64_t)nd * I * 8));
    stage_mark("KPopTwist", "  results to the host");
  }
  if (P.want_kmer_coords) {
    R.twister.col_names = knames;
    fill_kmer_coords(R, knames, colsum, P.normalize);
  } else
    R.twister.col_names = std::move(knames);
  return R;
}

}  // namespace kpop_host

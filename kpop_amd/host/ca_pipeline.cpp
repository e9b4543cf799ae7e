// ca_pipeline.cpp -- see ca_pipeline.h
#include "ca_pipeline.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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

CaResult run_ca(std::vector<std::string> kmers, const std::vector<std::string> &spectra, DVec N, const CaParams &P) {
  const size_t J = spectra.size();
  size_t I = kmers.size();
  if (N.size() != I * J) throw Error("count table is not k-mers x spectra");
  // [4/16] keep list, [5/16] resampling, [6/16] thresholding
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
      for (size_t r = lo; r < hi; ++r) {
        double t = 0.0;
        for (size_t j = 0; j < J; ++j) t += N[sel[r] * J + j];
        rsum[r] = t;
      }
    });
    double mx = 0.0;
    for (size_t r = 0; r < sel.size(); ++r) mx = std::max(mx, rsum[r]);
    std::vector<size_t> ns;
    for (size_t r = 0; r < sel.size(); ++r)
      if (rsum[r] >= mx * P.threshold) ns.push_back(sel[r]);
    sel.swap(ns);
  }
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
  const size_t nd = std::min(I, J) - 1;
  CaResult R;
  std::vector<std::string> dims(nd);
  for (size_t d = 0; d < nd; ++d) dims[d] = "Dim" + std::to_string(d + 1);
  R.twisted.col_names = dims;
  R.twisted.row_names = spectra;
  R.twisted.data.resize(J * nd);
  R.inertia.col_names = dims;
  R.inertia.row_names = {"inertia"};
  R.inertia.data.resize(nd);
  if (P.want_kmer_coords) R.twister.col_names = knames;  // (the names are needed once more below)
  else R.twister.col_names = std::move(knames);
  R.twister.row_names = dims;
  R.twister.data.resize(nd * I);
  stage_mark("KPopTwist", "  table and result buffers laid out");
  uint32_t nd_out = 0;
  if (kpop_ca(M.data(), I, (uint32_t)J, P.normalize ? 1 : 0, &nd_out, R.twisted.data.data(), R.inertia.data.data(),
              R.twister.data.data()) != 0)
    throw Error(std::string("libkpop_hip: ") + kpop_last_error());
  stage_mark("KPopTwist", "  kpop_ca");
  if (P.want_kmer_coords) {
    // principal row coordinates = standard ones x sv.  sv_d is not recoverable from the normalised inertia, so it is
    // recomputed from the class positions: for principal column coordinates sum_j c_j G_jd^2 = sv_d^2, c_j the column
    // masses (1/J once the columns are normalised).
    std::vector<double> colsum(J, 0.0);
    double total = 0.0;
    for (size_t r = 0; r < I; ++r)
      for (size_t j = 0; j < J; ++j) colsum[j] += M[r * J + j];
    for (size_t j = 0; j < J; ++j) total += P.normalize ? 1.0 : colsum[j];
    R.kmer_coords.col_names = dims;
    R.kmer_coords.row_names = knames;
    R.kmer_coords.data.resize(I * nd);
    for (size_t d = 0; d < nd; ++d) {
      double sv2 = 0.0;
      for (size_t j = 0; j < J; ++j)
        sv2 += (P.normalize ? 1.0 : colsum[j]) / total * R.twisted.data[j * nd + d] * R.twisted.data[j * nd + d];
      const double sv = sqrt(sv2);
      for (size_t r = 0; r < I; ++r) R.kmer_coords.data[r * nd + d] = R.twister.data[d * I + r] * sv;
    }
  }
  // the table is garbage now (gigabytes of it): a thread of its own gives it back while the results are being written
  auto *garbage = new std::tuple<DVec, DVec, std::vector<std::string>, std::vector<std::string>>(std::move(M), std::move(N), std::move(knames),
                                                                                              std::move(kmers));
  std::thread([garbage] { delete garbage; }).detach();
  return R;
}

}  // namespace kpop_host

// KPopTwistCA -- drop-in for the R stage of the reference's src/KPopTwist (the `Rscript --vanilla <(...)` block,
// src/KPopTwist:49-119): reads the count table KPopCountDB exported, runs the correspondence analysis on the GPU
// (kpop_ca: MFMA f64 GEMMs + Jacobi) and writes the same three tables R writes.
//
//   KPopTwistCA <TABLE> <NAMES> <PREFIX_OUT> <PREFIX_OUT_KMERS> <KMERS_KEEP> <KMERS_SAMPLE> <NORMALIZE>
//               <THRESHOLD_KMERS> <THREADS> <TEMPORARIES> <VERBOSE>
// -- the positional arguments of the Rscript call at src/KPopTwist:119, so the bash wrapper needs one word changed.
//   TABLE : header of spectrum names, then one row of counts per k-mer (KPopCountDB -t with
//           --table-output-row-names false, src/KPopTwist:38-40)
//   NAMES : one k-mer name per line (:41-44)
// Outputs (src/KPopTwist:98-116): <PREFIX_OUT>.KPopTwisted.txt (spectra x dims), .KPopInertia.txt,
// .KPopTwister.txt (dims x k-mers), and <PREFIX_OUT_KMERS>.KPopTwisted.txt (k-mers x dims) when requested.
// Differences: dimension signs are arbitrary (as in R); resampling with KMERS_SAMPLE < 1 uses SplitMix64, not
// R's RNG; numbers are printed %.15g.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/kpop_hip.h"
#include "kpop_text.h"

using namespace kpop_host;

namespace {

void check(int rc) {
  if (rc != 0) throw Error(std::string("libkpop_hip: ") + kpop_last_error());
}

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

std::string unquote(const std::string &s) { return strip_external_quotes_and_check(s); }

uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void write_named_table(const std::string &path, const char *corner, const std::vector<std::string> &cols,
                       const std::vector<std::string> &rows, const double *data, size_t ld, bool transposed) {
  // data(r, c) = transposed ? data[c * ld + r] : data[r * ld + c]
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) throw Error("cannot write '" + path + "'");
  std::vector<char> iobuf(1 << 22);
  setvbuf(f, iobuf.data(), _IOFBF, iobuf.size());
  fprintf(f, "\"%s\"", corner);
  for (const std::string &c : cols) fprintf(f, "\t\"%s\"", c.c_str());
  fputc('\n', f);
  for (size_t r = 0; r < rows.size(); ++r) {
    fprintf(f, "\"%s\"", rows[r].c_str());
    for (size_t c = 0; c < cols.size(); ++c) fprintf(f, "\t%.15g", transposed ? data[c * ld + r] : data[r * ld + c]);
    fputc('\n', f);
  }
  fclose(f);
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 4) {
    fprintf(stderr,
            "Usage: KPopTwistCA <TABLE> <NAMES> <PREFIX_OUT> [<PREFIX_OUT_KMERS> [<KMERS_KEEP> [<KMERS_SAMPLE> [<NORMALIZE>\n"
            "                   [<THRESHOLD_KMERS> [<THREADS> [<TEMPORARIES> [<VERBOSE>]]]]]]]]\n"
            "(the positional arguments of the Rscript call in KPop's src/KPopTwist)\n");
    return 1;
  }
  auto arg = [&](int i, const char *dflt) { return std::string(i < argc ? argv[i] : dflt); };
  const std::string table_path = arg(1, ""), names_path = arg(2, ""), out = arg(3, ""), out_kmers = arg(4, ""), keep_path = arg(5, "");
  const std::string frac_s = arg(6, "1."), norm_s = arg(7, "TRUE"), thr_s = arg(8, "0");
  const bool verbose = arg(11, "FALSE") == "TRUE" || arg(11, "") == "true";
  try {
    const double fraction = frac_s.empty() ? 1.0 : atof(frac_s.c_str());  // src/KPopTwist:60-62
    const bool normalize = norm_s == "TRUE" || norm_s == "true" || norm_s == "T";
    const double threshold = atof(thr_s.c_str());
    if (verbose) fprintf(stderr, "[2/16] Reading k-mers...\n");
    std::vector<std::string> kmers = read_lines(names_path);
    for (std::string &s : kmers) s = unquote(s);
    if (verbose) fprintf(stderr, "[3/16] Reading counts...\n");
    std::vector<std::string> spectra;
    std::vector<double> N;
    {
      FILE *f = fopen(table_path.c_str(), "rb");
      if (!f) throw Error("cannot open '" + table_path + "'");
      char *buf = nullptr;
      size_t cap = 0;
      ssize_t n = getline(&buf, &cap, f);
      if (n < 0) throw Error("'" + table_path + "' is empty");
      while (n > 0 && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) buf[--n] = 0;
      for (char *tok = buf, *e = buf + n; tok <= e;) {
        char *t = (char *)memchr(tok, '\t', (size_t)(e - tok));
        if (!t) t = e;
        spectra.push_back(unquote(std::string(tok, (size_t)(t - tok))));
        tok = t + 1;
      }
      const size_t J = spectra.size();
      uint64_t line = 1;
      while ((n = getline(&buf, &cap, f)) >= 0) {
        ++line;
        while (n > 0 && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) buf[--n] = 0;
        if (n == 0) continue;
        char *p = buf;
        for (size_t j = 0; j < J; ++j) {
          char *end = nullptr;
          N.push_back(strtod(p, &end));
          if (end == p || (j + 1 < J && *end != '\t') || (j + 1 == J && *end != 0))
            throw Error("table '" + table_path + "' line " + std::to_string(line) + ": expected " + std::to_string(J) + " numbers");
          p = end + 1;
        }
      }
      free(buf);
      fclose(f);
    }
    const size_t J = spectra.size();
    size_t I = J ? N.size() / J : 0;
    if (kmers.size() != I) throw Error("'" + names_path + "' has " + std::to_string(kmers.size()) + " k-mers, the table has " + std::to_string(I) + " rows");
    // [4/16] keep list (:76-82), [5/16] resampling (:84-86), [6/16] thresholding (:88-91)
    std::vector<size_t> sel(I);
    for (size_t i = 0; i < I; ++i) sel[i] = i;
    if (!keep_path.empty()) {
      std::unordered_map<std::string, size_t> idx;
      for (size_t i = 0; i < I; ++i) idx[kmers[i]] = i;
      sel.clear();
      for (const std::string &nm : read_lines(keep_path)) {
        auto it = idx.find(unquote(nm));
        if (it == idx.end()) throw Error("k-mer '" + nm + "' of the keep list is not in the table");
        sel.push_back(it->second);
      }
    }
    if (fraction < 1.0) {
      const size_t want = (size_t)((double)sel.size() * fraction);
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
      double mx = 0.0;
      for (size_t r = 0; r < sel.size(); ++r) {
        for (size_t j = 0; j < J; ++j) rsum[r] += N[sel[r] * J + j];
        mx = std::max(mx, rsum[r]);
      }
      std::vector<size_t> ns;
      for (size_t r = 0; r < sel.size(); ++r)
        if (rsum[r] >= mx * threshold) ns.push_back(sel[r]);
      sel.swap(ns);
    }
    std::vector<double> M(sel.size() * J);
    std::vector<std::string> knames(sel.size());
    for (size_t r = 0; r < sel.size(); ++r) {
      memcpy(&M[r * J], &N[sel[r] * J], J * sizeof(double));
      knames[r] = kmers[sel[r]];
    }
    N.clear();
    N.shrink_to_fit();
    I = sel.size();
    if (I < 2 || J < 2) throw Error("correspondence analysis needs at least 2 k-mers and 2 spectra");
    if (verbose) fprintf(stderr, "[8/16] Twisting counts (%zu k-mers x %zu spectra) on the GPU...\n", I, J);
    int dev = 0;
    if (const char *e = getenv("KPOP_DEVICE")) dev = atoi(e);
    check(kpop_init(dev));
    const size_t nd = std::min(I, J) - 1;
    std::vector<double> twisted(J * nd), inertia(nd), twister(nd * I);
    uint32_t nd_out = 0;
    check(kpop_ca(M.data(), I, (uint32_t)J, normalize ? 1 : 0, &nd_out, twisted.data(), inertia.data(), twister.data()));
    std::vector<std::string> dims(nd);
    for (size_t d = 0; d < nd; ++d) dims[d] = "Dim" + std::to_string(d + 1);
    if (verbose) fprintf(stderr, "[9/16] Writing twisted...\n");
    write_named_table(make_filename(out, "KPopTwisted", true), "rn", dims, spectra, twisted.data(), nd, false);
    if (!out_kmers.empty()) {  // principal row coordinates = standard ones x sv; sv_d^2 = inertia_d * sum(sv^2) is not
      // recoverable from the normalised inertia, so recompute sv from the class positions: for column coordinates
      // sum_j c_j G_jd^2 = sv_d^2.  With normalised columns c_j = 1/J.
      std::vector<double> colsum(J, 0.0);
      double total = 0.0;
      for (size_t r = 0; r < I; ++r)
        for (size_t j = 0; j < J; ++j) colsum[j] += M[r * J + j];
      for (size_t j = 0; j < J; ++j) total += normalize ? 1.0 : colsum[j];
      std::vector<double> koords(I * nd);
      for (size_t d = 0; d < nd; ++d) {
        double sv2 = 0.0;
        for (size_t j = 0; j < J; ++j) sv2 += (normalize ? 1.0 : colsum[j]) / total * twisted[j * nd + d] * twisted[j * nd + d];
        const double sv = sqrt(sv2);
        for (size_t r = 0; r < I; ++r) koords[r * nd + d] = twister[d * I + r] * sv;
      }
      write_named_table(make_filename(out_kmers, "KPopTwisted", true), "rn", dims, knames, koords.data(), nd, false);
    }
    if (verbose) fprintf(stderr, "[10/16] Writing inertia...\n");
    write_named_table(make_filename(out, "KPopInertia", true), "rn", dims, {"inertia"}, inertia.data(), nd, false);
    if (verbose) fprintf(stderr, "[13/16] Writing twister...\n");
    write_named_table(make_filename(out, "KPopTwister", true), "", knames, dims, twister.data(), I, false);
  } catch (const std::exception &e) {
    fprintf(stderr, "(KPopTwistCA): FATAL: %s\n", e.what());
    return 1;
  }
  return 0;
}

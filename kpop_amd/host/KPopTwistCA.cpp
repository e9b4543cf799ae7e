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
#include "ca_pipeline.h"
#include "kpop_text.h"

using namespace kpop_host;

namespace {

void check(int rc) {
  if (rc != 0) throw Error(std::string("libkpop_hip: ") + kpop_last_error());
}

std::string unquote(const std::string &s) { return strip_external_quotes_and_check(s); }

// fwrite(quote=TRUE, sep="\t") of a data.table with a row-name column (src/KPopTwist:100,108,116)
void write_r_table(const std::string &path, const char *corner, const Table &t) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) throw Error("cannot write '" + path + "'");
  std::vector<char> iobuf(1 << 22);
  setvbuf(f, iobuf.data(), _IOFBF, iobuf.size());
  fprintf(f, "\"%s\"", corner);
  for (const std::string &c : t.col_names) fprintf(f, "\t\"%s\"", c.c_str());
  fputc('\n', f);
  for (size_t r = 0; r < t.rows(); ++r) {
    fprintf(f, "\"%s\"", t.row_names[r].c_str());
    for (size_t c = 0; c < t.cols(); ++c) fprintf(f, "\t%.15g", t.data[r * t.cols() + c]);
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
    DVec N;
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
    const size_t I = J ? N.size() / J : 0;
    if (kmers.size() != I) throw Error("'" + names_path + "' has " + std::to_string(kmers.size()) + " k-mers, the table has " + std::to_string(I) + " rows");
    int dev = 0;
    if (const char *e = getenv("KPOP_DEVICE")) dev = atoi(e);
    check(kpop_init(dev));
    CaParams P;
    P.keep_path = keep_path;
    P.fraction = fraction;
    P.threshold = threshold;
    P.normalize = normalize;
    P.want_kmer_coords = !out_kmers.empty();
    P.verbose = verbose;
    const CaResult R = run_ca(std::move(kmers), spectra, std::move(N), P);
    if (verbose) fprintf(stderr, "[9/16] Writing twisted...\n");
    write_r_table(make_filename(out, "KPopTwisted", true), "rn", R.twisted);
    if (!out_kmers.empty()) write_r_table(make_filename(out_kmers, "KPopTwisted", true), "rn", R.kmer_coords);
    if (verbose) fprintf(stderr, "[10/16] Writing inertia...\n");
    write_r_table(make_filename(out, "KPopInertia", true), "rn", R.inertia);
    if (verbose) fprintf(stderr, "[13/16] Writing twister...\n");
    write_r_table(make_filename(out, "KPopTwister", true), "", R.twister);
  } catch (const std::exception &e) {
    fprintf(stderr, "(KPopTwistCA): FATAL: %s\n", e.what());
    return 1;
  }
  return 0;
}

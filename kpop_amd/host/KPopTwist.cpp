// KPopTwist -- drop-in for the reference's KPopTwist (the bash script src/KPopTwist around bin/KPopTwist_.ml):
// loads a '.KPopCounter' database, transforms and normalises its counts, runs the correspondence analysis and saves
// '<prefix>.KPopTwister' (twister + inertia) and '<prefix>.KPopTwisted' (the twisted training spectra).
//
// Same options as bin/KPopTwist_.ml:52-135.  What the reference does by piping text tables through KPopCountDB, R
// (data.table + ca) and three KPopTwistDB conversions (src/KPopTwist:36-131) happens in one process: the table
// transformation in kpop_counter_stats/_transform, the analysis in kpop_ca, the binaries through ocaml_marshal.
//
// Differences from the reference, on purpose: dimension signs are arbitrary (as in R); -s resamples with SplitMix64,
// not R's RNG; --keep-temporaries keeps the three '.txt' tables next to the binaries (no temporary directory is ever
// made); k-mers whose transformed counts are all zero are dropped together with their names (the reference's two
// exports can fall out of step there, src/KPopTwist:38-44).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/kpop_hip.h"
#include "ca_pipeline.h"
#include "counter_db.h"
#include "twist_args.h"

using namespace kpop_host;

namespace {

const char *kVersion = "27-hip";

void check(int rc) {
  if (rc != 0) throw Error(std::string("libkpop_hip: ") + kpop_last_error());
}

}  // namespace

int main(int argc, char **argv) {
  TwistArgs A = parse_twist_args(argc, argv, "KPopTwist", kVersion);
  const std::string &input = A.input, &output = A.output, &output_kmers = A.output_kmers;
  const Transform &transform = A.transform;
  CaParams &P = A.ca;
  const bool temporaries = A.temporaries;
  try {
    const int which = transform.code();
    int dev = 0;
    if (const char *e = getenv("KPOP_DEVICE")) dev = atoi(e);
    stage_mark("KPopTwist", "start");
    check(kpop_init(dev));
    stage_mark("KPopTwist", "HIP bring-up");
    if (P.verbose) fprintf(stderr, "[1/16] Exporting table...\n");
    CounterDB db = CounterDB::of_binary(input);
    stage_mark("KPopTwist", "database read");
    const size_t nr = db.n_rows(), nc = db.n_cols();
    std::vector<const int32_t *> cols = db.columns();
    // the export of src/KPopTwist:38-40: transformed counts, k-mers with an all-zero row left out
    std::vector<double> col_stats(4 * std::max<size_t>(1, nc)), row_stats(4 * std::max<size_t>(1, nr));
    check(kpop_counter_stats(cols.data(), (uint32_t)nc, nr, transform.threshold, transform.power, col_stats.data(), row_stats.data()));
    DVec table(std::max<size_t>(1, nr * nc));
    check(kpop_counter_transform(cols.data(), (uint32_t)nc, nr, which, transform.threshold, transform.power, col_stats.data(), 1,
                                 table.data()));
    stage_mark("KPopTwist", "statistics, transformed table");
    // rows with counts, in order: their numbers first, then names and values moved by the host threads
    std::vector<size_t> kept;
    kept.reserve(nr);
    for (size_t r = 0; r < nr; ++r)
      if (row_stats[4 * r + 2] > 0.) kept.push_back(r);
    std::vector<std::string> kmers;
    DVec counts;
    if (kept.size() == nr) {  // every k-mer occurs somewhere (the usual case): the table as it stands
      kmers.swap(db.core.row_names);
      counts.swap(table);
    } else {
      kmers.resize(kept.size());
      counts.resize(kept.size() * nc);
      parallel_for(kept.size(), 4096, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
          kmers[i] = std::move(db.core.row_names[kept[i]]);
          memcpy(counts.data() + i * nc, table.data() + kept[i] * nc, nc * sizeof(double));
        }
      });
    }
    DVec().swap(table);
    P.want_kmer_coords = !output_kmers.empty();
    const std::vector<std::string> spectra = db.core.col_names;
    {  // the database is not needed again: a thread of its own gives its gigabytes back
      auto *garbage = new CounterDB(std::move(db));
      std::thread([garbage] { delete garbage; }).detach();
    }
    stage_mark("KPopTwist", "rows without counts dropped");
    const CaResult R = run_ca(std::move(kmers), spectra, std::move(counts), P);
    stage_mark("KPopTwist", "correspondence analysis");
    if (P.verbose) fprintf(stderr, "[14/16] Encoding twisted...\n");
    write_binary_matrix(make_filename(output, "KPopTwisted", false), "KPopTwisted", R.twisted);
    if (!output_kmers.empty()) write_binary_matrix(make_filename(output_kmers, "KPopTwisted", false), "KPopTwisted", R.kmer_coords);
    if (P.verbose) fprintf(stderr, "[15/16] Encoding twister...\n");
    write_binary_twister(make_filename(output, "KPopTwister", false), R.twister, R.inertia);
    if (temporaries) {
      write_table(make_filename(output, "KPopTwisted", true), R.twisted, 15);
      write_table(make_filename(output, "KPopInertia", true), R.inertia, 15);
      write_table(make_filename(output, "KPopTwister", true), R.twister, 15);
    }
    stage_mark("KPopTwist", "outputs written");
    if (P.verbose) fprintf(stderr, "All done.\n");
    fflush(stdout);
    fflush(stderr);
    _exit(0);  // everything is written and closed: no need to take gigabytes of tables apart first
  } catch (const std::exception &e) {
    fprintf(stderr, "(KPopTwist): FATAL: %s\n", e.what());
    return 1;
  }
  return 0;
}

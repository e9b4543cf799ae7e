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
    // The export of src/KPopTwist:38-40 -- transformed counts, k-mers with an all-zero row left out -- without the table ever
    // leaving the device: the spectra go up once, statistics and transformation run there, the rows that stay are picked
    // there, and kpop_dev_ca standardises the table where it stands.  (Through the host entry points the spectra went up
    // twice and the table crossed PCIe three times.)
    auto chk = [](int rc) { check(rc); };
    struct Dev {
      void *p = nullptr;
      ~Dev() {
        if (p) (void)kpop_dev_free(p);
      }
      void alloc(uint64_t bytes) { check(kpop_dev_malloc(&p, bytes ? bytes : 8)); }
      void release() {
        if (p) (void)kpop_dev_free(p);
        p = nullptr;
      }
    };
    const uint64_t ld = kpop_dev_counter_ld(nr);
    Dev storage, ws, d_col_stats, d_row_stats, table;
    storage.alloc((uint64_t)std::max<size_t>(1, nc) * ld * 4);
    for (size_t c = 0; c < nc; ++c)
      if (nr) chk(kpop_memcpy_h2d(static_cast<int32_t *>(storage.p) + c * ld, cols[c], nr * 4));
    ws.alloc(kpop_dev_counter_workspace_bytes((uint32_t)nc, nr));
    d_col_stats.alloc((uint64_t)std::max<size_t>(1, nc) * 32);
    d_row_stats.alloc((uint64_t)std::max<size_t>(1, nr) * 32);
    if (nc) chk(kpop_dev_counter_stats(static_cast<const int32_t *>(storage.p), ld, (uint32_t)nc, nr, transform.threshold, transform.power, ws.p,
                                       static_cast<double *>(d_col_stats.p), static_cast<double *>(d_row_stats.p), nullptr));
    std::vector<double> row_stats(4 * std::max<size_t>(1, nr));
    if (nr && nc) chk(kpop_memcpy_d2h(row_stats.data(), d_row_stats.p, nr * 32));
    table.alloc((uint64_t)std::max<size_t>(1, nr * nc) * 8);
    if (nr && nc)
      chk(kpop_dev_counter_transform(static_cast<const int32_t *>(storage.p), ld, (uint32_t)nc, nr, which, transform.threshold, transform.power,
                                     static_cast<const double *>(d_col_stats.p), 1, static_cast<double *>(table.p), nullptr));
    chk(kpop_synchronize(nullptr));
    storage.release();
    ws.release();
    d_col_stats.release();
    d_row_stats.release();
    stage_mark("KPopTwist", "statistics, transformed table");
    // rows with counts, in order
    std::vector<uint64_t> kept;
    kept.reserve(nr);
    for (size_t r = 0; r < nr; ++r)
      if (row_stats[4 * r + 2] > 0.) kept.push_back(r);
    std::vector<std::string> kmers;
    if (kept.size() == nr) kmers.swap(db.core.row_names);  // every k-mer occurs somewhere (the usual case)
    else {
      kmers.resize(kept.size());
      for (size_t i = 0; i < kept.size(); ++i) kmers[i] = std::move(db.core.row_names[kept[i]]);
      Dev d_rows, picked;
      d_rows.alloc(kept.size() * 8);
      chk(kpop_memcpy_h2d(d_rows.p, kept.data(), kept.size() * 8));
      picked.alloc((uint64_t)std::max<size_t>(1, kept.size() * nc) * 8);
      chk(kpop_dev_table_gather_rows(static_cast<const double *>(table.p), (uint32_t)nc, static_cast<const uint64_t *>(d_rows.p), kept.size(),
                                     static_cast<double *>(picked.p), nullptr));
      chk(kpop_synchronize(nullptr));
      table.release();
      table.p = picked.p;
      picked.p = nullptr;
    }
    P.want_kmer_coords = !output_kmers.empty();
    const std::vector<std::string> spectra = db.core.col_names;
    stage_mark("KPopTwist", "rows without counts dropped");
    void *d_table = table.p;
    table.p = nullptr;  // (run_ca_device takes it over)
    const CaResult R = run_ca_device(std::move(kmers), spectra, d_table, P);
    {  // The database is not needed again: a thread of its own gives its gigabytes back while the results are written.  (Not
       // earlier: unmapping host memory holds up the device allocations and copies of the analysis -- 0.63 s where it takes
       // 0.3 s alone.)
      auto *garbage = new CounterDB(std::move(db));
      std::thread([garbage] { delete garbage; }).detach();
    }
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

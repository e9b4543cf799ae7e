// ca_pipeline.h -- the R stage of the reference's src/KPopTwist (:70-116) over an in-memory count table: k-mer keep
// list, resampling, k-mer thresholding, column normalisation and the correspondence analysis itself (kpop_ca on the
// GPU).  Shared by KPopTwistCA (text tables in, text tables out) and KPopTwist (.KPopCounter in, binaries out).
#pragma once
#include <string>
#include <vector>

#include "kpop_text.h"

namespace kpop_host {

struct CaParams {
  std::string keep_path;   // -k: one k-mer per line (src/KPopTwist:76-82)
  double fraction = 1.;    // -s (:84-86); SplitMix64, not R's RNG
  double threshold = 0.;   // --kmers-threshold (:88-91)
  bool normalize = true;   // --counts-normalize (:93-94)
  bool want_kmer_coords = false;  // -K (:101-103)
  bool verbose = false;
};

struct CaResult {
  Table twisted;      // spectra x dims              (:98-100)
  Table inertia;      // "inertia" x dims            (:105-108)
  Table twister;      // dims x k-mers               (:110-116)
  Table kmer_coords;  // k-mers x dims, on request   (:101-103)
};

// counts: kmers.size() x spectra.size() row-major (transformed counts, as KPopCountDB -t writes them)
CaResult run_ca(std::vector<std::string> kmers, const std::vector<std::string> &spectra, DVec counts, const CaParams &P);
// The same on a table that is on the device already (d_table: kmers.size() x spectra.size() doubles, row-major, from
// kpop_dev_malloc; this call takes it over and frees it): the table is selected from, standardised and analysed where it is,
// and only the results cross to the host.  The threshold on the row sums uses the device's sums (a fixed tree per row), which
// can differ from the host's running sums in the last bits.
CaResult run_ca_device(std::vector<std::string> kmers, const std::vector<std::string> &spectra, void *d_table, const CaParams &P);

std::vector<std::string> read_lines(const std::string &path);

}  // namespace kpop_host

// splits.h -- the container phylogenetic splits live in, and the "centroids" algorithm.
//
// The reference keeps splits in BiOCamLib's Trees.Splits (absent from the checkout): `create names`, `add_split t set
// weight`, `to_file ~precision t prefix` (lib/Matrix.ml:593-611, bin/KPopTwistDB.ml:534-535).  DECLARED here, like the
// k-mer encoding: splits are kept in the order they were added, nothing is merged; '<prefix>.PhyloSplits.txt' is
//     ""  "<name 0>"  "<name 1>" ...                      the leaves (the embeddings' row names), tab-separated
//     <weight %.{precision}g>  "<member>"  "<member>" ...  one line per split: its weight, then the names on its side
// The binary '.PhyloSplits' (Marshal of Trees.Splits.t) is not provided.
//
// "gaps" (the default) runs on the device (kpop_splits_gaps).  "centroids" (lib/Matrix.ml:361-521,601-612) is a
// randomised search -- recursive bipartition by simulated annealing, one element moved per step -- whose draws come from
// OCaml's Random in the reference; its state differs between OCaml versions, so no run is reproducible across builds
// even upstream.  Here the draws come from SplitMix64 with a fixed seed (bool = low bit, int n = value mod n,
// float 1. = top 53 bits / 2^53), the arithmetic follows the reference line by line, and oracle/pyref.py holds the same
// restatement: the two agree bit for bit.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "kpop_text.h"

namespace kpop_host {

struct Splits {
  struct Split {
    std::vector<uint32_t> members;  // ascending row numbers
    double weight = 0.;
  };
  std::vector<std::string> names;
  std::vector<Split> splits;
};

void write_splits(const std::string &path, const Splits &s, int precision);
Splits read_splits(const std::string &path);

// Matrix.get_splits ... Centroids (lib/Matrix.ml:601-612) on embeddings (rows x n_dims, row-major)
Splits splits_centroids(const std::vector<std::string> &row_names, const double *emb, size_t n_dims, bool verbose, uint64_t seed = 0x4B506F70);

}  // namespace kpop_host

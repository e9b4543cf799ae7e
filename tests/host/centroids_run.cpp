// Test harness (not product): reads the table argv[1] (rows = leaves, columns = dimensions), runs the "centroids" splits
// algorithm of the drop-in (kpop_amd/host/splits.cpp) and writes the declared '.PhyloSplits.txt' text to stdout.
#include <stdio.h>

#include "../../kpop_amd/host/splits.h"

using namespace kpop_host;

int main(int argc, char **argv) {
  if (argc < 2) return 64;
  try {
    const Table t = read_table(argv[1]);
    const Splits s = splits_centroids(t.row_names, t.data.data(), t.cols(), false);
    write_splits("/dev/stdout", s, 10);
    // and back in: the reader of the same format
    return 0;
  } catch (const std::exception &e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
  }
}

// Test harness (not product): the block parser of the drop-in's text-spectra path (SpectraTextStream + parse_spectra_block) with
// the twister's columns known to it (KmerLookup, kpop_text.h).  argv: spectra file, "hex" | "opaque", file of column names (one a
// line), block size.  Prints "L <label>" per spectrum and "<number> <value %.17g>" per kept line ("-" for a name that is no
// column), or "ERROR <message>"; tests/test_host_parsers.py compares that with what lib/Twister.ml:91-169 says, restated in Python.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include <algorithm>
#include <fstream>
#include <string>

#include "../../kpop_amd/host/kpop_text.h"

using namespace kpop_host;

int main(int argc, char **argv) {
  if (argc < 5) return 64;
  const bool opaque = std::string(argv[2]) == "opaque";
  std::vector<std::string> names;
  {
    std::ifstream f(argv[3]);
    std::string l;
    while (std::getline(f, l)) names.push_back(l);
  }
  std::vector<uint64_t> columns;
  std::unordered_map<std::string_view, uint64_t> index;
  size_t name_len = 0;
  if (opaque) {
    for (size_t c = 0; c < names.size(); ++c) index[std::string_view(names[c])] = c;
  } else {
    name_len = names.empty() ? 0 : names[0].size();
    for (const std::string &n : names) {
      uint64_t h = 0;
      if (!hex_to_hash(n, &h)) return 66;
      columns.push_back(h);
    }
    std::sort(columns.begin(), columns.end());
  }
  KmerLookup lk;
  lk.opaque = opaque;
  lk.columns = opaque ? nullptr : &columns;
  lk.index = opaque ? &index : nullptr;
  const uint64_t absent = ~0ull >> 1;
  try {
    const int fd = open(argv[1], O_RDONLY);
    if (fd < 0) return 65;
    SpectraTextStream ts(fd, nullptr, 0, (size_t)atoll(argv[4]));
    TextBlock b;
    bool first = true;
    uint64_t lines = 0, n = 0;
    std::string out;
    while (ts.next(b)) {
      HashedSpectra one;
      parse_spectra_block(b.data(), b.size(), name_len, absent, first, lines, one, &n, 0, nullptr, &lk);
      first = false;
      lines += n;
      for (size_t s = 0; s < one.labels.size(); ++s) {
        out += "L " + one.labels[s] + "\n";
        for (uint64_t i = one.offsets[s]; i < one.offsets[s + 1]; ++i) {
          char buf[64];
          const bool col = one.hash[i] != absent && (opaque || std::binary_search(columns.begin(), columns.end(), one.hash[i]));
          if (col) snprintf(buf, sizeof buf, "%llu %.17g\n", (unsigned long long)one.hash[i], one.values[i]);
          else snprintf(buf, sizeof buf, "-\n");
          out += buf;
        }
      }
    }
    close(fd);
    fputs(out.c_str(), stdout);
  } catch (const std::exception &e) {
    printf("ERROR %s\n", e.what());
  }
  return 0;
}

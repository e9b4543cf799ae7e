// Test harness (not product): parses argv[1] with the sequential spectra parser (read_spectra_file) and with the
// threaded one the GPU path uses (read_spectra_hashed); they must agree on success or failure, on the message, and on
// labels, offsets, values and name -> hash conversion.  With argv[3] = a block size, a third way: the block stream the
// drop-in reads text spectra through (SpectraTextStream + parse_spectra_block, blocks appended in order).
// Exit 0 = agree, 2 = messages differ, 3 = content differs.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include <string>

#include "../../kpop_amd/host/kpop_text.h"

using namespace kpop_host;

int main(int argc, char **argv) {
  if (argc < 3) return 64;
  const size_t name_len = (size_t)atoi(argv[2]);
  std::string e1, e2;
  HashedSpectra hs;
  Spectra sp;
  try {
    read_spectra_hashed(argv[1], name_len, ~0ull >> 1, hs);
  } catch (const std::exception &e) {
    e1 = e.what();
  }
  try {
    read_spectra_file(argv[1], sp);
  } catch (const std::exception &e) {
    e2 = e.what();
  }
  if (argc > 3) {  // the block stream must say what the whole-file parser says
    std::string e3;
    HashedSpectra all;
    try {
      const int fd = open(argv[1], O_RDONLY);
      if (fd < 0) return 65;
      char head[8];
      const ssize_t got = read(fd, head, 8);  // the caller has looked at the first bytes, as KPopTwistDB does
      SpectraTextStream ts(fd, head, got > 0 ? (size_t)got : 0, (size_t)atoll(argv[3]));
      TextBlock b;
      bool first = true;
      uint64_t lines = 0, n = 0;
      while (ts.next(b)) {
        HashedSpectra one;
        parse_spectra_block(b.data(), b.size(), name_len, ~0ull >> 1, first, lines, one, &n);
        if (!first && !one.labels.empty() && b[0] != '\t') {
          printf("A LATER BLOCK DOES NOT BEGIN WITH A HEADER\n");
          return 3;
        }
        first = false;
        lines += n;
        const uint64_t at = all.hash.size();
        all.labels.insert(all.labels.end(), one.labels.begin(), one.labels.end());
        for (size_t i = 1; i < one.offsets.size(); ++i) all.offsets.push_back(at + one.offsets[i]);
        all.hash.insert(all.hash.end(), one.hash.begin(), one.hash.end());
        all.values.insert(all.values.end(), one.values.begin(), one.values.end());
      }
      close(fd);
    } catch (const std::exception &e) {
      e3 = e.what();
    }
    if (e3 != e1) {
      printf("BLOCK STREAM MESSAGE DIFFERS: [%s] vs [%s]\n", e3.c_str(), e1.c_str());
      return 2;
    }
    if (e1.empty()) {
      bool same = all.labels == hs.labels && all.offsets == hs.offsets && all.hash == hs.hash && all.values.size() == hs.values.size();
      for (size_t i = 0; same && i < hs.values.size(); ++i)
        same = all.values[i] == hs.values[i] || (all.values[i] != all.values[i] && hs.values[i] != hs.values[i]);
      if (!same) {
        printf("BLOCK STREAM CONTENT DIFFERS\n");
        return 3;
      }
    }
  }
  if (e1 != e2) {
    printf("MESSAGES DIFFER: [%s] vs [%s]\n", e1.c_str(), e2.c_str());
    return 2;
  }
  if (!e1.empty()) return 0;
  bool ok = hs.labels == sp.labels && hs.offsets == sp.offsets && hs.values.size() == sp.values.size();
  for (size_t i = 0; ok && i < sp.values.size(); ++i) {
    ok = (hs.values[i] == sp.values[i]) || (hs.values[i] != hs.values[i] && sp.values[i] != sp.values[i]);
    uint64_t h = 0;
    const bool hex = sp.names[i].size() == name_len && hex_to_hash(sp.names[i], &h);
    ok = ok && hs.hash[i] == (hex ? h : (~0ull >> 1));
  }
  if (!ok) printf("CONTENT DIFFERS\n");
  return ok ? 0 : 3;
}

// Test harness (not product): parses argv[1] with the sequential spectra parser (read_spectra_file) and with the
// threaded one the GPU path uses (read_spectra_hashed); they must agree on success or failure, on the message, and on
// labels, offsets, values and name -> hash conversion.  Exit 0 = agree, 2 = messages differ, 3 = content differs.
#include <stdio.h>

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

// Test harness (not product): order_rows_by_label (kpop_text.h) against the reference's own bookkeeping restated with a
// std::map (lib/Twister.ml:78-82,189-204: existing rows overwrite each other, a new label that is already present raises
// Duplicate_label for the first such row, result in bytewise label order).  argv[1] = seed, argv[2] = rows, argv[3] = alphabet.
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <string>

#include "../../kpop_amd/host/kpop_text.h"

using namespace kpop_host;

int main(int argc, char **argv) {
  if (argc < 4) return 64;
  srand((unsigned)atoi(argv[1]));
  const size_t n = (size_t)atoi(argv[2]);
  const int alphabet = atoi(argv[3]);
  const size_t n_old = n ? (size_t)rand() % (n / 3 + 1) : 0;
  std::vector<std::string> labels(n);
  for (size_t i = 0; i < n; ++i) {
    const int len = 1 + rand() % 4;
    for (int c = 0; c < len; ++c) labels[i].push_back((char)(rand() % 3 == 0 ? 128 + rand() % alphabet : 'a' + rand() % alphabet));
  }
  std::string want_err, got_err;
  std::map<std::string, uint32_t> ref;
  for (size_t i = 0; i < n_old; ++i) ref[labels[i]] = (uint32_t)i;  // StringMap.add: the last one stays
  for (size_t i = n_old; i < n && want_err.empty(); ++i) {
    if (ref.count(labels[i])) want_err = "Duplicate_label(\"" + labels[i] + "\")";
    else ref[labels[i]] = (uint32_t)i;
  }
  std::vector<uint32_t> got;
  try {
    got = order_rows_by_label(labels, n_old);
  } catch (const std::exception &e) {
    got_err = e.what();
  }
  if (want_err != got_err) {
    printf("ERRORS DIFFER: [%s] vs [%s]\n", want_err.c_str(), got_err.c_str());
    return 2;
  }
  if (!want_err.empty()) return 0;
  std::vector<uint32_t> want;
  for (auto &kv : ref) want.push_back(kv.second);  // std::map<std::string>: bytewise (unsigned char) order, as OCaml's compare
  if (want != got) {
    printf("ORDER DIFFERS (%zu vs %zu rows)\n", want.size(), got.size());
    return 3;
  }
  return 0;
}

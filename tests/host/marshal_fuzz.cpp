// marshal_fuzz.cpp -- a mutation loop over the readers of the binary registers (kpop_amd/host/ocaml_marshal.cpp): valid
// '.KPopTwister' / '.KPopTwisted' / '.KPopCounter' files are written, bytes of them are flipped, truncated, duplicated or
// overwritten with the wire format's own interesting values (length prefixes, shared-reference codes, the custom-block
// identifier), and every reader is run on the result.  A reader may return or throw std::exception; anything else -- a crash, a
// sanitizer report (this file is built with -fsanitize=address,undefined by `make -C kpop_amd/host asan`), an allocation of
// gigabytes from a length in the file -- is a finding.  Test infrastructure only.
//   marshal_fuzz <scratch dir> <seconds> [seed]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <exception>
#include <random>
#include <string>
#include <vector>

#include "../../kpop_amd/host/ocaml_marshal.h"

using namespace kpop_host;

static std::vector<unsigned char> slurp(const std::string &p) {
  std::vector<unsigned char> b;
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) return b;
  unsigned char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
  fclose(f);
  return b;
}
static void spit(const std::string &p, const std::vector<unsigned char> &b) {
  FILE *f = fopen(p.c_str(), "wb");
  if (!f) abort();
  if (!b.empty()) fwrite(b.data(), 1, b.size(), f);
  fclose(f);
}

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: marshal_fuzz <scratch dir> <seconds> [seed]\n");
    return 2;
  }
  const std::string dir = argv[1];
  const double seconds = atof(argv[2]);
  std::mt19937_64 rng(argc > 3 ? strtoull(argv[3], nullptr, 10) : 12345);
  // the valid files
  Table tw, in, td;
  tw.col_names = {"000", "001", "00a", "0ff", "3ff"};
  tw.row_names = {"Dim1", "Dim2", "Dim3"};
  tw.data.resize(15);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 5; ++c) tw.data[r * 5 + c] = 0.25 * r - 0.125 * c;
  in.col_names = tw.row_names;
  in.row_names = {"inertia"};
  in.data.resize(3);
  in.data[0] = 0.5, in.data[1] = 0.3, in.data[2] = 0.2;
  td.col_names = tw.row_names;
  td.row_names = {"a", "b \001 c", std::string(300, 'x'), ""};
  td.data.resize(12);
  for (int i = 0; i < 12; ++i) td.data[i] = 1.5 + i;
  CounterCore db;
  db.col_names = {"s1", "s2", "s3"};
  db.row_names = {"000", "001", "010", "3ff"};
  db.meta_names = {"class", "note"};
  db.meta = {{"A", "x"}, {"B", ""}, {"A", std::string(40, 'y')}};
  db.storage = {{1, 0, 3, 7}, {0, 0, 0, 1}, {2147483647, -1, 5, 0}};
  const std::string ptw = dir + "/v.KPopTwister", ptd = dir + "/v.KPopTwisted", pdb = dir + "/v.KPopCounter", pm = dir + "/m.bin";
  write_binary_twister(ptw, tw, in);
  write_binary_matrix(ptd, "KPopTwisted", td);
  write_binary_counter(pdb, db);
  const std::vector<unsigned char> valid[3] = {slurp(ptw), slurp(ptd), slurp(pdb)};
  static const unsigned char kInteresting[] = {0x00, 0x01, 0x04, 0x05, 0x06, 0x07, 0x08, 0x09, 0x0A, 0x0D, 0x0E, 0x0F, 0x10, 0x12, 0x13, 0x15,
                                               0x18, 0x19, 0x20, 0x3F, 0x40, 0x7F, 0x80, 0x8F, 0x90, 0xBE, 0xBF, 0xFE, 0xFF};
  unsigned long iters = 0, threw = 0, ok = 0;
  const time_t t_end = time(nullptr) + (time_t)seconds;
  while (time(nullptr) < t_end) {
    for (int rep = 0; rep < 200; ++rep, ++iters) {
      const int which = (int)(rng() % 3);
      std::vector<unsigned char> b = valid[which];
      const int n_mut = 1 + (int)(rng() % 4);
      for (int m = 0; m < n_mut && !b.empty(); ++m) {
        const size_t at = rng() % b.size();
        switch (rng() % 7) {
          case 0: b[at] ^= (unsigned char)(1u << (rng() % 8)); break;
          case 1: b[at] = kInteresting[rng() % sizeof kInteresting]; break;
          case 2: b.resize(at); break;  // truncate
          case 3: {  // a big-endian length of 2, 4 or 8 bytes overwritten with something large
            const int w = 1 << (1 + rng() % 3);
            for (int i = 0; i < w && at + i < b.size(); ++i) b[at + i] = (i == 0 && rng() % 2) ? 0x7F : (unsigned char)rng();
            break;
          }
          case 4: {  // a stretch duplicated
            const size_t len = 1 + rng() % 24;
            std::vector<unsigned char> piece(b.begin() + at, b.begin() + std::min(b.size(), at + len));
            b.insert(b.begin() + at, piece.begin(), piece.end());
            break;
          }
          case 5: b.erase(b.begin() + at, b.begin() + std::min(b.size(), at + 1 + rng() % 8)); break;
          default: b[at] = (unsigned char)rng(); break;
        }
      }
      spit(pm, b);
      for (int reader = 0; reader < 4; ++reader) {
        try {
          if (reader == 0) {
            Table a, c;
            read_binary_twister(pm, &a, &c);
          } else if (reader == 1) {
            Table c;
            read_binary_twister_inertia(pm, &c);
          } else if (reader == 2) {
            (void)read_binary_matrix(pm, "KPopTwisted");
          } else {
            (void)read_binary_counter(pm);
          }
          ++ok;
        } catch (const std::exception &) {
          ++threw;
        }
      }
    }
  }
  printf("marshal_fuzz: %lu mutated files x 4 readers: %lu read, %lu refused with an exception, no crash, no sanitizer report\n", iters, ok, threw);
  return 0;
}

// kpop_synth -- deterministic synthetic sequence files for the benches and tests (tooling, not a reference tool).
//
//   kpop_synth genomes --n C --len L [--seed S]                       C random genomes, FASTA wrapped at 70 columns
//   kpop_synth reads   --from F.fa --n N --len L [--mutate p] [--seed S]   reads sampled from the sequences of F (either
//                                                                     strand), each base substituted with probability p
//   kpop_synth mutants --from F.fa --n N [--mutate p] [--seed S]      N copies of the FIRST sequence of F with substitutions
//                                                                     (near-identical assemblies, BASELINE config 3)
// SplitMix64 throughout; read i depends on (seed, i) only, so any prefix of a larger file is the smaller file.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "fast_seq.h"

using namespace kpop_host;

static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() { return mix64(s += 0x9E3779B97F4A7C15ull); }
  double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

static const char kBases[] = "ACGT";

static void put_wrapped(FILE *f, const std::string &s, size_t width) {
  for (size_t i = 0; i < s.size(); i += width) {
    fwrite(s.data() + i, 1, std::min(width, s.size() - i), f);
    fputc('\n', f);
  }
}

// substitutions at geometric gaps: each base changes with probability p, to one of the three other bases
static void mutate(std::string &s, double p, Rng &rng) {
  if (p <= 0.) return;
  size_t i = 0;
  const double lg = log(1. - p);
  for (;;) {
    const double u = rng.unit();
    const double gap = floor(log(1. - u) / lg);
    if (gap > 1e15) return;
    i += (size_t)gap;
    if (i >= s.size()) return;
    const char *q = strchr(kBases, s[i]);
    const int cur = q ? (int)(q - kBases) : 0;
    s[i] = kBases[(cur + 1 + (int)(rng.next() % 3)) & 3];
    ++i;
  }
}

static std::vector<std::string> load_fasta(const std::string &path, std::vector<std::string> *names) {
  FastSeqReader rd(path, SeqFormat::FASTA);
  FlatBatch b;
  std::vector<std::string> out;
  while (rd.next(b)) {
    size_t ob = 0, ot = 0;
    for (size_t i = 0; i < b.size(); ++i) {
      out.emplace_back((const char *)b.bases.data() + ob, b.lens[i]);
      if (names) names->emplace_back(b.tags.data() + ot, b.tag_lens[i]);
      ob += b.lens[i];
      ot += b.tag_lens[i];
    }
  }
  return out;
}

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: kpop_synth genomes|reads|mutants [--n N] [--len L] [--from F.fa] [--mutate p] [--seed S]\n");
    return 1;
  }
  const std::string mode = argv[1];
  uint64_t n = 1, len = 1000, seed = 0x4B506F70;
  double p = 0.;
  std::string from;
  for (int i = 2; i + 1 < argc; i += 2) {
    const std::string a = argv[i];
    if (a == "--n") n = strtoull(argv[i + 1], nullptr, 10);
    else if (a == "--len") len = strtoull(argv[i + 1], nullptr, 10);
    else if (a == "--seed") seed = strtoull(argv[i + 1], nullptr, 0);
    else if (a == "--mutate") p = atof(argv[i + 1]);
    else if (a == "--from") from = argv[i + 1];
    else {
      fprintf(stderr, "kpop_synth: unknown option %s\n", a.c_str());
      return 1;
    }
  }
  std::vector<char> iobuf(1 << 22);
  setvbuf(stdout, iobuf.data(), _IOFBF, iobuf.size());
  try {
    if (mode == "genomes") {
      for (uint64_t g = 0; g < n; ++g) {
        Rng rng(mix64(seed ^ (g * 0x632BE59BD9B4E019ull)));
        std::string s(len, 'A');
        for (uint64_t i = 0; i < len; i += 32) {
          uint64_t r = rng.next();
          for (uint64_t j = i; j < std::min(len, i + 32); ++j, r >>= 2) s[j] = kBases[r & 3];
        }
        printf(">g%llu\n", (unsigned long long)g);
        put_wrapped(stdout, s, 70);
      }
    } else if (mode == "reads") {
      if (from.empty()) throw Error("reads: --from is needed");
      const std::vector<std::string> src = load_fasta(from, nullptr);
      std::vector<size_t> ok;
      for (size_t g = 0; g < src.size(); ++g)
        if (src[g].size() >= len) ok.push_back(g);
      if (ok.empty()) throw Error("reads: no sequence of '" + from + "' is as long as a read");
      std::string r;
      char name[32];
      for (uint64_t i = 0; i < n; ++i) {
        Rng rng(mix64(seed ^ (i * 0x632BE59BD9B4E019ull)));
        const std::string &g = src[ok[rng.next() % ok.size()]];
        const size_t start = (size_t)(rng.next() % (g.size() - len + 1));
        r.assign(g, start, len);
        if (rng.next() & 1) {  // the other strand
          std::string rc(len, 'N');
          for (size_t j = 0; j < len; ++j) {
            const char c = r[len - 1 - j];
            rc[j] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
          }
          r.swap(rc);
        }
        mutate(r, p, rng);
        const int nl = snprintf(name, sizeof(name), ">r%llu\n", (unsigned long long)i);
        fwrite(name, 1, (size_t)nl, stdout);
        fwrite(r.data(), 1, r.size(), stdout);
        fputc('\n', stdout);
      }
    } else if (mode == "mutants") {
      if (from.empty()) throw Error("mutants: --from is needed");
      const std::vector<std::string> src = load_fasta(from, nullptr);
      if (src.empty()) throw Error("mutants: '" + from + "' holds no sequence");
      for (uint64_t i = 0; i < n; ++i) {
        Rng rng(mix64(seed ^ (i * 0x632BE59BD9B4E019ull)));
        std::string s = src[0];
        mutate(s, p, rng);
        printf(">m%llu\n", (unsigned long long)i);
        put_wrapped(stdout, s, 70);
      }
    } else {
      throw Error("unknown mode '" + mode + "'");
    }
  } catch (const std::exception &e) {
    fprintf(stderr, "kpop_synth: %s\n", e.what());
    return 1;
  }
  fflush(stdout);
  return 0;
}

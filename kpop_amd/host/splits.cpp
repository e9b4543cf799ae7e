// splits.cpp -- see splits.h
#include "splits.h"

#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <set>

namespace kpop_host {

void write_splits(const std::string &path, const Splits &s, int precision) {
  FILE *f = (path == "/dev/stdout") ? stdout : fopen(path.c_str(), "wb");
  if (!f) throw Error("cannot write '" + path + "': " + strerror(errno));
  std::string line = "\"\"";
  for (const std::string &n : s.names) line += "\t\"" + n + "\"";
  line += "\n";
  fwrite(line.data(), 1, line.size(), f);
  char num[64];
  for (const Splits::Split &sp : s.splits) {
    line.assign(num, (size_t)snprintf(num, sizeof num, "%.*g", precision, sp.weight));
    for (uint32_t m : sp.members) line += "\t\"" + s.names[m] + "\"";
    line += "\n";
    fwrite(line.data(), 1, line.size(), f);
  }
  if (f != stdout) fclose(f);
  else fflush(f);
}

Splits read_splits(const std::string &path) {
  FILE *f = (path == "/dev/stdin") ? stdin : fopen(path.c_str(), "rb");
  if (!f) throw Error("cannot open '" + path + "': " + strerror(errno));
  Splits s;
  char *buf = nullptr;
  size_t cap = 0;
  ssize_t len;
  bool first = true;
  std::vector<std::pair<std::string, uint32_t>> index;
  auto fields = [](const std::string &l) {
    std::vector<std::string> out;
    size_t st = 0;
    for (;;) {
      const size_t p = l.find('\t', st);
      out.push_back(l.substr(st, p == std::string::npos ? p : p - st));
      if (p == std::string::npos) break;
      st = p + 1;
    }
    return out;
  };
  while ((len = getline(&buf, &cap, f)) >= 0) {
    while (len > 0 && (buf[len - 1] == '\n' || buf[len - 1] == '\r')) --len;
    if (len == 0) continue;
    const std::vector<std::string> fl = fields(std::string(buf, (size_t)len));
    if (first) {
      first = false;
      for (size_t i = 1; i < fl.size(); ++i) s.names.push_back(strip_external_quotes_and_check(fl[i]));
      for (uint32_t i = 0; i < s.names.size(); ++i) index.emplace_back(s.names[i], i);
      std::sort(index.begin(), index.end());
      continue;
    }
    Splits::Split sp;
    char *end = nullptr;
    sp.weight = strtod(fl[0].c_str(), &end);
    if (!end || *end) throw Error("splits file '" + path + "': Float_expected(\"" + fl[0] + "\")");
    for (size_t i = 1; i < fl.size(); ++i) {
      const std::string nm = strip_external_quotes_and_check(fl[i]);
      auto it = std::lower_bound(index.begin(), index.end(), std::make_pair(nm, 0u));
      if (it == index.end() || it->first != nm) throw Error("splits file '" + path + "': unknown leaf '" + nm + "'");
      sp.members.push_back(it->second);
    }
    std::sort(sp.members.begin(), sp.members.end());
    s.splits.push_back(std::move(sp));
  }
  free(buf);
  if (f != stdin) fclose(f);
  return s;
}

namespace {

struct Rng {  // the declared stand-in for OCaml's Random (splits.h)
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  bool boolean() { return next() & 1; }
  uint64_t integer(uint64_t n) { return next() % n; }
  double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

// SplitsAlgorithm.Bipartition.make, lib/Matrix.ml:361-521 (acceptance_probability_at_zero 0.2, magnification 10)
void bipartition(const double *emb, size_t d, const std::vector<uint32_t> &elements, Rng &rng, bool verbose, std::set<uint32_t> &best_one,
                 std::set<uint32_t> &best_two, double &best_objective) {
  const double inverse_acceptance = (1. - 0.2) / 0.2, negative_scale = -10.;
  const size_t num = elements.size();
  std::set<uint32_t> one, two;
  std::vector<double> c1(d, 0.), c2(d, 0.), o1(d, 0.), o2(d, 0.);
  for (uint32_t i : elements) {  // IntSet.iter: ascending
    const double *v = emb + (size_t)i * d;
    if (rng.boolean()) {
      two.insert(i);
      for (size_t j = 0; j < d; ++j) c2[j] = c2[j] + v[j];
    } else {
      one.insert(i);
      for (size_t j = 0; j < d; ++j) c1[j] = c1[j] + v[j];
    }
  }
  auto objective_of = [&]() {
    const double n1 = (double)one.size(), n2 = (double)two.size();
    double res = 0.;
    if (n1 > 0. && n2 > 0.)
      for (size_t j = 0; j < d; ++j) {
        const double a = n1 > 1. ? c1[j] / n1 : c1[j], b = n2 > 1. ? c2[j] / n2 : c2[j];
        res = res + (std::max(a, b) - std::min(a, b));
      }
    return res / sqrt(1. + fabs(n1 - n2));
  };
  double objective = objective_of();
  best_objective = objective;
  best_one = one;
  best_two = two;
  const size_t terminator = std::max<size_t>(num, 40);
  size_t rejected = 0, steps = 0;
  while (rejected < terminator) {
    if (verbose && steps % 1000 == 0) fprintf(stderr, " Step #%zu: objective=%.3g, max_objective=%.3g\n", steps, objective, best_objective);
    ++steps;
    const double old_objective = objective;
    o1 = c1;
    o2 = c2;
    const uint32_t selected = elements[rng.integer(num)];
    const double *v = emb + (size_t)selected * d;
    const bool from_one = one.count(selected) != 0;
    if (from_one) {
      one.erase(selected);
      two.insert(selected);
      for (size_t j = 0; j < d; ++j) {
        c1[j] = o1[j] - v[j];
        c2[j] = o2[j] + v[j];
      }
    } else {
      two.erase(selected);
      one.insert(selected);
      for (size_t j = 0; j < d; ++j) {
        c2[j] = o2[j] - v[j];
        c1[j] = o1[j] + v[j];
      }
    }
    objective = objective_of();
    const double delta = objective - old_objective;
    const double score = 1. / (1. + inverse_acceptance * exp(negative_scale * delta));
    if (rng.unit() <= score) {
      rejected = 0;
      if (objective > best_objective) {
        best_objective = objective;
        best_one = one;
        best_two = two;
      }
    } else {
      ++rejected;
      if (from_one) {
        two.erase(selected);
        one.insert(selected);
      } else {
        one.erase(selected);
        two.insert(selected);
      }
      c1.swap(o1);
      c2.swap(o2);
      objective = old_objective;
    }
  }
}

void refine(const double *emb, size_t d, const std::set<uint32_t> &set, Rng &rng, bool verbose, Splits &res) {
  Splits::Split sp;
  if (set.size() > 1) {
    std::vector<uint32_t> elements(set.begin(), set.end());
    std::set<uint32_t> one, two;
    double objective = 0.;
    bipartition(emb, d, elements, rng, verbose, one, two, objective);
    sp.members.assign(one.begin(), one.end());
    sp.weight = objective;
    res.splits.push_back(sp);
    refine(emb, d, one, rng, verbose, res);
    refine(emb, d, two, rng, verbose, res);
  } else {
    sp.members.assign(set.begin(), set.end());
    sp.weight = 0.;
    res.splits.push_back(sp);
  }
}

}  // namespace

Splits splits_centroids(const std::vector<std::string> &row_names, const double *emb, size_t n_dims, bool verbose, uint64_t seed) {
  Splits res;
  res.names = row_names;
  Rng rng{seed};
  std::set<uint32_t> all;
  for (uint32_t i = 0; i < row_names.size(); ++i) all.insert(i);
  refine(emb, n_dims, all, rng, verbose, res);
  return res;
}

}  // namespace kpop_host

// counter_db.cpp -- see counter_db.h
#include "counter_db.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <regex>

#include "../../include/kpop_hip.h"

namespace kpop_host {

namespace {

void check(int rc) {
  if (rc != 0) throw Error(std::string("libkpop_hip: ") + kpop_last_error());
}

// the GPU is claimed by the first action that needs it: loading, merging, selecting and saving are host-only
void ensure_gpu() {
  static bool done = false;
  if (done) return;
  int dev = 0;
  if (const char *e = getenv("KPOP_DEVICE")) dev = atoi(e);
  check(kpop_init(dev));
  done = true;
}

struct Lines {
  FILE *f = nullptr;
  bool own = false;
  char *buf = nullptr;
  size_t cap = 0;
  explicit Lines(const std::string &p) {
    if (p == "/dev/stdin") f = stdin;
    else {
      f = fopen(p.c_str(), "rb");
      own = true;
    }
    if (!f) throw Error("cannot open '" + p + "': " + strerror(errno));
  }
  ~Lines() {
    if (own && f) fclose(f);
    free(buf);
  }
  bool next(char **line, size_t *len) {  // input_line: the trailing newline is dropped
    ssize_t n = getline(&buf, &cap, f);
    if (n < 0) return false;
    if (n > 0 && buf[n - 1] == '\n') buf[--n] = 0;
    *line = buf;
    *len = (size_t)n;
    return true;
  }
};

std::vector<std::string> split(const char *s, size_t n, char sep) {  // String.Split.on_char_as_array
  std::vector<std::string> out;
  const char *e = s + n;
  for (const char *p = s;;) {
    const char *t = (const char *)memchr(p, sep, (size_t)(e - p));
    if (!t) {
      out.emplace_back(p, (size_t)(e - p));
      break;
    }
    out.emplace_back(p, (size_t)(t - p));
    p = t + 1;
  }
  return out;
}

// Int32.of_string: optional sign, decimal or 0x/0o/0b digits with '_' separators; failure = Wrong_format
bool int32_of_string(const char *s, size_t n, int32_t *out) {
  size_t i = 0;
  bool neg = false;
  if (i < n && (s[i] == '-' || s[i] == '+')) neg = s[i++] == '-';
  int base = 10;
  if (i + 1 < n && s[i] == '0') {
    const char c = s[i + 1];
    if (c == 'x' || c == 'X') base = 16;
    else if (c == 'o' || c == 'O') base = 8;
    else if (c == 'b' || c == 'B') base = 2;
    else if (c == 'u' || c == 'U') base = 10;
    if (base != 10 || c == 'u' || c == 'U') i += 2;
  }
  if (i >= n) return false;
  uint64_t v = 0;
  bool any = false;
  for (; i < n; ++i) {
    const char c = s[i];
    int d;
    if (c == '_' && any) continue;
    if (c >= '0' && c <= '9') d = c - '0';
    else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
    else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
    else return false;
    if (d >= base) return false;
    v = v * (uint64_t)base + (uint64_t)d;
    any = true;
    if (v > 0xFFFFFFFFull) return false;
  }
  if (!any) return false;
  if (base == 10) {
    if (v > (neg ? 0x80000000ull : 0x7FFFFFFFull)) return false;
    *out = (int32_t)(neg ? -(int64_t)v : (int64_t)v);
  } else {  // prefixed literals may use the full 32 bits and wrap
    *out = (int32_t)(uint32_t)(neg ? (uint64_t)(-(int64_t)v) : v);
  }
  return true;
}

// OCaml Str syntax -> ECMAScript.  Str: special are $^.*+?[]; \| alternation, \( \) groups, \N back-reference,
// \b word boundary, \c quotes c; ( ) | { } are ordinary characters.
std::string str_to_ecma(const std::string &r) {
  std::string o;
  for (size_t i = 0; i < r.size(); ++i) {
    const char c = r[i];
    if (c == '\\') {
      if (i + 1 >= r.size()) throw Error("Str.regexp: trailing backslash in '" + r + "'");
      const char d = r[++i];
      if (d == '|') o += '|';
      else if (d == '(') o += '(';
      else if (d == ')') o += ')';
      else if (d == 'b') o += "\\b";
      else if (d >= '1' && d <= '9') {
        o += '\\';
        o += d;
      } else {
        if (strchr("^$.*+?()[]{}|\\/-", d)) o += '\\';
        o += d;
      }
    } else if (c == '[') {  // character set: copied through; a backslash is an ordinary member in Str
      size_t j = i + 1;
      o += '[';
      if (j < r.size() && r[j] == '^') {
        o += '^';
        ++j;
      }
      if (j < r.size() && r[j] == ']') {
        o += "\\]";
        ++j;
      }
      for (; j < r.size() && r[j] != ']'; ++j) {
        if (r[j] == '\\' || r[j] == '[') o += '\\';
        o += r[j];
      }
      if (j >= r.size()) throw Error("Str.regexp: unterminated character set in '" + r + "'");
      o += ']';
      i = j;
    } else if (strchr("(){}|/", c)) {
      o += '\\';
      o += c;
    } else {
      o += c;
    }
  }
  return o;
}

FILE *open_out(const std::string &path) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) throw Error("cannot write '" + path + "': " + strerror(errno));
  return f;
}

std::string counter_filename(const std::string &prefix, bool table) {  // lib/KMerDB.ml:391-393,1009-1011
  if (prefix.size() >= 5 && prefix.compare(0, 5, "/dev/") == 0) return prefix;
  return prefix + (table ? ".KPopCounter.txt" : ".KPopCounter");
}

}  // namespace

int Transform::code() const {  // Transformation.of_parameters, lib/KMerDB.ml:151-163
  if (which == "binary") return KPOP_TRANSF_BINARY;
  if (which == "power" || which == "pow") return KPOP_TRANSF_POWER;
  if (which == "clr" || which == "CLR") return KPOP_TRANSF_CLR;
  if (which == "pseudocounts" || which == "pseudo") return KPOP_TRANSF_PSEUDO;
  throw Error("Unknown_transformation(\"" + which + "\")");
}

RegexpSelector parse_regexp_selector(const std::string &s) {
  RegexpSelector out;
  for (const std::string &l : split(s.data(), s.size(), ',')) {
    const std::vector<std::string> f = split(l.data(), l.size(), '~');
    if (f.size() != 2) throw Error("Wrong number of fields in list (expected 2, found " + std::to_string(f.size()) + ")");
    (void)str_to_ecma(f[1]);  // Str.regexp raises at parse time
    out.push_back({f[0], f[1]});
  }
  return out;
}

bool str_string_match(const std::string &str_regexp, const std::string &s) {
  const std::regex re(str_to_ecma(str_regexp), std::regex::ECMAScript);
  return std::regex_search(s, re, std::regex_constants::match_continuous);
}

bool CounterDB::hex_key(const char *s, size_t n, uint64_t *key) {
  if (n == 0 || n > 15) return false;
  uint64_t v = 0;
  for (size_t i = 0; i < n; ++i) {
    const char c = s[i];
    int d;
    if (c >= '0' && c <= '9') d = c - '0';
    else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
    else return false;
    v = (v << 4) | (uint64_t)d;
  }
  *key = ((uint64_t)n << 60) | v;
  return true;
}

uint32_t CounterDB::row_of(const char *name, size_t len) {
  uint64_t key;
  if (hex_key(name, len, &key)) {
    bool added;
    const uint32_t row = hex_row_idx_.find_or_add(key, (uint32_t)core.row_names.size(), &added);
    if (added) core.row_names.emplace_back(name, len);
    return row;
  }
  std::string nm(name, len);
  auto it = row_idx_.find(nm);
  if (it != row_idx_.end()) return it->second;
  const uint32_t row = (uint32_t)core.row_names.size();
  row_idx_.emplace(nm, row);
  core.row_names.push_back(std::move(nm));
  return row;
}

void CounterDB::rebuild_indices() {  // invert_table, lib/KMerDB.ml:371-374 (Hashtbl.add: the last duplicate wins)
  col_idx_.clear();
  row_idx_.clear();
  hex_row_idx_.clear();
  meta_idx_.clear();
  for (size_t i = 0; i < core.col_names.size(); ++i) col_idx_[core.col_names[i]] = (uint32_t)i;
  hex_row_idx_.reserve(core.row_names.size());
  for (size_t i = 0; i < core.row_names.size(); ++i) {
    const std::string &nm = core.row_names[i];
    uint64_t key;
    if (hex_key(nm.data(), nm.size(), &key)) hex_row_idx_.set(key, (uint32_t)i);
    else row_idx_[nm] = (uint32_t)i;
  }
  for (size_t i = 0; i < core.meta_names.size(); ++i) meta_idx_[core.meta_names[i]] = (uint32_t)i;
}

uint32_t CounterDB::add_empty_column_if_needed(const std::string &label) {
  auto it = col_idx_.find(label);
  if (it != col_idx_.end()) return it->second;
  const uint32_t c = (uint32_t)core.col_names.size();
  col_idx_[label] = c;
  core.col_names.push_back(label);
  core.meta.emplace_back(n_meta(), std::string());
  core.storage.emplace_back();
  return c;
}

std::vector<const int32_t *> CounterDB::columns() {
  std::vector<const int32_t *> out(n_cols());
  const size_t nr = n_rows();
  parallel_for(n_cols(), 1, [&](size_t lo, size_t hi) {  // (a thousand spectra to pad: a reallocation and a copy each)
    for (size_t c = lo; c < hi; ++c) {
      if (core.storage[c].size() != nr) core.storage[c].resize(nr, 0);
      out[c] = core.storage[c].data();
    }
  });
  return out;
}

CounterDB CounterDB::of_binary(const std::string &prefix) {
  CounterDB db;
  db.core = read_binary_counter(counter_filename(prefix, false));
  db.rebuild_indices();
  return db;
}

void CounterDB::to_binary(const std::string &prefix) {
  columns();
  write_binary_counter(counter_filename(prefix, false), core);
}

void CounterDB::add_meta(const std::string &fname) {
  Lines in(fname);
  char *line;
  size_t len;
  if (!in.next(&line, &len)) throw Error("End_of_file");  // input_line on an empty file
  std::vector<std::string> header = split(line, len, '\t');
  for (std::string &h : header) h = strip_external_quotes_and_check(h);
  for (size_t i = 1; i < header.size(); ++i)
    if (!meta_idx_.count(header[i])) {
      meta_idx_[header[i]] = (uint32_t)core.meta_names.size();
      core.meta_names.push_back(header[i]);
    }
  for (auto &m : core.meta) m.resize(n_meta());
  uint64_t line_num = 1;
  while (in.next(&line, &len)) {
    ++line_num;
    std::vector<std::string> f = split(line, len, '\t');
    for (std::string &v : f) v = strip_external_quotes_and_check(v);
    if (f.size() != header.size())
      throw Error("Wrong_number_of_columns(" + std::to_string(line_num) + ", " + std::to_string(f.size()) + ", " +
                  std::to_string(header.size()) + ")");
    const uint32_t c = add_empty_column_if_needed(f[0]);
    for (size_t i = 1; i < header.size(); ++i) core.meta[c][meta_idx_[header[i]]] = f[i];
  }
}

// One file of spectra (lib/KMerDB.ml:505-575), a block at a time (blocks end where a spectrum ends, kpop_text.h).  A block whose
// every line is what KPopCount writes -- a fixed number of lowercase hexadecimal digits, a tab, a decimal count -- is parsed by
// the host threads into (hash, count) arrays and entered from there; any other block goes line by line, as the whole file used
// to (30 M lines of a thousand genomes: 4.2 s).  Rows are still numbered in the order in which their names first appear.
void CounterDB::add_spectra_text(int fd, const std::string &fname) {
  SpectraTextStream ts(fd, nullptr, 0);
  TextBlock block;
  const bool timing = getenv("KPOP_TIMING") != nullptr;
  double t_phase[6] = {0, 0, 0, 0, 0, 0};  // read, parse, look-ups, new rows, columns, line by line
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto lap = [&](int which, std::chrono::steady_clock::time_point &t0) {
    const auto t1 = now();
    t_phase[which] += std::chrono::duration<double>(t1 - t0).count();
    t0 = t1;
  };
  auto t0 = now();
  uint64_t line_num = 0, n_spectra = 0;
  uint32_t col = 0;
  bool first_block = true;
  size_t name_len = 0;
  auto line_by_line = [&](const char *b, const char *e) {
    while (b < e) {
      const char *nl = (const char *)memchr(b, '\n', (size_t)(e - b));
      const char *line = b;
      const size_t len = (size_t)((nl ? nl : e) - b);
      b = nl ? nl + 1 : e;
      ++line_num;
      const char *tab = (const char *)memchr(line, '\t', len);
      if (!tab || memchr(tab + 1, '\t', (size_t)(line + len - tab - 1))) {
        size_t n = 1;
        for (size_t i = 0; i < len; ++i) n += line[i] == '\t';
        throw Error("Wrong_number_of_columns(" + std::to_string(line_num) + ", " + std::to_string(n) + ", 2)");
      }
      const size_t l0 = (size_t)(tab - line);
      if (line_num == 1 && l0 != 0) throw Error("Header_expected(\"" + std::string(line, len) + "\")");
      if (l0 == 0) {  // header: a new (or an existing) spectrum
        col = add_empty_column_if_needed(strip_external_quotes_and_check(std::string(tab + 1, len - 1)));
        if (core.storage[col].size() < n_rows()) core.storage[col].resize(n_rows(), 0);  // one allocation for the rows known so far
        ++n_spectra;
        continue;
      }
      const uint32_t row = row_of(line, l0);
      int32_t v;
      if (!int32_of_string(tab + 1, len - l0 - 1, &v))
        throw Error("Wrong_format(" + std::to_string(line_num) + ", \"" + std::string(tab + 1, len - l0 - 1) + "\")");
      std::vector<int32_t> &s = core.storage[col];
      if (s.size() <= row) s.resize(std::max<size_t>(row + 1, s.size() + s.size() / 2), 0);
      s[row] = (int32_t)((uint32_t)s[row] + (uint32_t)v);  // repeated k-mers accumulate (:561-562)
    }
  };
  while (ts.next(block)) {
    lap(0, t0);
    const char *b = block.data(), *e = b + block.size();
    if (first_block) {  // the width of the names, from the first line that is not a header
      for (const char *p = b; p < e;) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        const char *le = nl ? nl : e;
        if (le > p && *p != '\t') {
          const char *tab = (const char *)memchr(p, '\t', (size_t)(le - p));
          uint64_t key;
          if (tab && hex_key(p, (size_t)(tab - p), &key)) name_len = (size_t)(tab - p);
          break;
        }
        if (!nl) break;
        p = nl + 1;
      }
    }
    bool plain = false;
    HashedSpectra sp;
    uint64_t n_lines = 0;
    static const bool lines_only = getenv("KPOP_COUNTDB_LINES") != nullptr;  // (tests: everything line by line)
    if (name_len && !lines_only && !(first_block && (b == e || *b != '\t'))) {
      try {
        parse_spectra_block(b, block.size(), name_len, ~0ull >> 1, first_block, line_num, sp, &n_lines, 0, &plain);
      } catch (const std::exception &) {
        plain = false;  // the line-by-line pass raises it, with the message this tool gives
      }
      for (size_t i = 0; plain && i < sp.values.size(); ++i) plain = sp.values[i] <= 2147483647.0;
    }
    first_block = false;
    lap(1, t0);
    if (!plain) {
      line_by_line(b, e);
      lap(5, t0);
      continue;
    }
    // rows of the k-mers this database knows already, looked up by the host threads; the new ones entered one by one in the
    // order of their first appearance (what numbers the rows); then every spectrum adds its counts into its own column
    const size_t n_lines_data = sp.hash.size();
    const uint64_t tag = (uint64_t)name_len << 60;
    std::vector<uint32_t> row(n_lines_data);
    parallel_for(n_lines_data, 65536, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) row[i] = hex_row_idx_.find(tag | sp.hash[i]);
    });
    lap(2, t0);
    char name[16];
    for (size_t i = 0; i < n_lines_data; ++i)
      if (row[i] == HexIndex::kNone) {
        bool added;
        row[i] = hex_row_idx_.find_or_add(tag | sp.hash[i], (uint32_t)core.row_names.size(), &added);
        if (added) {
          uint64_t v = sp.hash[i];
          for (size_t d = name_len; d-- > 0;) {
            name[d] = "0123456789abcdef"[v & 15];
            v >>= 4;
          }
          core.row_names.emplace_back(name, name_len);
        }
      }
    lap(3, t0);
    std::vector<uint32_t> col_of(sp.labels.size());
    bool distinct = true;  // (a label that comes twice in a block shares a column: then the spectra are entered in order)
    {
      std::vector<uint32_t> seen;
      for (size_t s_i = 0; s_i < sp.labels.size(); ++s_i) {
        col_of[s_i] = add_empty_column_if_needed(sp.labels[s_i]);
        seen.push_back(col_of[s_i]);
      }
      std::sort(seen.begin(), seen.end());
      distinct = std::adjacent_find(seen.begin(), seen.end()) == seen.end();
    }
    n_spectra += sp.labels.size();
    if (!sp.labels.empty()) col = col_of.back();
    const size_t rows_now = n_rows();
    auto enter = [&](size_t lo, size_t hi) {
      for (size_t s_i = lo; s_i < hi; ++s_i) {
        std::vector<int32_t> &s = core.storage[col_of[s_i]];
        if (s.size() < rows_now) s.resize(rows_now, 0);
        for (uint64_t i = sp.offsets[s_i]; i < sp.offsets[s_i + 1]; ++i)
          s[row[i]] = (int32_t)((uint32_t)s[row[i]] + (uint32_t)(int32_t)sp.values[i]);  // repeated k-mers accumulate (:561-562)
      }
    };
    if (distinct) parallel_for(sp.labels.size(), 1, enter);
    else enter(0, sp.labels.size());
    lap(4, t0);
    line_num += n_lines;
  }
  if (timing)
    fprintf(stderr, "[timing] KPopCountDB:   spectra text: reading %.3f s, parsing %.3f s, look-ups %.3f s, new rows %.3f s, columns %.3f s, line by line %.3f s\n",
            t_phase[0], t_phase[1], t_phase[2], t_phase[3], t_phase[4], t_phase[5]);
  if (verbose)
    fprintf(stderr, "(KPopCountDB): File '%s': Read %llu spectra on %llu lines.\n", fname.c_str(), (unsigned long long)n_spectra,
            (unsigned long long)line_num);
}

void CounterDB::add_files(const std::vector<std::string> &prefixes) {
  for (const std::string &prefix : prefixes) {
    const std::string fname = make_filename(prefix, "KPopSpectra", true);
    const bool is_stdin = fname == "/dev/stdin";
    const int fd = is_stdin ? 0 : open(fname.c_str(), O_RDONLY);
    if (fd < 0) throw Error("cannot open '" + fname + "': " + strerror(errno));
    try {
      add_spectra_text(fd, fname);
    } catch (...) {
      if (!is_stdin) close(fd);
      throw;
    }
    if (!is_stdin) close(fd);
  }
  columns();
}

std::set<std::string> CounterDB::selected_from_regexps(const RegexpSelector &regexps) const {
  std::vector<std::pair<int, std::regex>> compiled;  // metadata index (-1 label, -2 unknown field), regexp
  for (const auto &pr : regexps) {
    int which = -1;
    if (!pr.first.empty()) {
      auto it = meta_idx_.find(pr.first);
      which = it == meta_idx_.end() ? -2 : (int)it->second;
      if (which == -2 && verbose)
        fprintf(stderr, "(KPopCountDB): WARNING: Metadata field '%s' not found, no column will match\n", pr.first.c_str());
    }
    compiled.emplace_back(which, std::regex(str_to_ecma(pr.second), std::regex::ECMAScript));
  }
  std::set<std::string> res;
  for (size_t c = 0; c < n_cols(); ++c) {
    bool ok = true;
    for (const auto &cr : compiled) {
      if (cr.first == -2) ok = false;
      else {
        const std::string &subject = cr.first == -1 ? core.col_names[c] : core.meta[c][(size_t)cr.first];
        ok = std::regex_search(subject, cr.second, std::regex_constants::match_continuous);
      }
      if (!ok) break;
    }
    if (ok) res.insert(core.col_names[c]);
  }
  return res;
}

std::set<std::string> CounterDB::selected_negate(const std::set<std::string> &sel) const {
  std::set<std::string> res;
  for (const std::string &c : core.col_names)
    if (!sel.count(c)) res.insert(c);
  return res;
}

void CounterDB::add_combined_selected(const std::string &new_label, const std::set<std::string> &selection, int criterion) {
  // the reference takes its statistics, and lets its workers read the spectra, before the new column exists
  // (:644-648): the combination of a label with itself uses the old counts
  std::vector<uint32_t> found;  // `found_cols`: labels visited ascending, accumulated in front (:650-660)
  for (const std::string &label : selection) {
    auto it = col_idx_.find(label);
    if (it != col_idx_.end()) found.insert(found.begin(), it->second);
  }
  ensure_gpu();
  std::vector<const int32_t *> all = columns();
  std::vector<const int32_t *> cols(found.size());
  std::vector<uint32_t> sel(found.size());
  for (size_t i = 0; i < found.size(); ++i) {
    cols[i] = all[found[i]];
    sel[i] = (uint32_t)i;
  }
  std::vector<double> stats(4 * std::max<size_t>(1, cols.size())), col_sum(std::max<size_t>(1, cols.size()));
  check(kpop_counter_stats(cols.data(), (uint32_t)cols.size(), n_rows(), 1., 1., stats.data(), nullptr));
  for (size_t i = 0; i < cols.size(); ++i) col_sum[i] = stats[4 * i + 2];
  std::vector<int32_t> out(n_rows());
  double norm = 0.;
  check(kpop_counter_combine(cols.data(), n_rows(), sel.data(), (uint32_t)sel.size(), col_sum.data(), criterion, out.data(), &norm));
  if (verbose)
    fprintf(stderr, "(KPopCountDB): Adding/replacing spectrum '%s': n_found=%zu. Norm=%.16g\n", new_label.c_str(), found.size(), norm);
  const uint32_t nc = add_empty_column_if_needed(new_label);
  core.storage[nc] = std::move(out);
  if (n_meta() > 0) {  // one value shared by every selected spectrum is inherited, anything else is blank (:737-761)
    std::vector<std::set<std::string>> vals(n_meta());
    for (uint32_t c : found)
      for (size_t m = 0; m < n_meta(); ++m) vals[m].insert(core.meta[c][m]);
    for (size_t m = 0; m < n_meta(); ++m) core.meta[nc][m] = vals[m].size() == 1 ? *vals[m].begin() : std::string();
  }
}

void CounterDB::remove_selected(const std::set<std::string> &selected) {
  CounterCore nc;
  nc.row_names = std::move(core.row_names);
  nc.meta_names = std::move(core.meta_names);
  for (size_t c = 0; c < core.col_names.size(); ++c)
    if (!selected.count(core.col_names[c])) {
      nc.col_names.push_back(std::move(core.col_names[c]));
      nc.meta.push_back(std::move(core.meta[c]));
      nc.storage.push_back(std::move(core.storage[c]));
    }
  core = std::move(nc);
  rebuild_indices();
}

void CounterDB::split_spectra(const std::string &classes_label, int criterion) {
  auto mi = meta_idx_.find(classes_label);
  if (mi == meta_idx_.end()) throw Error("Classes_label_not_found(\"" + classes_label + "\")");  // :740-741
  // get_indicator_vector (:742-764): classes numbered in order of first appearance
  std::map<std::string, uint32_t> class_to_ind;
  std::vector<std::string> ind_to_class;
  std::vector<std::set<std::string>> members;
  for (size_t c = 0; c < n_cols(); ++c) {
    const std::string &cl = core.meta[c][mi->second];
    auto it = class_to_ind.find(cl);
    uint32_t ind;
    if (it == class_to_ind.end()) {
      ind = (uint32_t)ind_to_class.size();
      class_to_ind[cl] = ind;
      ind_to_class.push_back(cl);
      members.emplace_back();
    } else {
      ind = it->second;
    }
    members[ind].insert(core.col_names[c]);
  }
  const std::set<std::string> originals(core.col_names.begin(), core.col_names.end());
  for (size_t ind = 0; ind < ind_to_class.size(); ++ind) {
    if (originals.count(ind_to_class[ind])) throw Error("Class_label_is_also_spectrum_name(\"" + ind_to_class[ind] + "\")");  // :806-807
    add_combined_selected(ind_to_class[ind], members[ind], criterion);
  }
  remove_selected(originals);
}

void CounterDB::output_summary() const {
  fprintf(stderr, "[Spectrum labels (%zu)]:", n_cols());
  for (const std::string &s : core.col_names) fprintf(stderr, " '%s'", s.c_str());
  fprintf(stderr, "\n");
  if (verbose) {
    fprintf(stderr, "[K-mer hashes (%zu)]:", n_rows());
    for (const std::string &s : core.row_names) fprintf(stderr, " '%s'", s.c_str());
    fprintf(stderr, "\n");
  }
  fprintf(stderr, "[Meta-data fields (%zu)]:", n_meta());
  for (const std::string &s : core.meta_names) fprintf(stderr, " '%s'", s.c_str());
  fprintf(stderr, "\n");
}

namespace {

// what to_table and to_spectra share: statistics over the whole register and the surviving rows and columns.  The
// transformed counts are then produced block by block (a band of k-mers for the k-mer-major table, a batch of
// spectra for the transposed table and for spectra), so the host never holds more than block_values() doubles.
size_t block_values() {  // KPOP_HOST_BLOCK: the tests force tiny blocks through the banding logic
  if (const char *e = getenv("KPOP_HOST_BLOCK")) return (size_t)std::max(1, atoi(e));
  return 64u << 20;
}

struct Plan {
  std::vector<uint32_t> rows, cols;
  std::vector<const int32_t *> col_ptrs;
  std::vector<double> cs;  // 4 statistics per surviving column
  int which = KPOP_TRANSF_POWER;
  double threshold = 1., power = 1.;
  size_t n_rows = 0;
};

Plan plan_register(CounterDB &db, const TableFilter &filter) {
  Plan p;
  p.which = filter.transform.code();
  p.threshold = filter.transform.threshold;
  p.power = filter.transform.power;
  ensure_gpu();
  const size_t n_rows = db.n_rows(), n_cols = db.n_cols();
  p.n_rows = n_rows;
  std::vector<const int32_t *> all = db.columns();
  std::vector<double> col_stats(4 * std::max<size_t>(1, n_cols)), row_stats(4 * std::max<size_t>(1, n_rows));
  check(kpop_counter_stats(all.data(), (uint32_t)n_cols, n_rows, p.threshold, p.power, col_stats.data(), row_stats.data()));
  for (size_t r = 0; r < n_rows; ++r)
    if (row_stats[4 * r + 2] > 0. || filter.print_zero_rows) p.rows.push_back((uint32_t)r);
  for (size_t c = 0; c < n_cols; ++c)
    if (!filter.filter_columns.count(db.core.col_names[c])) {
      p.cols.push_back((uint32_t)c);
      p.col_ptrs.push_back(all[c]);
      p.cs.insert(p.cs.end(), col_stats.begin() + 4 * (long)c, col_stats.begin() + 4 * (long)c + 4);
    }
  return p;
}

// out[(c - c0) * (r1 - r0) + (r - r0)] for surviving columns [c0, c1) and k-mers [r0, r1)
void eval_block(const Plan &p, size_t c0, size_t c1, size_t r0, size_t r1, std::vector<double> &out) {
  out.resize(std::max<size_t>(1, (c1 - c0) * (r1 - r0)));
  if (c1 == c0 || r1 == r0) return;
  std::vector<const int32_t *> ptrs(c1 - c0);
  for (size_t c = c0; c < c1; ++c) ptrs[c - c0] = p.col_ptrs[c] + r0;
  check(kpop_counter_transform(ptrs.data(), (uint32_t)(c1 - c0), r1 - r0, p.which, p.threshold, p.power, p.cs.data() + 4 * c0, 0,
                               out.data()));
}

}  // namespace

void CounterDB::to_table(const TableFilter &filter, const std::string &prefix) {
  const Plan t = plan_register(*this, filter);
  const std::string fname = counter_filename(prefix, true);
  FILE *out = open_out(fname);
  std::vector<char> iobuf(1 << 22);
  setvbuf(out, iobuf.data(), _IOFBF, iobuf.size());
  const size_t nr = n_rows(), ncs = t.cols.size();
  std::vector<uint32_t> meta;
  if (filter.print_metadata)
    for (size_t m = 0; m < n_meta(); ++m) meta.push_back((uint32_t)m);
  const bool rn = filter.print_row_names;
  std::vector<double> values;
  if (meta.size() + t.rows.size() > 0) {
    if (filter.transpose) {  // rows are spectra (:1058-1112)
      if (filter.print_col_names) {
        bool first_done = false;
        for (uint32_t m : meta) {
          fprintf(out, "%s%s", (first_done || rn) ? "\t" : "", core.meta_names[m].c_str());
          first_done = true;
        }
        for (uint32_t r : t.rows) {
          fprintf(out, "%s%s", (first_done || rn) ? "\t" : "", core.row_names[r].c_str());
          first_done = true;
        }
        fputc('\n', out);
      }
      const size_t batch = std::max<size_t>(1, block_values() / std::max<size_t>(1, nr));
      for (size_t c0 = 0; c0 < ncs; c0 += batch) {
        const size_t c1 = std::min(ncs, c0 + batch);
        eval_block(t, c0, c1, 0, nr, values);
        for (size_t i = c0; i < c1; ++i) {
          const uint32_t c = t.cols[i];
          if (rn) fputs(core.col_names[c].c_str(), out);
          bool first_done = false;
          for (uint32_t m : meta) {
            fprintf(out, "%s%s", (first_done || rn) ? "\t" : "", core.meta[c][m].c_str());
            first_done = true;
          }
          for (uint32_t r : t.rows) {
            fprintf(out, "%s%.*g", (first_done || rn) ? "\t" : "", filter.precision, values[(i - c0) * nr + r]);
            first_done = true;
          }
          fputc('\n', out);
        }
      }
    } else {  // rows are k-mers (:1113-1160)
      if (filter.print_col_names) {
        for (size_t i = 0; i < ncs; ++i) fprintf(out, "%s%s", (i > 0 || rn) ? "\t" : "", core.col_names[t.cols[i]].c_str());
        fputc('\n', out);
      }
      for (uint32_t m : meta) {
        if (rn) fputs(core.meta_names[m].c_str(), out);
        for (size_t i = 0; i < ncs; ++i) fprintf(out, "%s%s", (i > 0 || rn) ? "\t" : "", core.meta[t.cols[i]][m].c_str());
        fputc('\n', out);
      }
      const size_t band = std::max<size_t>(1, block_values() / std::max<size_t>(1, ncs));
      size_t next = 0;  // position in t.rows
      for (size_t r0 = 0; r0 < nr && next < t.rows.size(); r0 += band) {
        const size_t r1 = std::min(nr, r0 + band);
        if (t.rows[next] >= r1) continue;
        eval_block(t, 0, ncs, r0, r1, values);
        for (; next < t.rows.size() && t.rows[next] < r1; ++next) {
          const uint32_t r = t.rows[next];
          if (rn) fputs(core.row_names[r].c_str(), out);
          for (size_t i = 0; i < ncs; ++i) fprintf(out, "%s%.*g", (i > 0 || rn) ? "\t" : "", filter.precision, values[i * (r1 - r0) + (r - r0)]);
          fputc('\n', out);
        }
      }
    }
  }
  if (fclose(out) != 0) throw Error("write to '" + fname + "' failed");
}

void CounterDB::to_spectra(const TableFilter &filter, const std::string &prefix) {
  const Plan t = plan_register(*this, filter);
  const std::string fname = make_filename(prefix, "KPopSpectra", true);
  FILE *out = open_out(fname);
  std::vector<char> iobuf(1 << 22);
  setvbuf(out, iobuf.data(), _IOFBF, iobuf.size());
  const size_t nr = n_rows(), ncs = t.cols.size();
  std::vector<double> values;
  const size_t batch = std::max<size_t>(1, block_values() / std::max<size_t>(1, nr));
  for (size_t c0 = 0; c0 < ncs; c0 += batch) {
    const size_t c1 = std::min(ncs, c0 + batch);
    eval_block(t, c0, c1, 0, nr, values);
    for (size_t i = c0; i < c1; ++i) {
      fprintf(out, "\t%s\n", core.col_names[t.cols[i]].c_str());
      for (uint32_t r : t.rows) {
        const double v = values[(i - c0) * nr + r];
        if (v > 0.) fprintf(out, "%s\t%.*g\n", core.row_names[r].c_str(), filter.precision, v);  // :1222-1223
      }
    }
  }
  if (fclose(out) != 0) throw Error("write to '" + fname + "' failed");
}

void CounterDB::to_distances(int kind, double p, bool normalise, const std::set<std::string> &sel1, const std::set<std::string> &sel2,
                             const std::string &prefix) {
  const size_t nr = n_rows(), nc = n_cols();
  ensure_gpu();
  std::vector<const int32_t *> all = columns();
  std::vector<double> stats(4 * std::max<size_t>(1, nc));
  check(kpop_counter_stats(all.data(), (uint32_t)nc, nr, 1., 1., stats.data(), nullptr));
  Table result;
  auto submatrix = [&](const std::set<std::string> &sel, std::vector<std::string> *names) {  // make_submatrix, :1243-1268
    std::vector<double> m;
    for (size_t c = 0; c < nc; ++c)
      if (sel.count(core.col_names[c])) {
        names->push_back(core.col_names[c]);
        double norm = stats[4 * c + 2];
        if (!normalise || norm == 0.) norm = 1.;
        const size_t at = m.size();
        m.resize(at + nr);
        for (size_t r = 0; r < nr; ++r) m[at + r] = (double)all[c][r] / norm;
      }
    return m;
  };
  std::vector<double> m1 = submatrix(sel1, &result.col_names), m2 = submatrix(sel2, &result.row_names);
  if (nr > 0xFFFFFFFFull) throw Error("more than 2^32 k-mers");
  std::vector<double> metric(std::max<size_t>(1, nr), 1.);
  result.data.resize(result.col_names.size() * result.row_names.size());
  if (!result.data.empty()) {
    if (nr == 0) std::fill(result.data.begin(), result.data.end(), 0.);
    else
      check(kpop_distance_rowwise(m1.data(), (uint32_t)result.col_names.size(), m2.data(), (uint32_t)result.row_names.size(), (uint32_t)nr,
                                  metric.data(), kind, p, 1, result.data.data()));
  }
  write_binary_matrix(make_filename(prefix, "KPopDMatrix", false), "KPopDMatrix", result);
}

}  // namespace kpop_host

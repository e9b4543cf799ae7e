// kpop_text.cpp -- see kpop_text.h
#include "kpop_text.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

namespace kpop_host {

// ------------------------------------------------------------------ names
std::string strip_external_quotes_and_check(const std::string &s) {
  std::string r = s;
  if (r.size() >= 2 && r.front() == '"' && r.back() == '"') r = r.substr(1, r.size() - 2);
  if (r.find('"') != std::string::npos) throw Error("Quotes_in_name(\"" + s + "\")");
  return r;
}

std::string make_filename(const std::string &prefix, const std::string &type_name, bool table) {
  if (prefix.size() >= 5 && prefix.compare(0, 5, "/dev/") == 0) return prefix;
  return prefix + "." + type_name + (table ? ".txt" : "");
}

std::string hash_to_hex(uint64_t h, int k) {
  const int digits = (k + 1) / 2;
  std::string s((size_t)digits, '0');
  static const char hx[] = "0123456789abcdef";
  for (int i = digits - 1; i >= 0; --i) {
    s[(size_t)i] = hx[h & 15];
    h >>= 4;
  }
  return s;
}

bool hex_to_hash(const std::string &s, uint64_t *h) {
  if (s.empty() || s.size() > 16) return false;
  uint64_t v = 0;
  for (char c : s) {
    int d;
    if (c >= '0' && c <= '9') d = c - '0';
    else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
    else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
    else return false;
    v = (v << 4) | (uint64_t)d;
  }
  *h = v;
  return true;
}

std::string format_g(double x, int precision) {
  char buf[64];
  snprintf(buf, sizeof(buf), "%.*g", precision, x);
  return buf;
}

// ------------------------------------------------------------------ line reader
namespace {
struct LineReader {
  FILE *f = nullptr;
  bool own = false;
  char *buf = nullptr;
  size_t cap = 0;
  std::string path;
  explicit LineReader(const std::string &p) : path(p) {
    if (p == "/dev/stdin" || p == "-") f = stdin;
    else {
      f = fopen(p.c_str(), "rb");
      own = true;
    }
    if (!f) throw Error("cannot open '" + p + "': " + strerror(errno));
  }
  ~LineReader() {
    if (own && f) fclose(f);
    free(buf);
  }
  // returns false at EOF; strips the trailing \n and \r
  bool next(std::string &line) {
    ssize_t n = getline(&buf, &cap, f);
    if (n < 0) return false;
    while (n > 0 && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) --n;
    line.assign(buf, (size_t)n);
    return true;
  }
};

void split_tabs(const std::string &line, std::vector<std::string> &out) {
  out.clear();
  size_t start = 0;
  for (;;) {
    size_t p = line.find('\t', start);
    if (p == std::string::npos) {
      out.emplace_back(line, start);
      return;
    }
    out.emplace_back(line, start, p - start);
    start = p + 1;
  }
}

bool parse_float(const std::string &s, double *v) {
  // OCaml float_of_string accepts decimal/hex floats, "inf", "nan", '_' separators; strtod covers the
  // forms the writers of this format produce (%d, %.15g)
  if (s.empty()) return false;
  char *end = nullptr;
  errno = 0;
  *v = strtod(s.c_str(), &end);
  return end && *end == 0;
}
}  // namespace

// ------------------------------------------------------------------ sequences
struct SeqReader::Impl {
  LineReader lr;
  SeqFormat fmt;
  std::string pending;  // a FASTA header read ahead
  bool have_pending = false;
  Impl(const std::string &p, SeqFormat f) : lr(p), fmt(f) {}
};

SeqReader::SeqReader(const std::string &path, SeqFormat fmt) : p_(new Impl(path, fmt)) {}
SeqReader::~SeqReader() { delete p_; }

static inline void lint_append(const std::string &line, std::string &seq) {
  for (unsigned char c : line) {
    if (c == '-' || c == ' ' || c == '\t') continue;      // keep_dashes:false
    if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);  // keep_lowercase:false
    seq.push_back((char)c);
  }
}

bool SeqReader::next_record(std::string &tag, std::string &seq) {
  Impl &I = *p_;
  std::string line;
  seq.clear();
  if (I.fmt == SeqFormat::FASTA) {
    if (!I.have_pending) {
      for (;;) {
        if (!I.lr.next(line)) return false;
        if (line.empty()) continue;
        if (line[0] != '>') throw Error("FASTA file '" + I.lr.path + "': expected '>' at the start of a record");
        I.pending = line;
        break;
      }
    }
    tag = I.pending.substr(1);
    I.have_pending = false;
    while (I.lr.next(line)) {
      if (!line.empty() && line[0] == '>') {
        I.pending = line;
        I.have_pending = true;
        break;
      }
      lint_append(line, seq);
    }
    return true;
  }
  // FASTQ: 4 lines per record
  for (;;) {
    if (!I.lr.next(line)) return false;
    if (!line.empty()) break;
  }
  if (line[0] != '@') throw Error("FASTQ file '" + I.lr.path + "': expected '@' at the start of a record");
  tag = line.substr(1);
  std::string s, plus, qual;
  if (!I.lr.next(s) || !I.lr.next(plus) || !I.lr.next(qual) || plus.empty() || plus[0] != '+')
    throw Error("FASTQ file '" + I.lr.path + "': truncated record '" + tag + "'");
  lint_append(s, seq);
  return true;
}

bool SeqReader::next_batch(ReadBatch &out, uint64_t max_bases, uint64_t max_reads) {
  if (out.offsets.empty()) out.offsets.assign(1, 0);
  std::string tag, seq;
  bool any = false;
  while (out.size() < max_reads && out.bases.size() < max_bases) {
    if (!next_record(tag, seq)) break;
    any = true;
    out.bases.insert(out.bases.end(), seq.begin(), seq.end());
    out.offsets.push_back(out.bases.size());
    out.tags.push_back(tag);
  }
  return any;
}

// ------------------------------------------------------------------ spectra text
void read_spectra_file(const std::string &path, Spectra &out) {
  LineReader lr(path);
  std::string line;
  std::vector<std::string> f;
  uint64_t line_num = 0;
  bool open = false;
  while (lr.next(line)) {
    ++line_num;
    split_tabs(line, f);
    if (f.size() != 2)  // lib/Twister.ml:103-104
      throw Error("Wrong_number_of_columns(" + std::to_string(line_num) + ", " + std::to_string(f.size()) + ", 2)");
    if (line_num == 1 && !f[0].empty()) throw Error("Header_expected(\"" + line + "\")");  // :106-107
    if (f[0].empty()) {  // :108-114 new header
      if (open) out.offsets.push_back(out.names.size());
      out.labels.push_back(strip_external_quotes_and_check(f[1]));
      open = true;
    } else {
      double v;
      if (!parse_float(f[1], &v)) throw Error("Float_expected(\"" + f[1] + "\")");  // :155-157
      out.names.push_back(f[0]);
      out.values.push_back(v);
    }
  }
  if (open) out.offsets.push_back(out.names.size());
}

void write_spectrum_body(FILE *f, const uint64_t *hash, const uint32_t *count, uint64_t n, int k) {
  // "%s\t%d\n" (KIH.to_hex k) f, bin/KPopCount.ml:46,60
  const int digits = (k + 1) / 2;
  char buf[64];
  static const char hx[] = "0123456789abcdef";
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t h = hash[i];
    for (int d = digits - 1; d >= 0; --d) {
      buf[d] = hx[h & 15];
      h >>= 4;
    }
    int len = digits;
    len += snprintf(buf + digits, sizeof(buf) - (size_t)digits, "\t%u\n", count[i]);
    fwrite(buf, 1, (size_t)len, f);
  }
}

void write_spectrum(FILE *f, const std::string &label, const uint64_t *hash, const uint32_t *count, uint64_t n, int k) {
  fprintf(f, "\t%s\n", label.c_str());  // bin/KPopCount.ml:34,45
  write_spectrum_body(f, hash, count, n, k);
}

// ------------------------------------------------------------------ tables
Table read_table(const std::string &path) {
  LineReader lr(path);
  Table t;
  std::string line;
  std::vector<std::string> f;
  if (!lr.next(line)) return t;  // empty file = empty matrix
  split_tabs(line, f);
  for (size_t i = 1; i < f.size(); ++i) t.col_names.push_back(strip_external_quotes_and_check(f[i]));
  uint64_t line_num = 1;
  while (lr.next(line)) {
    ++line_num;
    if (line.empty()) continue;
    split_tabs(line, f);
    if (f.size() != t.col_names.size() + 1)
      throw Error("table '" + path + "' line " + std::to_string(line_num) + ": " + std::to_string(f.size() - 1) +
                  " values for " + std::to_string(t.col_names.size()) + " columns");
    t.row_names.push_back(strip_external_quotes_and_check(f[0]));
    for (size_t i = 1; i < f.size(); ++i) {
      double v;
      if (!parse_float(f[i], &v)) throw Error("table '" + path + "' line " + std::to_string(line_num) + ": Float_expected(\"" + f[i] + "\")");
      t.data.push_back(v);
    }
  }
  return t;
}

void write_table(const std::string &path, const Table &t, int precision) {
  FILE *f = (path == "/dev/stdout") ? stdout : fopen(path.c_str(), "wb");
  if (!f) throw Error("cannot write '" + path + "': " + strerror(errno));
  std::string buf = "\"\"";
  for (const std::string &c : t.col_names) {
    buf += "\t\"";
    buf += c;
    buf += '"';
  }
  buf += '\n';
  fwrite(buf.data(), 1, buf.size(), f);
  const size_t nc = t.cols();
  char num[64];
  for (size_t r = 0; r < t.rows(); ++r) {
    buf.clear();
    buf += '"';
    buf += t.row_names[r];
    buf += '"';
    for (size_t c = 0; c < nc; ++c) {
      int len = snprintf(num, sizeof(num), "\t%.*g", precision, t.data[r * nc + c]);
      buf.append(num, (size_t)len);
    }
    buf += '\n';
    fwrite(buf.data(), 1, buf.size(), f);
  }
  if (f != stdout) fclose(f);
  else fflush(f);
}

void merge_rowwise(Table &into, const Table &add) {
  if (into.empty()) {
    into = add;
    return;
  }
  if (add.empty()) return;
  if (into.col_names != add.col_names) throw Error("Incompatible_geometries: column names differ");
  into.row_names.insert(into.row_names.end(), add.row_names.begin(), add.row_names.end());
  into.data.insert(into.data.end(), add.data.begin(), add.data.end());
}

}  // namespace kpop_host

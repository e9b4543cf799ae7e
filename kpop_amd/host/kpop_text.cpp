// kpop_text.cpp -- see kpop_text.h
#include "kpop_text.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

namespace kpop_host {

// ------------------------------------------------------------------ timing
void stage_mark(const char *tool, const char *stage) {
  static const bool on = getenv("KPOP_TIMING") != nullptr;
  if (!on) return;
  static double t0 = -1., last = 0.;
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  const double now = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
  if (t0 < 0.) t0 = last = now;
  fprintf(stderr, "[timing] %s: %-44s +%.3f s  (%.3f s)\n", tool, stage, now - last, now - t0);
  last = now;
}

// ------------------------------------------------------------------ large blocks
namespace {
const bool g_huge_pages = [] {
  const char *e = getenv("KPOP_HUGE_PAGES");
  return !(e && e[0] == '0');
}();
std::mutex g_huge_mutex;
std::vector<std::pair<void *, size_t>> g_huge_blocks;  // (address, mapped bytes) of the live ones: a handful at any time
}  // namespace

void *huge_block_alloc(size_t bytes) {
  if (!g_huge_pages) return nullptr;
  const size_t two_mb = 2u << 20, mapped = (bytes + two_mb - 1) / two_mb * two_mb + two_mb;
  char *raw = (char *)mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (raw == (char *)MAP_FAILED) return nullptr;
  char *p = (char *)(((uintptr_t)raw + two_mb - 1) & ~(uintptr_t)(two_mb - 1));
  if (p > raw) munmap(raw, (size_t)(p - raw));
  const size_t keep = (bytes + two_mb - 1) / two_mb * two_mb;
  if (p + keep < raw + mapped) munmap(p + keep, (size_t)(raw + mapped - (p + keep)));
#ifdef MADV_HUGEPAGE
  (void)madvise(p, keep, MADV_HUGEPAGE);
#endif
  std::lock_guard<std::mutex> lk(g_huge_mutex);
  g_huge_blocks.push_back({p, keep});
  return p;
}

bool huge_block_free(void *p, size_t) {
  size_t mapped = 0;
  {
    std::lock_guard<std::mutex> lk(g_huge_mutex);
    for (size_t i = 0; i < g_huge_blocks.size(); ++i)
      if (g_huge_blocks[i].first == p) {
        mapped = g_huge_blocks[i].second;
        g_huge_blocks[i] = g_huge_blocks.back();
        g_huge_blocks.pop_back();
        break;
      }
  }
  if (!mapped) return false;
  munmap(p, mapped);
  return true;
}

// ------------------------------------------------------------------ threads
void parallel_for(size_t n, size_t min_per_thread, const std::function<void(size_t, size_t)> &fn) {
  unsigned t = std::thread::hardware_concurrency();
  if (const char *e = getenv("KPOP_HOST_THREADS")) t = (unsigned)atoi(e);
  t = std::max(1u, std::min(t, 32u));
  t = (unsigned)std::max<size_t>(1, std::min<size_t>(t, n / std::max<size_t>(1, min_per_thread) + 1));
  if (t == 1 || n == 0) {
    fn(0, n);
    return;
  }
  std::vector<std::thread> pool;
  std::vector<std::exception_ptr> err(t);
  auto run = [&](unsigned i) {
    try {
      fn(n * i / t, n * (i + 1) / t);
    } catch (...) {
      err[i] = std::current_exception();
    }
  };
  for (unsigned i = 1; i < t; ++i) pool.emplace_back(run, i);
  run(0);
  for (std::thread &th : pool) th.join();
  for (auto &e : err)
    if (e) std::rethrow_exception(e);
}

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
  std::string s;
  append_g(s, x, precision);
  return s;
}

void append_g(std::string &out, double x, int precision) {
  char buf[512];
  if (precision >= 1 && precision <= 17) {
    const std::to_chars_result r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::general, precision);
    if (r.ec == std::errc()) {
      out.append(buf, (size_t)(r.ptr - buf));
      return;
    }
  }
  const int len = snprintf(buf, sizeof(buf), "%.*g", precision, x);
  if (len > 0) out.append(buf, std::min((size_t)len, sizeof(buf) - 1));
}

void write_rows_parallel(FILE *f, const std::string &path, size_t n, size_t reserve_per_row,
                         const std::function<void(size_t row, std::string &out)> &fmt) {
  // about 64 MB of text per slab and not less than 64 KB of it per thread, whatever a row is (a read's spectrum, a genome's)
  const size_t row_bytes = std::max<size_t>(1, reserve_per_row);
  const size_t slab = std::max<size_t>(1, std::min<size_t>(1u << 18, (64u << 20) / row_bytes));
  const size_t min_rows = std::max<size_t>(1, (64u << 10) / row_bytes);
  struct Slab {
    std::vector<std::string> text;
    std::vector<unsigned> order;
  };
  std::mutex m;
  std::condition_variable cv;
  std::deque<Slab> ready;
  bool done = false;
  std::atomic<bool> failed{false};
  std::thread writer([&] {
    for (;;) {
      Slab sl;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done || !ready.empty(); });
        if (ready.empty()) return;
        sl = std::move(ready.front());
        ready.pop_front();
      }
      cv.notify_all();
      for (unsigned i : sl.order)
        if (!failed && fwrite(sl.text[i].data(), 1, sl.text[i].size(), f) != sl.text[i].size()) failed = true;
    }
  });
  std::exception_ptr error;
  try {
    for (size_t s0 = 0; s0 < n && !failed; s0 += slab) {
      const size_t s1 = std::min(n, s0 + slab);
      Slab sl;
      sl.text.resize(64);
      std::vector<std::pair<size_t, size_t>> span(64, {0, 0});
      std::atomic<unsigned> next{0};
      parallel_for(s1 - s0, min_rows, [&](size_t lo, size_t hi) {
        const unsigned me = next++;
        if (me >= sl.text.size()) throw Error("write_rows_parallel: more pieces than expected");
        span[me] = {lo, hi};
        std::string &o = sl.text[me];
        o.reserve((hi - lo) * reserve_per_row);
        for (size_t r = s0 + lo; r < s0 + hi; ++r) fmt(r, o);
      });
      for (unsigned i = 0; i < next && i < sl.text.size(); ++i) sl.order.push_back(i);
      std::sort(sl.order.begin(), sl.order.end(), [&](unsigned a, unsigned b) { return span[a].first < span[b].first; });
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return ready.size() < 2; });  // at most two slabs of text waiting
        ready.push_back(std::move(sl));
      }
      cv.notify_all();
    }
  } catch (...) {
    error = std::current_exception();
  }
  {
    std::lock_guard<std::mutex> lk(m);
    done = true;
  }
  cv.notify_all();
  writer.join();
  if (error) std::rethrow_exception(error);
  if (failed) throw Error("cannot write '" + path + "'");
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

// ------------------------------------------------------------------ parallel spectra parser / writer
namespace {

// KPOP_HOST_THREADS / KPOP_HOST_CHUNK (work items per thread) override the defaults; the tests use them to force many
// small chunks through the cut-and-merge logic
unsigned pick_threads(unsigned asked, size_t work_items, size_t per_thread) {
  unsigned t = asked ? asked : std::thread::hardware_concurrency();
  if (const char *e = getenv("KPOP_HOST_THREADS")) t = (unsigned)atoi(e);
  if (const char *e = getenv("KPOP_HOST_CHUNK")) per_thread = (size_t)std::max(1, atoi(e));
  if (t == 0) t = 1;
  t = std::min<unsigned>(t, 32);
  return (unsigned)std::max<size_t>(1, std::min<size_t>(t, work_items / per_thread + 1));
}

std::vector<char> slurp(const std::string &path) {
  FILE *f;
  bool own = false;
  if (path == "/dev/stdin" || path == "-") f = stdin;
  else {
    f = fopen(path.c_str(), "rb");
    own = true;
  }
  if (!f) throw Error("cannot open '" + path + "': " + strerror(errno));
  std::vector<char> buf;
  size_t cap = 1 << 22, len = 0;
  buf.resize(cap);
  for (;;) {
    if (len == cap) {
      cap *= 2;
      buf.resize(cap);
    }
    const size_t got = fread(buf.data() + len, 1, cap - len, f);
    if (got == 0) break;
    len += got;
  }
  if (own) fclose(f);
  buf.resize(len);
  return buf;
}

struct ChunkResult {
  std::vector<uint64_t> hash;
  std::vector<double> values;
  std::vector<std::pair<uint64_t, std::string>> headers;  // (number of value lines of this chunk before it, label)
  uint64_t n_lines = 0;
  bool failed = false;
  uint64_t error_line = 0;  // 1-based within the chunk
  std::string error;
  bool first_line_is_header = true;
  std::string first_line;
  bool plain = true;  // every data line was name_len LOWERCASE hexadecimal digits, a tab, 1-15 decimal digits, a newline; no CR anywhere
};

struct HexTable {
  uint8_t v[256];
  HexTable() {
    for (int c = 0; c < 256; ++c) v[c] = 0x80;
    for (int c = '0'; c <= '9'; ++c) v[c] = (uint8_t)(c - '0');
    for (int c = 'a'; c <= 'f'; ++c) v[c] = (uint8_t)(c - 'a' + 10);
    for (int c = 'A'; c <= 'F'; ++c) v[c] = (uint8_t)(c - 'A' + 10);
  }
};
const HexTable kHex;

void parse_chunk(const char *s, const char *e, size_t name_len, uint64_t absent, ChunkResult &r, const KmerLookup *lk) {
  r.hash.reserve((size_t)(e - s) / 9 + 16);
  r.values.reserve((size_t)(e - s) / 9 + 16);
  const bool fast_names = name_len >= 1 && name_len <= 16 && !(lk && lk->opaque);
  const bool strict = lk != nullptr;  // names are strings: an uppercase digit makes another name than the columns' (all lowercase)
  while (s < e) {
    // The line KPopCount writes -- name_len hexadecimal digits, a tab, up to 15 decimal digits, a newline -- is taken in
    // one pass; anything else about a line (and a chunk's first line, which has bookkeeping of its own) goes the general way
    // below, which gives the same answer for these too.
    if (fast_names && r.n_lines > 0 && (size_t)(e - s) >= name_len + 3 && s[name_len] == '\t') {
      uint64_t v = 0;
      unsigned bad = 0, upper = 0;
      for (size_t i = 0; i < name_len; ++i) {
        const uint8_t d = kHex.v[(uint8_t)s[i]];
        bad |= d;
        upper |= (unsigned)((uint8_t)(s[i] - 'A') < 6u);
        v = (v << 4) | (uint64_t)(d & 15);
      }
      if (upper) r.plain = false;
      if (!(bad & 0x80) && !(upper && strict)) {
        const char *p = s + name_len + 1;
        uint64_t iv = 0;
        int nd = 0;
        while (p < e && (unsigned)(*p - '0') <= 9u && nd < 16) {
          iv = iv * 10 + (uint64_t)(*p - '0');
          ++p;
          ++nd;
        }
        if (nd >= 1 && nd <= 15 && p < e && *p == '\n') {
          r.hash.push_back(v);
          r.values.push_back((double)iv);
          ++r.n_lines;
          s = p + 1;
          continue;
        }
      }
    }
    const char *nl = (const char *)memchr(s, '\n', (size_t)(e - s));
    const char *le = nl ? nl : e;
    const char *next = nl ? nl + 1 : e;
    if (le > s && le[-1] == '\r') {
      --le;
      r.plain = false;
    }
    if (!nl) r.plain = false;  // (a last line without its newline)
    ++r.n_lines;
    const char *tab = (const char *)memchr(s, '\t', (size_t)(le - s));
    if (!tab || memchr(tab + 1, '\t', (size_t)(le - tab - 1))) {  // lib/Twister.ml:103-104
      size_t n = 1;
      for (const char *p = s; p < le; ++p) n += *p == '\t';
      r.failed = true;
      r.error_line = r.n_lines;
      r.error = ", " + std::to_string(n) + ", 2)";
      return;
    }
    if (r.n_lines == 1 && tab != s) {
      r.first_line_is_header = false;
      r.first_line.assign(s, (size_t)(le - s));
    }
    if (tab == s) {  // header
      try {
        r.headers.emplace_back(r.hash.size(), strip_external_quotes_and_check(std::string(tab + 1, (size_t)(le - tab - 1))));
      } catch (const Error &ex) {
        r.failed = true;
        r.error_line = 0;
        r.error = ex.what();
        return;
      }
    } else {
      // (a data line here was not the plain form, except a chunk's first line, which always comes this way: checked below)
      bool plain_line = (size_t)(tab - s) == name_len && name_len >= 1 && name_len <= 16;
      for (const char *q = s; plain_line && q < tab; ++q) plain_line = (*q >= '0' && *q <= '9') || (*q >= 'a' && *q <= 'f');
      {
        const size_t vl0 = (size_t)(le - (tab + 1));
        plain_line = plain_line && vl0 >= 1 && vl0 <= 15;
        for (const char *q = tab + 1; plain_line && q < le; ++q) plain_line = *q >= '0' && *q <= '9';
      }
      if (!plain_line) r.plain = false;
      uint64_t h = absent;
      if (lk && lk->opaque) {
        const auto it = lk->index->find(std::string_view(s, (size_t)(tab - s)));
        if (it != lk->index->end()) h = it->second;
      } else if ((size_t)(tab - s) == name_len && name_len <= 16) {
        uint64_t v = 0;
        bool ok = name_len > 0;
        for (const char *p = s; p < tab; ++p) {
          const char c = *p;
          int d;
          if (c >= '0' && c <= '9') d = c - '0';
          else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
          else if (c >= 'A' && c <= 'F' && !strict) d = c - 'A' + 10;
          else {
            ok = false;
            break;
          }
          v = (v << 4) | (uint64_t)d;
        }
        if (ok) h = v;
      }
      // with the columns known, a line whose name is no column is dropped before its value is looked at (lib/Twister.ml:151-169)
      const bool is_column = !lk || (h != absent && (lk->opaque || !lk->columns || std::binary_search(lk->columns->begin(), lk->columns->end(), h)));
      if (!is_column) {
        r.hash.push_back(absent);
        r.values.push_back(0.);
        s = next;
        continue;
      }
      const char *vs = tab + 1;
      const size_t vl = (size_t)(le - vs);
      double val = 0.;
      bool digits = vl > 0 && vl <= 15;
      uint64_t iv = 0;
      for (size_t i = 0; digits && i < vl; ++i) {
        if (vs[i] < '0' || vs[i] > '9') digits = false;
        else iv = iv * 10 + (uint64_t)(vs[i] - '0');
      }
      if (digits) val = (double)iv;  // what KPopCount writes ("%d"): exact
      else if (!parse_float(std::string(vs, vl), &val)) {
        r.failed = true;
        r.error_line = 0;
        r.error = "Float_expected(\"" + std::string(vs, vl) + "\")";  // :155-157
        return;
      }
      r.hash.push_back(h);
      r.values.push_back(val);
    }
    s = next;
  }
}

}  // namespace

static void parse_spectra_buffer(const char *base, size_t size, size_t name_len, uint64_t absent, HashedSpectra &out, unsigned threads,
                                 bool first_block = true, uint64_t lines_before = 0, uint64_t *n_lines = nullptr, bool *plain = nullptr,
                                 const KmerLookup *lk = nullptr);

void read_spectra_hashed(const std::string &path, size_t name_len, uint64_t absent, HashedSpectra &out, unsigned threads) {
  const std::vector<char> buf = slurp(path);
  parse_spectra_buffer(buf.data(), buf.size(), name_len, absent, out, threads);
}

void read_spectra_hashed_fd(int fd, const char *head, size_t head_len, size_t name_len, uint64_t absent, HashedSpectra &out,
                            unsigned threads) {
  std::vector<char> buf;
  size_t cap = 1 << 22, len = head_len;
  buf.resize(cap);
  if (head_len) memcpy(buf.data(), head, head_len);
  for (;;) {
    if (len == cap) {
      cap *= 2;
      buf.resize(cap);
    }
    const ssize_t got = read(fd, buf.data() + len, cap - len);
    if (got < 0) {
      if (errno == EINTR) continue;
      throw Error(std::string("read failed: ") + strerror(errno));
    }
    if (got == 0) break;
    len += (size_t)got;
  }
  buf.resize(len);
  parse_spectra_buffer(buf.data(), buf.size(), name_len, absent, out, threads);
}

static void parse_spectra_buffer(const char *base, size_t size, size_t name_len, uint64_t absent, HashedSpectra &out, unsigned threads,
                                 bool first_block, uint64_t lines_before_block, uint64_t *n_lines, bool *plain, const KmerLookup *lk) {
  const unsigned T = pick_threads(threads, size, 2u << 20);
  std::vector<size_t> cut(T + 1, size);
  cut[0] = 0;
  for (unsigned t = 1; t < T; ++t) {
    size_t at = std::max(cut[t - 1], size / T * t);
    const char *nl = at < size ? (const char *)memchr(base + at, '\n', size - at) : nullptr;
    cut[t] = nl ? (size_t)(nl - base) + 1 : size;
  }
  std::vector<ChunkResult> res(T);
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < T; ++t)
    pool.emplace_back([&, t] { parse_chunk(base + cut[t], base + cut[t + 1], name_len, absent, res[t], lk); });
  parse_chunk(base + cut[0], base + cut[1], name_len, absent, res[0], lk);
  for (std::thread &th : pool) th.join();
  // errors in file order, with the sequential parser's precedence on line 1 (column count, then Header_expected)
  uint64_t lines_before = lines_before_block;
  for (unsigned t = 0; t < T; ++t) {
    const ChunkResult &r = res[t];
    const bool col_error_on_line_1 = r.failed && r.error_line == 1 && t == 0;
    if (t == 0 && first_block && r.n_lines >= 1 && !r.first_line_is_header && !col_error_on_line_1)
      throw Error("Header_expected(\"" + r.first_line + "\")");  // lib/Twister.ml:106-107
    if (r.failed) {
      if (r.error_line) throw Error("Wrong_number_of_columns(" + std::to_string(lines_before + r.error_line) + r.error);
      throw Error(r.error);
    }
    lines_before += r.n_lines;
  }
  if (n_lines) *n_lines = lines_before - lines_before_block;
  if (plain) {
    *plain = true;
    for (const ChunkResult &r : res) *plain = *plain && r.plain;
  }
  size_t total = out.hash.size();
  for (const ChunkResult &r : res) total += r.hash.size();
  const size_t start = out.hash.size();
  out.hash.resize(total);
  out.values.resize(total);
  size_t at = start;
  bool open = false;
  for (const ChunkResult &r : res) {
    for (const auto &h : r.headers) {
      if (open) out.offsets.push_back(at + h.first);
      out.labels.push_back(h.second);
      open = true;
    }
    at += r.hash.size();
  }
  if (open) out.offsets.push_back(at);
  std::vector<size_t> place(T + 1, start);
  for (unsigned t = 0; t < T; ++t) place[t + 1] = place[t] + res[t].hash.size();
  parallel_for(T, 1, [&](size_t lo, size_t hi) {
    for (size_t t = lo; t < hi; ++t)
      if (!res[t].hash.empty()) {
        memcpy(&out.hash[place[t]], res[t].hash.data(), res[t].hash.size() * 8);
        memcpy(&out.values[place[t]], res[t].values.data(), res[t].values.size() * 8);
      }
  });
}

void parse_spectra_block(const char *data, size_t size, size_t name_len, uint64_t absent, bool first_block, uint64_t lines_before,
                         HashedSpectra &out, uint64_t *n_lines, unsigned threads, bool *plain, const KmerLookup *lookup) {
  out = HashedSpectra();
  parse_spectra_buffer(data, size, name_len, absent, out, threads, first_block, lines_before, n_lines, plain, lookup);
}

SpectraTextStream::SpectraTextStream(int fd, const char *head, size_t head_len, size_t block_bytes)
    : fd_(fd), block_bytes_(std::max<size_t>(block_bytes, 2)) {
  if (const char *e = getenv("KPOP_TEXT_BLOCK")) block_bytes_ = (size_t)std::max(2, atoi(e));  // tests: many small blocks
  if (head_len) carry_.assign(head, head + head_len);
}

bool SpectraTextStream::next(TextBlock &block) {
  block.clear();
  block.swap(carry_);
  size_t target = block_bytes_;
  for (;;) {
    while (!eof_ && block.size() < target) {
      const size_t at = block.size();
      block.resize(target);
      size_t len = at;
      while (len < target) {
        const ssize_t got = read(fd_, block.data() + len, target - len);
        if (got < 0) {
          if (errno == EINTR) continue;
          throw Error(std::string("read failed: ") + strerror(errno));
        }
        if (got == 0) {
          eof_ = true;
          break;
        }
        len += (size_t)got;
      }
      block.resize(len);
    }
    if (eof_) return !block.empty();
    // the last header line that is not the block's first line: the block ends before it
    const char *base = block.data();
    size_t limit = block.size() - 1;  // bytes in which to look for a newline (one that has a byte after it)
    while (limit > 0) {
      const char *nl = (const char *)memrchr(base, '\n', limit);
      if (!nl) break;
      const size_t line = (size_t)(nl - base) + 1;  // where the line after it begins
      if (base[line] == '\t') {
        carry_.assign(base + line, base + block.size());
        block.resize(line);
        return true;
      }
      limit = (size_t)(nl - base);
      if (block.size() - limit > (16u << 20)) break;  // (a last spectrum of more than 16 MB: take more input instead)
    }
    target = block.size() * 2;  // one spectrum longer than the block: keep reading
  }
}

void write_spectra_parallel(FILE *f, const std::vector<std::string> &labels, const uint64_t *hash, const uint32_t *count,
                            const uint64_t *offsets, int digits, unsigned threads) {
  (void)threads;  // (the host threads of parallel_for)
  const size_t n = labels.size();
  if (n == 0) return;
  const size_t per_row = (size_t)((offsets[n] - offsets[0]) / n + 1) * (size_t)(digits + 8) + 24;
  write_rows_parallel(f, "the spectra", n, per_row, [&](size_t r, std::string &o) {
    static const char hx[] = "0123456789abcdef";
    char buf[40];
    o.push_back('\t');
    o.append(labels[r]);
    o.push_back('\n');
    for (uint64_t i = offsets[r]; i < offsets[r + 1]; ++i) {
      uint64_t h = hash[i];
      for (int d = digits - 1; d >= 0; --d) {
        buf[d] = hx[h & 15];
        h >>= 4;
      }
      int len = digits;
      buf[len++] = '\t';
      char tmp[12];
      int tl = 0;
      uint32_t c = count[i];
      do {
        tmp[tl++] = (char)('0' + c % 10);
        c /= 10;
      } while (c);
      while (tl) buf[len++] = tmp[--tl];
      buf[len++] = '\n';
      o.append(buf, (size_t)len);
    }
  });
}

void write_spectrum_body(FILE *f, const uint64_t *hash, const uint32_t *count, uint64_t n, int digits) {
  // "%s\t%d\n" (KIH.to_hex k) f, bin/KPopCount.ml:46,60
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

void write_spectrum(FILE *f, const std::string &label, const uint64_t *hash, const uint32_t *count, uint64_t n, int digits) {
  fprintf(f, "\t%s\n", label.c_str());  // bin/KPopCount.ml:34,45
  write_spectrum_body(f, hash, count, n, digits);
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
  try {
    write_rows_parallel(f, path, t.rows(), 8 + nc * 20, [&](size_t r, std::string &o) {
      o += '"';
      o += t.row_names[r];
      o += '"';
      for (size_t c = 0; c < nc; ++c) {
        o += '\t';
        append_g(o, t.data[r * nc + c], precision);
      }
      o += '\n';
    });
  } catch (...) {
    if (f != stdout) fclose(f);
    throw;
  }
  if (f != stdout) fclose(f);
  else fflush(f);
}

std::vector<uint32_t> order_rows_by_label(const std::vector<std::string> &labels, size_t n_existing) {
  const size_t n = labels.size();
  if (n > 0xFFFFFFFFull) throw Error("more than 2^32 rows");
  std::vector<uint32_t> idx(n);
  for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
  auto less = [&](uint32_t a, uint32_t b) {
    const int c = labels[a].compare(labels[b]);  // char_traits<char>::compare: bytewise, as OCaml's String.compare
    return c < 0 || (c == 0 && a < b);
  };
  // sorted runs by the threads, then pairwise merges
  unsigned T = std::thread::hardware_concurrency();
  if (const char *e = getenv("KPOP_HOST_THREADS")) T = (unsigned)atoi(e);
  T = std::max(1u, std::min(T, 16u));
  while (T > 1 && n / T < 8192) T /= 2;
  unsigned P = 1;
  while (P * 2 <= T) P *= 2;  // a power of two of runs
  std::vector<size_t> edge(P + 1);
  for (unsigned i = 0; i <= P; ++i) edge[i] = n * i / P;
  {
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < P; ++i) pool.emplace_back([&, i] { std::sort(idx.begin() + edge[i], idx.begin() + edge[i + 1], less); });
    std::sort(idx.begin() + edge[0], idx.begin() + edge[1], less);
    for (std::thread &th : pool) th.join();
  }
  for (unsigned w = 1; w < P; w *= 2) {
    std::vector<std::thread> pool;
    for (unsigned i = 0; i + w < P + 1 && i < P; i += 2 * w) {
      const size_t a = edge[i], m = edge[std::min(i + w, P)], b = edge[std::min(i + 2 * w, P)];
      if (m >= b) continue;
      pool.emplace_back([&, a, m, b] { std::inplace_merge(idx.begin() + a, idx.begin() + m, idx.begin() + b, less); });
    }
    for (std::thread &th : pool) th.join();
  }
  // groups of equal labels
  std::vector<uint32_t> out;
  out.reserve(n);
  uint32_t first_dup = 0xFFFFFFFFu;
  for (size_t i = 0; i < n;) {
    size_t j = i + 1;
    while (j < n && labels[idx[j]] == labels[idx[i]]) ++j;
    // members ascend by arrival number; a new row (>= n_existing) that is not the first of its group is a duplicate
    uint32_t keep = idx[i];
    for (size_t q = i + 1; q < j; ++q) {
      if (idx[q] >= n_existing) {
        first_dup = std::min(first_dup, idx[q]);
        break;
      }
      keep = idx[q];  // among existing rows the last one wins
    }
    out.push_back(keep);
    i = j;
  }
  if (first_dup != 0xFFFFFFFFu) throw Error("Duplicate_label(\"" + labels[first_dup] + "\")");  // lib/Twister.ml:195
  return out;
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

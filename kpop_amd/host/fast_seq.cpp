// fast_seq.cpp -- see fast_seq.h
#include "fast_seq.h"

#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <thread>

namespace kpop_host {

void FlatBatch::append(const FlatBatch &o) {
  bases.insert(bases.end(), o.bases.begin(), o.bases.end());
  lens.insert(lens.end(), o.lens.begin(), o.lens.end());
  tags.insert(tags.end(), o.tags.begin(), o.tags.end());
  tag_lens.insert(tag_lens.end(), o.tag_lens.begin(), o.tag_lens.end());
}

namespace {

// dnaize ~keep_lowercase:false ~keep_dashes:false (and proteinize, taken as its twin): upper-case, dashes and
// blanks dropped, everything else stays where it is (and breaks the k-mer window in the kernels)
struct LintTable {
  uint8_t map[256], keep[256];
  LintTable() {
    for (int c = 0; c < 256; ++c) {
      map[c] = (uint8_t)((c >= 'a' && c <= 'z') ? c - 32 : c);
      keep[c] = (c == '-' || c == ' ' || c == '\t') ? 0 : 1;
    }
  }
};
const LintTable kLint;

inline const char *strip_cr(const char *s, const char *e) {  // LineReader::next strips every trailing \r
  while (e > s && e[-1] == '\r') --e;
  return e;
}

inline void lint_line(const char *s, const char *e, std::vector<uint8_t> &out) {
  const size_t at = out.size();
  out.resize(at + (size_t)(e - s));
  uint8_t *o = out.data() + at;
  for (const char *p = s; p < e; ++p) {
    const uint8_t c = (uint8_t)*p;
    *o = kLint.map[c];
    o += kLint.keep[c];
  }
  out.resize((size_t)(o - out.data()));
}

// records of [s, e): s sits on a '>' (or on blank lines before the first one of the file)
void parse_fasta_range(const char *s, const char *e, const std::string &path, FlatBatch &out) {
  out.bases.reserve((size_t)(e - s));
  const char *p = s;
  while (p < e) {  // blank lines before a header
    const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
    const char *le = strip_cr(p, nl ? nl : e);
    if (le != p) break;
    p = nl ? nl + 1 : e;
  }
  while (p < e) {
    if (*p != '>') throw Error("FASTA file '" + path + "': expected '>' at the start of a record");
    const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
    const char *le = strip_cr(p, nl ? nl : e);
    out.tags.insert(out.tags.end(), p + 1, le);
    out.tag_lens.push_back((uint32_t)(le - (p + 1)));
    p = nl ? nl + 1 : e;
    const size_t start = out.bases.size();
    while (p < e && *p != '>') {
      nl = (const char *)memchr(p, '\n', (size_t)(e - p));
      le = strip_cr(p, nl ? nl : e);
      lint_line(p, le, out.bases);
      p = nl ? nl + 1 : e;
    }
    const size_t n = out.bases.size() - start;
    if (n > 0xFFFFFFFFull) throw Error("FASTA file '" + path + "': a sequence longer than 2^32 bases");
    out.lens.push_back((uint32_t)n);
  }
}

unsigned seq_threads(unsigned asked, size_t bytes) {
  unsigned t = asked ? asked : std::thread::hardware_concurrency();
  if (const char *e = getenv("KPOP_HOST_THREADS")) t = (unsigned)atoi(e);
  size_t per = 4u << 20;
  if (const char *e = getenv("KPOP_HOST_CHUNK")) per = (size_t)std::max(1, atoi(e));
  t = std::max(1u, std::min(t, 32u));
  return (unsigned)std::max<size_t>(1, std::min<size_t>(t, bytes / per + 1));
}

}  // namespace

struct FastSeqReader::Impl {
  int fd = -1;
  bool own = false, eof = false;
  SeqFormat fmt;
  std::string path;
  std::vector<char> buf;  // text not yet handed out
  size_t len = 0;         // bytes of buf in use

  void fill(size_t target) {
    if (buf.size() < target) buf.resize(target);
    while (!eof && len < target) {
      const ssize_t got = read(fd, buf.data() + len, target - len);
      if (got < 0) {
        if (errno == EINTR) continue;
        throw Error("cannot read '" + path + "': " + strerror(errno));
      }
      if (got == 0) eof = true;
      else len += (size_t)got;
    }
  }
  void consume(size_t n) {
    if (n < len) memmove(buf.data(), buf.data() + n, len - n);
    len -= n;
  }
};

FastSeqReader::FastSeqReader(const std::string &path, SeqFormat fmt) : p_(new Impl) {
  p_->fmt = fmt;
  p_->path = path;
  if (path == "/dev/stdin" || path == "-") p_->fd = 0;
  else {
    p_->fd = open(path.c_str(), O_RDONLY);
    p_->own = true;
  }
  if (p_->fd < 0) {
    const std::string msg = "cannot open '" + path + "': " + strerror(errno);
    delete p_;
    throw Error(msg);
  }
}

FastSeqReader::~FastSeqReader() {
  if (p_->own && p_->fd >= 0) close(p_->fd);
  delete p_;
}

bool FastSeqReader::next(FlatBatch &out, size_t block_bytes, unsigned threads) {
  Impl &I = *p_;
  out.clear();
  if (const char *e = getenv("KPOP_SEQ_BLOCK")) block_bytes = (size_t)std::max(64, atoi(e));  // tests: many small blocks
  size_t target = std::max<size_t>(block_bytes, 64);
  if (I.fmt == SeqFormat::FASTA) {
    size_t cut = 0;
    for (;;) {
      I.fill(target);
      if (I.len == 0) return false;
      if (I.eof) {
        cut = I.len;
        break;
      }
      // the last record start: whatever follows it may be incomplete and stays for the next block
      const char *b = I.buf.data();
      size_t at = I.len;
      cut = 0;
      while (at > 0) {
        const char *g = (const char *)memrchr(b, '>', at);
        if (!g) break;
        const size_t pos = (size_t)(g - b);
        if (pos > 0 && b[pos - 1] == '\n') {
          cut = pos;
          break;
        }
        at = pos;
      }
      if (cut > 0) break;
      target *= 2;  // one record larger than the block: take more of the file
    }
    const char *b = I.buf.data();
    const unsigned T = seq_threads(threads, cut);
    std::vector<size_t> edge(T + 1, cut);
    edge[0] = 0;
    for (unsigned t = 1; t < T; ++t) {
      size_t at = std::max(edge[t - 1], cut / T * t);
      size_t found = cut;
      while (at < cut) {
        const char *g = (const char *)memchr(b + at, '>', cut - at);
        if (!g) break;
        const size_t pos = (size_t)(g - b);
        if (pos > 0 && b[pos - 1] == '\n') {
          found = pos;
          break;
        }
        at = pos + 1;
      }
      edge[t] = found;
    }
    if (T == 1) {
      parse_fasta_range(b, b + cut, I.path, out);
    } else {
      std::vector<FlatBatch> part(T);
      std::vector<std::string> err(T);
      auto work = [&](unsigned t) {
        try {
          parse_fasta_range(b + edge[t], b + edge[t + 1], I.path, part[t]);
        } catch (const std::exception &e) {
          err[t] = e.what();
        }
      };
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < T; ++t) pool.emplace_back(work, t);
      work(0);
      for (std::thread &th : pool) th.join();
      for (unsigned t = 0; t < T; ++t)
        if (!err[t].empty()) throw Error(err[t]);
      size_t nb = 0, nr = 0, nt = 0;
      for (const FlatBatch &p : part) {
        nb += p.bases.size();
        nr += p.lens.size();
        nt += p.tags.size();
      }
      out.bases.resize(nb);
      out.lens.resize(nr);
      out.tags.resize(nt);
      out.tag_lens.resize(nr);
      nb = nr = nt = 0;
      for (const FlatBatch &p : part) {
        if (!p.bases.empty()) memcpy(out.bases.data() + nb, p.bases.data(), p.bases.size());
        if (!p.lens.empty()) {
          memcpy(out.lens.data() + nr, p.lens.data(), p.lens.size() * 4);
          memcpy(out.tag_lens.data() + nr, p.tag_lens.data(), p.tag_lens.size() * 4);
        }
        if (!p.tags.empty()) memcpy(out.tags.data() + nt, p.tags.data(), p.tags.size());
        nb += p.bases.size();
        nr += p.lens.size();
        nt += p.tags.size();
      }
    }
    I.consume(cut);
    if (out.size() == 0) return next(out, block_bytes, threads);  // a block of blank lines
    return true;
  }
  // FASTQ: four lines per record.  A '@' can also start a quality line, so record starts cannot be found from the middle
  // of a block: one sequential pass finds the lines of every whole record (and raises what the line-by-line reader
  // raises, in its order), then the threads lint the sequences and copy the names, as for FASTA.
  struct Rec {
    size_t tag_b, tag_e, seq_b, seq_e;
  };
  for (;;) {
    I.fill(target);
    if (I.len == 0) return false;
    const char *b = I.buf.data(), *e = b + I.len, *p = b;
    std::vector<Rec> recs;
    recs.reserve(I.len / 200 + 16);
    size_t done = 0;
    for (;;) {
      const char *q = p;
      const char *line[4], *lend[4];
      int have = 0;
      while (q < e) {  // blank lines before a record
        const char *nl = (const char *)memchr(q, '\n', (size_t)(e - q));
        if (!nl && !I.eof) break;
        const char *le = strip_cr(q, nl ? nl : e);
        if (le != q) break;
        q = nl ? nl + 1 : e;
      }
      const char *rec = q;
      for (; have < 4 && q < e; ++have) {
        const char *nl = (const char *)memchr(q, '\n', (size_t)(e - q));
        if (!nl && !I.eof) break;  // the line may go on in the next block
        line[have] = q;
        lend[have] = strip_cr(q, nl ? nl : e);
        q = nl ? nl + 1 : e;
      }
      if (have == 0 && (rec >= e)) {
        done = (size_t)(rec - b);
        break;
      }
      if (have < 4) {
        if (I.eof) {
          if (*line[0] != '@') throw Error("FASTQ file '" + I.path + "': expected '@' at the start of a record");
          throw Error("FASTQ file '" + I.path + "': truncated record '" + std::string(line[0] + 1, lend[0]) + "'");
        }
        done = (size_t)(rec - b);
        break;
      }
      if (*line[0] != '@') throw Error("FASTQ file '" + I.path + "': expected '@' at the start of a record");
      if (lend[2] == line[2] || *line[2] != '+')
        throw Error("FASTQ file '" + I.path + "': truncated record '" + std::string(line[0] + 1, lend[0]) + "'");
      recs.push_back({(size_t)(line[0] + 1 - b), (size_t)(lend[0] - b), (size_t)(line[1] - b), (size_t)(lend[1] - b)});
      p = q;
      done = (size_t)(p - b);
    }
    if (!recs.empty() || I.eof) {
      const size_t n = recs.size();
      const unsigned T = n ? seq_threads(threads, done) : 1;
      std::vector<FlatBatch> part(T);
      std::vector<std::string> err(T);
      auto work = [&](unsigned t) {
        try {
          FlatBatch &o = part[t];
          const size_t lo = n * t / T, hi = n * (t + 1) / T;
          if (hi > lo) o.bases.reserve(recs[hi - 1].seq_e - recs[lo].seq_b);
          for (size_t r = lo; r < hi; ++r) {
            const Rec &R = recs[r];
            o.tags.insert(o.tags.end(), b + R.tag_b, b + R.tag_e);
            o.tag_lens.push_back((uint32_t)(R.tag_e - R.tag_b));
            const size_t start = o.bases.size();
            lint_line(b + R.seq_b, b + R.seq_e, o.bases);
            o.lens.push_back((uint32_t)(o.bases.size() - start));
          }
        } catch (const std::exception &ex) {
          err[t] = ex.what();
        }
      };
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < T; ++t) pool.emplace_back(work, t);
      work(0);
      for (std::thread &th : pool) th.join();
      for (unsigned t = 0; t < T; ++t)
        if (!err[t].empty()) throw Error(err[t]);
      if (T == 1) out = std::move(part[0]);
      else {
        size_t nb = 0, nr = 0, nt = 0;
        for (const FlatBatch &q : part) {
          nb += q.bases.size();
          nr += q.lens.size();
          nt += q.tags.size();
        }
        out.bases.resize(nb);
        out.lens.resize(nr);
        out.tags.resize(nt);
        out.tag_lens.resize(nr);
        nb = nr = nt = 0;
        for (const FlatBatch &q : part) {
          if (!q.bases.empty()) memcpy(out.bases.data() + nb, q.bases.data(), q.bases.size());
          if (!q.lens.empty()) {
            memcpy(out.lens.data() + nr, q.lens.data(), q.lens.size() * 4);
            memcpy(out.tag_lens.data() + nr, q.tag_lens.data(), q.tag_lens.size() * 4);
          }
          if (!q.tags.empty()) memcpy(out.tags.data() + nt, q.tags.data(), q.tags.size());
          nb += q.bases.size();
          nr += q.lens.size();
          nt += q.tags.size();
        }
      }
      I.consume(done);
      if (out.size() == 0) {
        if (I.len == 0 || I.eof) return false;
      }
      return out.size() > 0;
    }
    target *= 2;  // not even one whole record in the block
  }
}

// ------------------------------------------------------------------ the reads stream
void write_all(int fd, const void *p, size_t n) {
  const char *c = reinterpret_cast<const char *>(p);
  while (n) {
    const ssize_t w = write(fd, c, n);
    if (w < 0) {
      if (errno == EINTR) continue;
      throw Error(std::string("write failed: ") + strerror(errno));
    }
    c += w;
    n -= (size_t)w;
  }
}

bool read_all(int fd, void *p, size_t n) {
  char *c = reinterpret_cast<char *>(p);
  size_t got = 0;
  while (got < n) {
    const ssize_t r = read(fd, c + got, n - got);
    if (r < 0) {
      if (errno == EINTR) continue;
      throw Error(std::string("read failed: ") + strerror(errno));
    }
    if (r == 0) {
      if (got == 0) return false;
      throw Error("reads stream: truncated");
    }
    got += (size_t)r;
  }
  return true;
}

void write_stream_header(int fd, const ReadStreamHeader &h) {
  char b[16];
  memcpy(b, kReadStreamMagic, 8);
  memcpy(b + 8, &h.k, 4);
  memcpy(b + 12, &h.content, 4);
  write_all(fd, b, 16);
}

void write_stream_block(int fd, const FlatBatch &b) {
  if (b.size() == 0) return;
  if (b.size() > 0xFFFFFFFFull) throw Error("reads stream: block of more than 2^32 reads");
  const uint32_t n = (uint32_t)b.size(), zero = 0;
  const uint64_t nb = b.bases.size(), nt = b.tags.size();
  char h[24];
  memcpy(h, &n, 4);
  memcpy(h + 4, &zero, 4);
  memcpy(h + 8, &nb, 8);
  memcpy(h + 16, &nt, 8);
  write_all(fd, h, 24);
  write_all(fd, b.lens.data(), (size_t)n * 4);
  write_all(fd, b.tag_lens.data(), (size_t)n * 4);
  write_all(fd, b.tags.data(), nt);
  write_all(fd, b.bases.data(), nb);
}

void write_stream_end(int fd) {
  const uint32_t z[2] = {0, 0};
  write_all(fd, z, 8);
}

ReadStreamReader::ReadStreamReader(int fd) : fd_(fd) {
  char b[8];
  if (!read_all(fd_, b, 8)) throw Error("reads stream: truncated header");
  memcpy(&header.k, b, 4);
  memcpy(&header.content, b + 4, 4);
}

bool ReadStreamReader::next(FlatBatch &out) {
  out.clear();
  uint32_t h[2];
  if (!read_all(fd_, h, 8)) throw Error("reads stream: the end marker is missing (was the producer killed?)");
  if (h[0] == 0) return false;
  uint64_t sz[2];
  if (!read_all(fd_, sz, 16)) throw Error("reads stream: truncated block");
  const uint32_t n = h[0];
  if (sz[0] > (1ull << 40) || sz[1] > (1ull << 40)) throw Error("reads stream: implausible block size");
  out.lens.resize(n);
  out.tag_lens.resize(n);
  out.tags.resize(sz[1]);
  out.bases.resize(sz[0]);
  if (!read_all(fd_, out.lens.data(), (size_t)n * 4) || !read_all(fd_, out.tag_lens.data(), (size_t)n * 4))
    throw Error("reads stream: truncated block");
  if (sz[1] && !read_all(fd_, out.tags.data(), sz[1])) throw Error("reads stream: truncated block");
  if (sz[0] && !read_all(fd_, out.bases.data(), sz[0])) throw Error("reads stream: truncated block");
  uint64_t sb = 0, st = 0;
  for (uint32_t i = 0; i < n; ++i) {
    sb += out.lens[i];
    st += out.tag_lens[i];
  }
  if (sb != sz[0] || st != sz[1]) throw Error("reads stream: lengths do not add up");
  return true;
}

// ------------------------------------------------------------------ who reads our stdout?
namespace {
std::string link_of(const std::string &p) {
  char buf[4096];
  const ssize_t n = readlink(p.c_str(), buf, sizeof(buf) - 1);
  return n > 0 ? std::string(buf, (size_t)n) : std::string();
}
}  // namespace

bool stdout_reader_is_dropin_twistdb() {
  if (const char *e = getenv("KPOP_PIPE_FORMAT")) {
    if (!strcmp(e, "text")) return false;
    if (!strcmp(e, "reads")) return true;
  }
  struct stat st;
  if (fstat(1, &st) != 0 || !S_ISFIFO(st.st_mode)) return false;
  const std::string want = "pipe:[" + std::to_string((unsigned long long)st.st_ino) + "]";
  std::string self = link_of("/proc/self/exe");
  const size_t slash = self.rfind('/');
  if (slash == std::string::npos) return false;
  const std::string mate = self.substr(0, slash + 1) + "KPopTwistDB";
  const pid_t me = getpid();
  for (int attempt = 0; attempt < 40; ++attempt) {  // a reader between fork and exec still shows its parent's image
    bool other_reader = false, undecided = false;
    DIR *d = opendir("/proc");
    if (!d) return false;
    while (struct dirent *de = readdir(d)) {
      char *end = nullptr;
      const long pid = strtol(de->d_name, &end, 10);
      if (!end || *end || pid <= 0 || pid == me) continue;
      const std::string base = std::string("/proc/") + de->d_name;
      if (link_of(base + "/fd/0") != want) continue;
      const std::string exe = link_of(base + "/exe");
      if (exe == mate) {
        closedir(d);
        return true;
      }
      other_reader = true;
      const size_t s2 = exe.rfind('/');
      const std::string name = s2 == std::string::npos ? exe : exe.substr(s2 + 1);
      // a shell or an interpreter that still holds the pipe is most likely about to exec the real reader
      if (name == "bash" || name == "sh" || name == "dash" || name == "zsh" || name.compare(0, 6, "python") == 0 || exe.empty())
        undecided = true;
    }
    closedir(d);
    if (other_reader && !undecided) return false;
    struct timespec ts = {0, 5 * 1000 * 1000};
    nanosleep(&ts, nullptr);
  }
  return false;
}

}  // namespace kpop_host

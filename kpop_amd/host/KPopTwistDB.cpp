// KPopTwistDB -- drop-in for the hot actions of the reference's bin/KPopTwistDB.ml.
//
// Same register machine (bin/KPopTwistDB.ml:410-417), same option names, actions
// executed in command-line order (:361,:437).  Hot actions run in libkpop_hip.so:
//   -k  Twister.add_twisted_from_files   lib/Twister.ml:58-206   -> kpop_twist
//   -d  Matrix.get_distance_rowwise      lib/Matrix.ml:621-630   -> kpop_distance_rowwise
//   -s  Matrix.summarize_rowwise         lib/Matrix.ml:691-766   -> kpop_distance_summary
//   -S  Matrix.summarize_distance        lib/Matrix.ml:767-810   -> kpop_summarize_distances
// This file is argument parsing, the text formats, and the label bookkeeping of
// lib/Twister.ml:189-206.
//
// Binary registers (-i/-a/-o, and the twisted operand of -d/-s) are OCaml Marshal streams, read and written
// by ocaml_marshal.cpp; if '<prefix>.KPopTwisted' does not exist, -d/-s fall back to '<prefix>.KPopTwisted.txt'.
//   -e  Matrix.get_embeddings            lib/Matrix.ml:78-128    -> kpop_embeddings  (register 'e', '.KPopVectors')
//   -p  Matrix.get_splits                lib/Matrix.ml:524-612   -> kpop_splits_gaps (gaps) / host/splits.cpp (centroids);
//       register 's' is written as text only ('.PhyloSplits.txt', container declared in splits.h)
// Runtime failures exit 1 (the reference prints the exception and exits 0).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fcntl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <string_view>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/kpop_hip.h"
#include "fast_seq.h"
#include "kpop_text.h"
#include "ocaml_marshal.h"
#include "splits.h"

using namespace kpop_host;

namespace {

const char *kVersion = "38-hip";

enum class Reg { Metrics, Twister, Twisted, Embeddings, Distances, Splits };

Reg reg_of_string(const std::string &s) {  // bin/KPopTwistDB.ml:30-39
  if (s == "m") return Reg::Metrics;
  if (s == "T") return Reg::Twister;
  if (s == "t") return Reg::Twisted;
  if (s == "e") return Reg::Embeddings;
  if (s == "d") return Reg::Distances;
  if (s == "s") return Reg::Splits;
  throw Error("Invalid_register_type(\"" + s + "\")");
}

struct Action {
  enum Kind {
    Empty, TablesToRegister, AddTablesToRegister, BinaryToRegister, AddBinaryToRegister, RegisterToBinary, SetKmersNormalize, AddKmersFiles, RegisterToTables,
    SetPrecision, SetDistance, SetDistanceNormalize, SetMetric, DistancesFromTwisted, SetSummaryKeepAtMost,
    SummaryFromTwisted, SummaryFromDistances, EmbeddingsFromTwisted, SetSplitsAlgorithm, SetSplitsKeepAtMost, SetPrecisionSplits,
    SplitsFromEmbeddings, Unsupported
  } kind;
  Reg reg = Reg::Twisted;
  std::string s1, s2;
  std::vector<std::string> files;
  bool flag = false;
  long long num = 0;
};

struct Metric {
  int kind = KPOP_METRIC_POWERS;
  double pi = 1., thr = 1., pe = 2.;  // powers(1,1,2), bin/KPopTwistDB.ml:92
};

Metric metric_of_string(const std::string &s) {  // lib/Space.ml:109-131
  Metric m;
  if (s == "flat") {
    m.kind = KPOP_METRIC_FLAT;
    return m;
  }
  double a, b, c;
  char tail;
  if (sscanf(s.c_str(), "powers(%lf,%lf,%lf%c", &a, &b, &c, &tail) != 4 || tail != ')' || s.back() != ')')
    throw Error("Unknown_metric(\"" + s + "\")");
  if (a < 0.) throw Error("Negative_power(" + format_g(a, 15) + ")");
  if (b < 0. || b > 1.) throw Error("Invalid_threshold(" + format_g(b, 15) + ")");
  if (c < 0.) throw Error("Negative_power(" + format_g(c, 15) + ")");
  m.kind = KPOP_METRIC_POWERS;
  m.pi = a;
  m.thr = b;
  m.pe = c;
  return m;
}

struct Distance {
  int kind = KPOP_EUCLIDEAN;
  double p = 2.;
};

Distance distance_of_string(const std::string &s) {  // lib/Space.ml:208-225
  Distance d;
  if (s == "euclidean") return d;
  if (s == "cosine") {
    d.kind = KPOP_COSINE;
    return d;
  }
  double p;
  char tail;
  if (sscanf(s.c_str(), "minkowski(%lf%c", &p, &tail) != 2 || tail != ')' || s.back() != ')')
    throw Error("Unknown_distance(\"" + s + "\")");
  if (p < 0.) throw Error("Negative_power(" + format_g(p, 15) + ")");
  d.kind = KPOP_MINKOWSKI;
  d.p = p;
  return d;
}

bool bool_of_string(const std::string &s) {
  if (s == "true") return true;
  if (s == "false") return false;
  throw Error("expected 'true' or 'false', got '" + s + "'");
}

void usage(FILE *f) {
  fprintf(f,
          "This is KPopTwistDB (MI355X/HIP hot path) version %s\n"
          "Usage: KPopTwistDB [ACTIONS]   (executed in order of specification)\n"
          " -z|--zero|--empty T|t|d                  empty the register\n"
          " -I|--Input T|t|d <table_prefix>          load .KPopTwister.txt+.KPopInertia.txt | .KPopTwisted.txt | .KPopDMatrix.txt\n"
          " -A|--Add t|d <table_prefix>              add the rows of a table to the register\n"
          " --counts-normalize true|false            normalise spectra before twisting (default true)\n"
          " -k|--kmers|--add-kmers|--add-kmer-files <file>[,<file>...]   twist spectra, add to the twisted register\n"
          " --distance euclidean|cosine|minkowski(p) (default euclidean)\n"
          " --distance-normalize true|false          (default true)\n"
          " -m|--metric flat|powers(a,b,c)           (default powers(1,1,2))\n"
          " -d|--distances <twisted_prefix>          distances between the twisted register and <prefix>.KPopTwisted.txt\n"
          " --precision-for-tables <n>               (default 15)\n"
          " -O|--Output T|t|d|m <table_prefix>       write the register as table(s)\n"
          " --summary-at-most|--summary-keep-at-most <n>|all   (default 2)\n"
          " -s|--compute-and-summarize-distances <twisted_prefix> <summary_prefix>\n"
          " -S|--summarize-distances <summary_prefix>\n"
          " -T|--threads <n> (ignored)  -v|--verbose  -V|--version  -h|--help\n"
          " -i|--input T|t|d <binary_prefix>         load .KPopTwister | .KPopTwisted | .KPopDMatrix (OCaml Marshal)\n"
          " -a|--add t|d <binary_prefix>             add the rows of a binary register\n"
          " -o|--output T|t|d <binary_prefix>        write the register in binary form\n"
          " -e|--embeddings|--compute-embeddings|--twisted-to-embeddings   twisted register -> embeddings register ('e')\n"
          " -p|--splits|--compute-splits|--embeddings-to-splits   embeddings register -> splits register ('s'); -O s <prefix> writes '.PhyloSplits.txt'\n"
          " --splits-algorithm gaps|centroids (default gaps)  --splits-keep-at-most <n> (default 10000)  --precision-for-splits <n> (default 10)\n",
          kVersion);
}

[[noreturn]] void parse_error(const std::string &msg) {
  usage(stderr);
  fprintf(stderr, "(KPopTwistDB): ERROR: %s\n", msg.c_str());
  exit(1);
}

void check(int rc) {
  if (rc != 0) throw Error(std::string("libkpop_hip: ") + kpop_last_error());
}

bool g_gpu = false;
std::thread g_warm;  // brings the HIP runtime up while the main thread reads its first archive
void warm_gpu() {
  g_warm = std::thread([] { (void)kpop_device_count(); });
}
// KPOP_DEVICES = n | all: several GPUs from THIS process (kpop_init_devices; the library runs one host thread per device
// and cuts every batch itself -- round 2 forked a worker process per GPU and moved reads through pipes and rows through
// shared memory).  More slots than GPUs (a test rig on a one-GPU box) wrap around.
int devices_requested(int visible) {
  const char *e = getenv("KPOP_DEVICES");
  if (!e || !*e) return 1;
  if (!strcmp(e, "all")) return std::max(visible, 1);
  const int n = atoi(e);
  return n > 1 ? std::min(n, 16) : 1;
}
void need_gpu() {
  if (g_warm.joinable()) g_warm.join();
  if (g_gpu) return;
  int dev = 0;
  if (const char *e = getenv("KPOP_DEVICE")) dev = atoi(e);
  const int visible = kpop_device_count();
  const int n = devices_requested(visible);
  if (n > 1 && visible > 0) {
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) devs[i] = (dev + i) % visible;
    check(kpop_init_devices(devs.data(), n));
  } else {
    check(kpop_init(dev));
  }
  g_gpu = true;
}

struct TwisterReg {
  Table twister, inertia;  // dims x k-mers ; 1 x dims
  kpop_twister *dev = nullptr;
  size_t name_len = 0;
  std::string lazy_path;  // a binary archive whose twister matrix has not been read yet (-d / -s / -e only need the inertia)
  void reset() {
    if (dev) kpop_twister_free(dev);
    dev = nullptr;
    twister = Table();
    inertia = Table();
    lazy_path.clear();
    opaque = false;
    columns.clear();
    index.clear();
  }
  void need_matrix() {
    if (lazy_path.empty()) return;
    const std::string p = lazy_path;
    lazy_path.clear();
    read_binary_twister(p, &twister, &inertia);
  }
  // How k-mer names meet the twister's columns (lib/Twister.ml:71-76,151: a Hashtbl over the names, which are opaque strings there).
  // Every column name the same number (<= 15) of lowercase hexadecimal digits -- what KPopCount writes -- and a k-mer's number is the
  // value of its name: the reads stream and the fused kernels apply.  Anything else (names of a reduced alphabet, of a tool of the
  // user's own) and a k-mer's number is its column's index through a dictionary over the names; only text spectra can meet such a twister.
  bool opaque = false;
  std::vector<uint64_t> columns;                            // hex: the columns' numbers, ascending
  std::unordered_map<std::string_view, uint64_t> index;     // opaque: name -> column (views into twister.col_names)
  KmerLookup lookup_;
  const KmerLookup *lookup() {  // (pointers into this register: set at every use, the register may have been moved)
    lookup_.opaque = opaque;
    lookup_.columns = opaque ? nullptr : &columns;
    lookup_.index = opaque ? &index : nullptr;
    return &lookup_;
  }
  void upload() {
    if (dev) return;
    need_matrix();
    stage_mark("KPopTwistDB", "  twister matrix in memory");
    need_gpu();
    stage_mark("KPopTwistDB", "  HIP runtime up");
    const size_t n = twister.cols();
    std::vector<uint64_t> col_hash(n);
    name_len = n ? twister.col_names[0].size() : 0;
    std::atomic<bool> all_hex{name_len >= 1 && name_len <= 15};  // (millions of names: the host threads convert them)
    if (all_hex.load())
      parallel_for(n, 65536, [&](size_t lo, size_t hi) {
        for (size_t c = lo; c < hi && all_hex.load(std::memory_order_relaxed); ++c) {
          const std::string &nm = twister.col_names[c];
          bool ok = nm.size() == name_len;
          uint64_t v = 0;
          for (size_t i = 0; ok && i < name_len; ++i) {
            const char ch = nm[i];
            if (ch >= '0' && ch <= '9') v = (v << 4) | (uint64_t)(ch - '0');
            else if (ch >= 'a' && ch <= 'f') v = (v << 4) | (uint64_t)(ch - 'a' + 10);
            else ok = false;
          }
          if (!ok) {
            all_hex.store(false);
            return;
          }
          col_hash[c] = v;
        }
      });
    opaque = n > 0 && !all_hex.load();
    int k;
    if (opaque) {
      if (n >= (1ull << 32) - 2) throw Error("more than 2^32-2 twister columns");
      index.reserve(n * 2);
      for (size_t c = 0; c < n; ++c) {
        index[std::string_view(twister.col_names[c])] = c;  // (a later column of the same name shadows, Hashtbl.add)
        col_hash[c] = c;
      }
      name_len = 0;
      k = 30;  // numbers below 2^32: the library finds rows by bisection over them, no k-mer arithmetic is involved
    } else {
      k = (int)std::min<size_t>(2 * name_len, 30);  // names carry ceil(k/2) hex digits; the larger k covers both
      columns = col_hash;
      if (!std::is_sorted(columns.begin(), columns.end())) std::sort(columns.begin(), columns.end());
    }
    stage_mark("KPopTwistDB", opaque ? "  column names into a dictionary" : "  column names to hashes");
    check(kpop_twister_load(twister.data.data(), n, (uint32_t)twister.rows(), col_hash.data(), std::max(k, 1), &dev));
    stage_mark("KPopTwistDB", "  twister on the device");
  }
};

void load_twister_tables(TwisterReg &T, const std::string &prefix) {  // Twister.of_files, lib/Twister.ml:32-51
  T.reset();
  T.twister = read_table(make_filename(prefix, "KPopTwister", true));
  T.inertia = read_table(make_filename(prefix, "KPopInertia", true));
  if (T.inertia.row_names != std::vector<std::string>{"inertia"} || T.twister.row_names != T.inertia.col_names)
    throw Error("Mismatched_twister_files");  // :36-49
}

// the twisted operand of -d / -s is a binary prefix in the reference (bin/KPopTwistDB.ml:546,554)
Table load_twisted_operand(const std::string &prefix) {
  const std::string bin = make_filename(prefix, "KPopTwisted", false);
  if (FILE *f = fopen(bin.c_str(), "rb")) {
    fclose(f);
    return read_binary_matrix(bin, "KPopTwisted");
  }
  return read_table(make_filename(prefix, "KPopTwisted", true));
}

std::vector<double> metric_vector(const Metric &m, const TwisterReg &T) {  // Twister.get_metrics_vector, lib/Twister.ml:208-209
  std::vector<double> out(T.inertia.cols());
  check(kpop_metric_compute(m.kind, T.inertia.data.data(), (uint32_t)out.size(), m.pi, m.thr, m.pe, out.data()));
  return out;
}

// One piece of twisted rows in arrival order (a spectra file, or a block of the reads stream)
struct RowPiece {
  std::vector<std::string> labels;
  DVec rows;
};

// blocks of the reads stream, read ahead of the GPU by one thread
struct BlockQueue {
  std::mutex m;
  std::condition_variable cv;
  std::deque<FlatBatch> q;
  bool done = false;
  std::string error;
  void push(FlatBatch &&b) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return q.size() < 2; });
    q.push_back(std::move(b));
    cv.notify_all();
  }
  bool pop(FlatBatch &b) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return !q.empty() || done; });
    if (q.empty()) return false;
    b = std::move(q.front());
    q.pop_front();
    cv.notify_all();
    return true;
  }
  void finish(const std::string &err) {
    std::lock_guard<std::mutex> l(m);
    done = true;
    error = err;
    cv.notify_all();
  }
};

// The reads stream (fast_seq.h): KPopCount sent the linted reads instead of their spectra; count and twist them in one
// kernel.  Equal to parsing the text KPopCount would have written (bin/KPopCount.ml:44-46) and twisting that.
void twist_read_stream(int fd, TwisterReg &T, bool normalize, std::vector<RowPiece> &pieces, bool verbose) {
#ifdef F_SETPIPE_SZ
  (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);  // (fails on anything but a pipe: fine)
#endif
  ReadStreamReader rs(fd);
  const int k = (int)rs.header.k, content = (int)rs.header.content;
  const bool protein = content == KPOP_PROTEIN;
  if (k < 1 || k > (protein ? 12 : 30) || (content != KPOP_DNA_DS && content != KPOP_DNA_SS && !protein)) throw Error("reads stream: unsupported k or content");
  T.upload();
  const size_t d = T.twister.rows();
  // the names KPopCount would have written carry name_digits(k) hex digits; if the twister's names are of another
  // width no k-mer of the stream can be a column of it (lib/Twister.ml:167-169): every row is the zero vector
  const bool can_match = !T.opaque && (size_t)name_digits(k, protein) == T.name_len;
  // Blocks whose sequences all fit one wavefront (<= 512 windows) and more than 32 dimensions: the fused count->twist
  // kernel IS count + twist there (same ascending chain of unfused multiply-adds), and the block goes through the
  // library's streaming pipeline -- chunks of it going up, being twisted and coming down at the same time -- on every
  // device slot (kpop_sharded_run).  Anything else (genomes, few dimensions) takes kpop_spectra_twist, cut over the slots.
  struct Job {
    kpop_sharded *sh = nullptr;
    ~Job() {
      if (sh) kpop_sharded_destroy(sh);
    }
  } job;
  const int slots = kpop_device_slots();
  if (can_match) {
    check(kpop_twister_set_count_k(T.dev, k));
    kpop_pipeline_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.content = content;
    cfg.normalize_counts = normalize ? 1 : 0;
    cfg.kind = KPOP_EUCLIDEAN;
    cfg.p = 2.0;
    cfg.outputs = KPOP_OUT_TWISTED;
    check(kpop_sharded_create(T.dev, nullptr, 0, nullptr, &cfg, &job.sh));
    if (verbose && slots > 1) fprintf(stderr, "(KPopTwistDB): %d device slots, one host thread each, for the reads stream\n", slots);
  }
  BlockQueue bq;
  std::thread reader([&] {
    std::string err;
    try {
      FlatBatch b;
      while (rs.next(b)) bq.push(std::move(b));
    } catch (const std::exception &e) {
      err = e.what();
    }
    bq.finish(err);
  });
  try {
    FlatBatch b;
    std::vector<uint64_t> offsets;
    uint64_t n_reads = 0;
    while (bq.pop(b)) {
      stage_mark("KPopTwistDB", "  stream block arrived");
      const size_t n = b.size();
      RowPiece piece;
      piece.labels.resize(n);
      parallel_for(n, 65536, [&](size_t lo, size_t hi) {
        size_t at = 0;
        for (size_t r = 0; r < lo; ++r) at += b.tag_lens[r];
        for (size_t r = lo; r < hi; ++r) {
          // once by KPopCount when it prints the label (bin/KPopCount.ml:45), once by the parser (lib/Twister.ml:110)
          piece.labels[r] = strip_external_quotes_and_check(strip_external_quotes_and_check(std::string(b.tags.data() + at, b.tag_lens[r])));
          at += b.tag_lens[r];
        }
      });
      offsets.resize(n + 1);
      offsets[0] = 0;
      for (size_t r = 0; r < n; ++r) offsets[r + 1] = offsets[r] + b.lens[r];
      piece.rows.resize(n * d);
      if (can_match && n) {
        static const uint8_t dummy = 0;
        const uint8_t *bp = b.bases.empty() ? &dummy : b.bases.data();
        uint64_t max_len = 0;
        for (size_t r = 0; r < n; ++r) max_len = std::max<uint64_t>(max_len, b.lens[r]);
        const bool fused = d > 32 && max_len < (uint64_t)k + 512;
        if (fused) {
          kpop_pipeline_outputs po;
          memset(&po, 0, sizeof po);
          po.twisted = piece.rows.data();
          check(kpop_sharded_run(job.sh, bp, offsets.data(), (uint32_t)n, &po));
        } else {
          check(kpop_sharded_spectra_twist(job.sh, bp, offsets.data(), (uint32_t)n, k, content, normalize ? 1 : 0, piece.rows.data()));
        }
      } else if (T.opaque && n) {
        // a twister over names of its own: the stream's k-mers meet it by the names KPopCount would have written for them
        // (bin/KPopCount.ml:46) -- counted on the device, named and looked up here, twisted on the device
        const uint64_t cap = offsets[n] + 1;
        std::vector<uint64_t, DefaultInitAlloc<uint64_t>> h(cap);
        std::vector<uint32_t, DefaultInitAlloc<uint32_t>> c(cap);
        std::vector<uint64_t> so(n + 1);
        static const uint8_t dummy = 0;
        check(kpop_count_reads(b.bases.empty() ? &dummy : b.bases.data(), offsets.data(), (uint32_t)n, k, content, 1, h.data(), c.data(), so.data(), cap));
        const uint64_t m = so[n];
        DVec v(m);
        const int digits = name_digits(k, protein);
        parallel_for(m, 65536, [&](size_t lo, size_t hi) {
          static const char hx[] = "0123456789abcdef";
          char name[16];
          for (size_t i = lo; i < hi; ++i) {
            uint64_t x = h[i];
            for (int q = digits - 1; q >= 0; --q) {
              name[q] = hx[x & 15];
              x >>= 4;
            }
            const auto it = T.index.find(std::string_view(name, (size_t)digits));
            h[i] = it == T.index.end() ? (~0ull >> 1) : it->second;
            v[i] = (double)c[i];
          }
        });
        check(kpop_twist(T.dev, h.data(), v.data(), so.data(), (uint32_t)n, normalize ? 1 : 0, piece.rows.data()));
      } else {
        std::fill(piece.rows.begin(), piece.rows.end(), 0.);
      }
      n_reads += n;
      pieces.push_back(std::move(piece));
      stage_mark("KPopTwistDB", "  stream block counted+twisted");
      if (verbose) fprintf(stderr, "(KPopTwistDB): reads stream: %llu sequences counted and twisted so far\n", (unsigned long long)n_reads);
    }
  } catch (...) {
    {  // let the reader run to its end so that it can be joined
      std::lock_guard<std::mutex> l(bq.m);
      bq.q.clear();
      bq.cv.notify_all();
    }
    FlatBatch drop;
    while (bq.pop(drop)) {
    }
    reader.join();
    throw;
  }
  reader.join();
  if (!bq.error.empty()) throw Error(bq.error);
}

// a queue of at most two items between two stages of a pipeline; finish() by the producer, cancel() by the consumer
template <class Item>
struct StageQueue {
  std::mutex m;
  std::condition_variable cv;
  std::deque<Item> q;
  bool done = false, cancelled = false;
  std::string error;
  bool push(Item &&b) {  // false: the consumer has gone away
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return q.size() < 2 || cancelled; });
    if (cancelled) return false;
    q.push_back(std::move(b));
    cv.notify_all();
    return true;
  }
  bool pop(Item &b) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return !q.empty() || done; });
    if (q.empty()) return false;
    b = std::move(q.front());
    q.pop_front();
    cv.notify_all();
    return true;
  }
  void finish(const std::string &err) {
    std::lock_guard<std::mutex> l(m);
    done = true;
    error = err;
    cv.notify_all();
  }
  void cancel() {
    std::lock_guard<std::mutex> l(m);
    cancelled = true;
    q.clear();
    cv.notify_all();
  }
};

// Text spectra (lib/Twister.ml:91-145): one thread reads blocks that end where a spectrum ends, a second parses them (on
// the host threads), this one twists them -- the three overlap, where reading everything, then parsing, then twisting was
// 2.1 s for a million read spectra.  Same rows, same errors in the same order as the whole-file parser.
void twist_text_spectra(int fd, const char *head, size_t head_len, TwisterReg &T, uint64_t absent, bool normalize, size_t d,
                        std::vector<RowPiece> &pieces) {
  StageQueue<TextBlock> blocks;
  StageQueue<HashedSpectra> parsed;
  const bool timing = getenv("KPOP_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  double t_read = 0., t_parse = 0., t_twist = 0.;  // busy time of each stage (each written by its own thread, read after the joins)
  std::thread reader([&] {
    std::string err;
    try {
      SpectraTextStream ts(fd, head, head_len);
      TextBlock b;
      for (;;) {
        const auto t0 = now();
        const bool more = ts.next(b);
        t_read += secs(t0, now());
        if (!more || !blocks.push(std::move(b))) break;
      }
    } catch (const std::exception &e) {
      err = e.what();
    }
    blocks.finish(err);
  });
  std::thread parser([&] {
    std::string err;
    try {
      TextBlock b;
      bool first = true;
      uint64_t lines = 0, n = 0;
      while (blocks.pop(b)) {
        HashedSpectra sp;
        const auto t0 = now();
        parse_spectra_block(b.data(), b.size(), T.name_len, absent, first, lines, sp, &n, 0, nullptr, T.lookup());
        t_parse += secs(t0, now());
        first = false;
        lines += n;
        if (!parsed.push(std::move(sp))) break;
      }
      if (!blocks.error.empty()) err = blocks.error;
    } catch (const std::exception &e) {
      err = e.what();
      blocks.cancel();
    }
    parsed.finish(err);
  });
  try {
    HashedSpectra sp;
    while (parsed.pop(sp)) {
      RowPiece piece;
      const size_t n = sp.labels.size();
      const auto t0 = now();
      piece.rows.resize(n * d);
      if (n) check(kpop_twist(T.dev, sp.hash.data(), sp.values.data(), sp.offsets.data(), (uint32_t)n, normalize ? 1 : 0, piece.rows.data()));
      t_twist += secs(t0, now());
      piece.labels.swap(sp.labels);
      pieces.push_back(std::move(piece));
    }
  } catch (...) {
    parsed.cancel();
    blocks.cancel();
    parser.join();
    reader.join();
    throw;
  }
  parser.join();
  reader.join();
  if (timing) fprintf(stderr, "[timing] KPopTwistDB:   text spectra: reading %.3f s, parsing %.3f s, twisting %.3f s (busy time of the three stages)\n", t_read, t_parse, t_twist);
  if (!parsed.error.empty()) throw Error(parsed.error);
}

// Twister.add_twisted_from_files, lib/Twister.ml:58-206
void add_twisted_from_files(TwisterReg &T, Table &twisted, const std::vector<std::string> &files, bool normalize, bool verbose) {
  // with worker processes the parent needs the dimension names only (the inertia's columns, lib/Twister.ml:36-38); the
  // matrix is read and uploaded here only if text spectra turn up, which the parent twists itself
  T.need_matrix();
  stage_mark("KPopTwistDB", "twister archive read");
  const std::vector<std::string> dims = T.twister.row_names;
  if (!twisted.empty() && twisted.col_names != dims) throw Error("Incompatible_twister_and_twisted");  // :64-69
  T.upload();
  stage_mark("KPopTwistDB", "HIP bring-up + twister upload");
  const size_t d = dims.size();
  std::vector<RowPiece> pieces;
  const uint64_t absent = ~0ull >> 1;  // no twister column carries this hash (k <= 30)
  for (const std::string &f : files) {
    const bool is_stdin = f == "/dev/stdin" || f == "-";
    const int fd = is_stdin ? 0 : open(f.c_str(), O_RDONLY);
    if (fd < 0) throw Error("cannot open '" + f + "': " + strerror(errno));
    try {
      for (;;) {  // a file is text spectra, or reads streams (one per KPopCount that wrote to the pipe) and then maybe text
        char head[8];
        size_t got = 0;
        while (got < 8) {
          const ssize_t r = read(fd, head + got, 8 - got);
          if (r < 0 && errno == EINTR) continue;
          if (r <= 0) break;
          got += (size_t)r;
        }
        if (got == 8 && memcmp(head, kReadStreamMagic, 8) == 0) {
          twist_read_stream(fd, T, normalize, pieces, verbose);
          continue;
        }
        if (got == 0 && !pieces.empty()) break;  // end of file right after a stream
        // names -> hashes while parsing; a name the twister cannot hold is simply an unknown k-mer (:167-169)
        T.upload();  // (a no-op unless worker processes had made it unnecessary so far)
        twist_text_spectra(fd, head, got, T, absent, normalize, d, pieces);
        break;
      }
    } catch (...) {
      if (!is_stdin) close(fd);
      throw;
    }
    if (!is_stdin) close(fd);
    if (verbose) {
      size_t n = 0;
      for (const RowPiece &p : pieces) n += p.labels.size();
      fprintf(stderr, "(KPopTwistDB): File '%s': read %zu spectra so far\n", f.c_str(), n);
    }
  }
  stage_mark("KPopTwistDB", "input consumed, rows twisted");
  // existing rows first, new labels must be new (:78-82,:189-195), result in bytewise label order (:197-204)
  const size_t n_old = twisted.rows();
  std::vector<std::string> labels;
  std::vector<const double *> row_of;
  {
    size_t n_new = 0;
    for (const RowPiece &p : pieces) n_new += p.labels.size();
    labels.reserve(n_old + n_new);
    row_of.reserve(n_old + n_new);
    for (size_t r = 0; r < n_old; ++r) {
      labels.push_back(twisted.row_names[r]);
      row_of.push_back(twisted.data.data() + r * d);
    }
    for (RowPiece &p : pieces)
      for (size_t r = 0; r < p.labels.size(); ++r) {
        labels.push_back(std::move(p.labels[r]));
        row_of.push_back(p.rows.data() + r * d);
      }
  }
  const std::vector<uint32_t> order = order_rows_by_label(labels, n_old);
  stage_mark("KPopTwistDB", "rows ordered by label");
  Table out;
  out.col_names = dims;
  out.row_names.resize(order.size());
  out.data.resize(order.size() * d);
  parallel_for(order.size(), 8192, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      out.row_names[i] = std::move(labels[order[i]]);
      memcpy(out.data.data() + i * d, row_of[order[i]], d * sizeof(double));
    }
  });
  twisted.col_names.swap(out.col_names);
  twisted.row_names.swap(out.row_names);
  twisted.data.swap(out.data);
  stage_mark("KPopTwistDB", "register assembled");
  // millions of label strings and gigabytes of rows are now garbage: giving them back takes 0.2 s at 4M reads, so a
  // thread of its own does it while the next action runs
  auto *garbage = new std::tuple<std::vector<RowPiece>, std::vector<std::string>, Table>(std::move(pieces), std::move(labels), std::move(out));
  std::thread([garbage] { delete garbage; }).detach();
}

void write_summary(const std::string &path, const std::vector<std::string> &row_names, const std::vector<std::string> &col_names,
                   const std::vector<double> &stats, const std::vector<uint32_t> &n, const std::vector<uint32_t> &idx,
                   const std::vector<double> &dist, const std::vector<double> &z, uint32_t stride, bool quote) {
  for (size_t j = 0; j < row_names.size(); ++j)
    if (n[j] > stride) throw Error("summary row '" + row_names[j] + "' has " + std::to_string(n[j]) + " tied neighbours, more than fit");
  FILE *f = (path == "/dev/stdout") ? stdout : fopen(path.c_str(), "wb");
  if (!f) throw Error("cannot write '" + path + "'");
  const char *q = quote ? "\"" : "";
  // lib/Matrix.ml:684-690
  try {
    write_rows_parallel(f, path, row_names.size(), (size_t)(96 + 40 * std::min<uint32_t>(stride, 4)), [&](size_t j, std::string &o) {
      o += q;
      o += row_names[j];
      o += q;
      for (int c = 0; c < 4; ++c) {
        o += '\t';
        append_g(o, stats[j * 4 + c], 15);
      }
      for (uint32_t e = 0; e < n[j]; ++e) {
        o += '\t';
        o += q;
        o += col_names[idx[j * stride + e]];
        o += q;
        o += '\t';
        append_g(o, dist[j * stride + e], 15);
        o += '\t';
        append_g(o, z[j * stride + e], 15);
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

}  // namespace

int main(int argc, char **argv) {
  std::vector<Action> program;
  bool verbose = false, quote_summary = false;
  auto need = [&](int &i, const std::string &opt) -> std::string {
    if (i + 1 >= argc) parse_error("Option '" + opt + "' needs a parameter");
    return argv[++i];
  };
  try {
    for (int i = 1; i < argc; ++i) {
      const std::string a = argv[i];
      Action act;
      if (a == "-z" || a == "--zero" || a == "--empty") {
        act.kind = Action::Empty;
        act.reg = reg_of_string(need(i, a));
        if (act.reg == Reg::Metrics || act.reg == Reg::Splits) parse_error("You cannot load content into the metric or splits registers");
      } else if (a == "-i" || a == "--input") {
        act.kind = Action::BinaryToRegister;
        act.reg = reg_of_string(need(i, a));
        if (act.reg == Reg::Metrics || act.reg == Reg::Splits) parse_error("You cannot load content into the metric or splits registers");
        act.s1 = need(i, a);
      } else if (a == "-a" || a == "--add") {
        act.kind = Action::AddBinaryToRegister;
        act.reg = reg_of_string(need(i, a));
        if (act.reg == Reg::Twister || act.reg == Reg::Metrics || act.reg == Reg::Splits)
          parse_error("You cannot add content to the twister, metric or splits registers");
        act.s1 = need(i, a);
      } else if (a == "-o" || a == "--output") {
        act.kind = Action::RegisterToBinary;
        act.reg = reg_of_string(need(i, a));
        if (act.reg == Reg::Metrics) parse_error("You cannot output binary content from the metrics registers");
        act.s1 = need(i, a);
      } else if (a == "-I" || a == "--Input") {
        act.kind = Action::TablesToRegister;
        act.reg = reg_of_string(need(i, a));
        if (act.reg == Reg::Metrics || act.reg == Reg::Splits) parse_error("You cannot load content into the metric or splits registers");
        act.s1 = need(i, a);
      } else if (a == "-A" || a == "--Add") {
        act.kind = Action::AddTablesToRegister;
        act.reg = reg_of_string(need(i, a));
        if (act.reg == Reg::Twister || act.reg == Reg::Metrics || act.reg == Reg::Splits)
          parse_error("You cannot add content to the twister, metric or splits registers");
        act.s1 = need(i, a);
      } else if (a == "--counts-normalize" || a == "--counts-normalization") {
        act.kind = Action::SetKmersNormalize;
        act.flag = bool_of_string(need(i, a));
      } else if (a == "-k" || a == "--kmers" || a == "--add-kmers" || a == "--add-kmer-files") {
        act.kind = Action::AddKmersFiles;
        std::string l = need(i, a);
        size_t st = 0;
        for (;;) {
          size_t p = l.find(',', st);
          act.files.push_back(l.substr(st, p == std::string::npos ? p : p - st));
          if (p == std::string::npos) break;
          st = p + 1;
        }
      } else if (a == "--distance" || a == "--distance-function") {
        act.kind = Action::SetDistance;
        act.s1 = need(i, a);
        distance_of_string(act.s1);
      } else if (a == "--distance-normalize" || a == "--distance-normalization") {
        act.kind = Action::SetDistanceNormalize;
        act.flag = bool_of_string(need(i, a));
      } else if (a == "-m" || a == "--metric" || a == "--metric-function") {
        act.kind = Action::SetMetric;
        act.s1 = need(i, a);
        metric_of_string(act.s1);
      } else if (a == "-d" || a == "--distances" || a == "--compute-distances" || a == "--compute-twisted-distances") {
        act.kind = Action::DistancesFromTwisted;
        act.s1 = need(i, a);
      } else if (a == "--precision-for-tables") {
        act.kind = Action::SetPrecision;
        act.num = atoll(need(i, a).c_str());
        if (act.num <= 0) parse_error("precision must be positive");
      } else if (a == "-O" || a == "--Output") {
        act.kind = Action::RegisterToTables;
        act.reg = reg_of_string(need(i, a));
        act.s1 = need(i, a);
      } else if (a == "--summary-at-most" || a == "--summary-keep-at-most" || a == "--keep-at-most") {  // the last: README.md:768,1101
        act.kind = Action::SetSummaryKeepAtMost;
        std::string v = need(i, a);
        if (v == "all") act.num = 0;
        else {
          act.num = atoll(v.c_str());
          if (act.num <= 0) parse_error("Invalid_keep_at_most(\"" + v + "\")");  // bin/KPopTwistDB.ml:46-56
        }
      } else if (a == "-s" || a == "--compute-and-summarize-distances" || a == "--compute-and-summarize-twisted-distances") {
        act.kind = Action::SummaryFromTwisted;
        act.s1 = need(i, a);
        act.s2 = need(i, a);
      } else if (a == "-S" || a == "--summarize-distances" || a == "--summarize-twisted-distances") {
        act.kind = Action::SummaryFromDistances;
        act.s1 = need(i, a);
      } else if (a == "-e" || a == "--embeddings" || a == "--compute-embeddings" || a == "--twisted-to-embeddings") {
        act.kind = Action::EmbeddingsFromTwisted;
      } else if (a == "-p" || a == "--splits" || a == "--compute-splits" || a == "--embeddings-to-splits") {
        act.kind = Action::SplitsFromEmbeddings;
      } else if (a == "--splits-algorithm") {  // bin/KPopTwistDB.ml:234-239
        act.kind = Action::SetSplitsAlgorithm;
        act.s1 = need(i, a);
        if (act.s1 != "gaps" && act.s1 != "centroids") parse_error("Unknown_algorithm(\"" + act.s1 + "\")");
      } else if (a == "--splits-at-most" || a == "--splits-keep-at-most") {  // :240-246
        act.kind = Action::SetSplitsKeepAtMost;
        act.num = atoll(need(i, a).c_str());
        if (act.num <= 0) parse_error("the number of splits to keep must be a positive integer");
      } else if (a == "--precision-for-splits") {
        act.kind = Action::SetPrecisionSplits;
        act.num = atoll(need(i, a).c_str());
        if (act.num <= 0) parse_error("precision must be positive");
      } else if (a == "-T" || a == "--threads") {
        need(i, a);
        continue;
      } else if (a == "--summary-quote-names") {
        quote_summary = true;
        continue;
      } else if (a == "-v" || a == "--verbose") {
        verbose = true;
        continue;
      } else if (a == "-x" || a == "--print-exception-backtrace" || a == "--debug-twisting") {
        continue;
      } else if (a == "-V" || a == "--version") {
        printf("%s\n", kVersion);
        return 0;
      } else if (a == "-h" || a == "--help") {
        usage(stdout);
        return 0;
      } else {
        parse_error("Unknown option '" + a + "'");
      }
      program.push_back(act);
    }
  } catch (const Error &e) {
    parse_error(e.what());
  }
  if (program.empty()) {
    usage(stdout);
    return 0;
  }
  // static checks, bin/KPopTwistDB.ml:368-408
  {
    bool twister_loaded = false;
    for (const Action &a : program) {
      if ((a.kind == Action::TablesToRegister || a.kind == Action::BinaryToRegister) && a.reg == Reg::Twister) twister_loaded = true;
      if (a.kind == Action::AddKmersFiles && !twister_loaded) parse_error("Option '-k' requires a twister in the twister register!");
      if (((a.kind == Action::RegisterToTables && a.reg == Reg::Metrics) || a.kind == Action::DistancesFromTwisted ||
           a.kind == Action::SummaryFromTwisted || a.kind == Action::EmbeddingsFromTwisted) && !twister_loaded)
        parse_error("Options '-O m', '-e', '-d', and '-s' require a twister in the twister register to provide a metric!");
    }
  }

  stage_mark("KPopTwistDB", "start");
  for (const Action &a : program)
    if (a.kind == Action::AddKmersFiles || a.kind == Action::DistancesFromTwisted || a.kind == Action::SummaryFromTwisted ||
        a.kind == Action::SummaryFromDistances || a.kind == Action::EmbeddingsFromTwisted) {
      warm_gpu();
      break;
    }
  TwisterReg T;
  Table twisted, embeddings, distances;
  Metric metric;
  Distance distance;
  bool kmers_normalize = true, distance_normalize = true;  // bin/KPopTwistDB.ml:87-98
  long long keep_at_most = 2, splits_keep_at_most = 10000;
  int precision = 15, precision_splits = 10;
  std::string splits_algorithm = "gaps";
  Splits splits;
  try {
    for (const Action &a : program) {
      stage_mark("KPopTwistDB", "-- next action");
      switch (a.kind) {
        case Action::Empty:
          if (a.reg == Reg::Twister) T.reset();
          else if (a.reg == Reg::Twisted) twisted = Table();
          else if (a.reg == Reg::Embeddings) embeddings = Table();
          else if (a.reg == Reg::Distances) distances = Table();
          break;
        case Action::BinaryToRegister:  // bin/KPopTwistDB.ml:449-456
          if (a.reg == Reg::Twister) {
            T.reset();
            const std::string path = make_filename(a.s1, "KPopTwister", false);
            if (path.compare(0, 5, "/dev/") == 0) read_binary_twister(path, &T.twister, &T.inertia);  // not seekable
            else {
              read_binary_twister_inertia(path, &T.inertia);
              T.lazy_path = path;
            }
          } else if (a.reg == Reg::Twisted) twisted = read_binary_matrix(make_filename(a.s1, "KPopTwisted", false), "KPopTwisted");
          else if (a.reg == Reg::Distances) distances = read_binary_matrix(make_filename(a.s1, "KPopDMatrix", false), "KPopDMatrix");
          else if (a.reg == Reg::Embeddings) embeddings = read_binary_matrix(make_filename(a.s1, "KPopVectors", false), "KPopVectors");
          else throw Error("nothing can be loaded into the metrics or splits registers");
          break;
        case Action::AddBinaryToRegister:  // :462-467
          if (a.reg == Reg::Twisted) merge_rowwise(twisted, read_binary_matrix(make_filename(a.s1, "KPopTwisted", false), "KPopTwisted"));
          else if (a.reg == Reg::Distances) merge_rowwise(distances, read_binary_matrix(make_filename(a.s1, "KPopDMatrix", false), "KPopDMatrix"));
          else if (a.reg == Reg::Embeddings) merge_rowwise(embeddings, read_binary_matrix(make_filename(a.s1, "KPopVectors", false), "KPopVectors"));
          else throw Error("nothing can be added to the twister, metrics or splits registers");
          break;
        case Action::RegisterToBinary:  // :509-518
          if (a.reg == Reg::Twister) {
            T.need_matrix();
            write_binary_twister(make_filename(a.s1, "KPopTwister", false), T.twister, T.inertia);
          }
          else if (a.reg == Reg::Twisted) write_binary_matrix(make_filename(a.s1, "KPopTwisted", false), "KPopTwisted", twisted);
          else if (a.reg == Reg::Distances) write_binary_matrix(make_filename(a.s1, "KPopDMatrix", false), "KPopDMatrix", distances);
          else if (a.reg == Reg::Embeddings) write_binary_matrix(make_filename(a.s1, "KPopVectors", false), "KPopVectors", embeddings);
          else throw Error("binary splits ('.PhyloSplits' is BiOCamLib's Marshal of Trees.Splits.t, absent from the reference checkout) are not provided: use -O s");
          break;
        case Action::Unsupported:
          throw Error("action '" + a.s1 + "' is not provided by this tool");
        case Action::SetSplitsAlgorithm: splits_algorithm = a.s1; break;
        case Action::SetSplitsKeepAtMost: splits_keep_at_most = a.num; break;
        case Action::SetPrecisionSplits: precision_splits = (int)a.num; break;
        case Action::SplitsFromEmbeddings: {  // bin/KPopTwistDB.ml:503-506 -> Matrix.get_splits, lib/Matrix.ml:524-612
          const size_t n = embeddings.rows(), d = embeddings.cols();
          if (splits_algorithm == "centroids") {
            splits = splits_centroids(embeddings.row_names, embeddings.data.data(), d, verbose);
          } else {
            need_gpu();
            splits = Splits();
            splits.names = embeddings.row_names;
            const uint32_t want = (uint32_t)std::min<long long>(splits_keep_at_most, 0x7FFFFFFFll);
            std::vector<double> gap(want ? want : 1);
            std::vector<uint32_t> dim(want ? want : 1), idx(want ? want : 1), perm(std::max<size_t>(1, n * d));
            uint32_t got = 0;
            if (n > 1 && d > 0)
              check(kpop_splits_gaps(embeddings.data.data(), (uint32_t)n, (uint32_t)d, want, &got, gap.data(), dim.data(), idx.data(), perm.data()));
            for (uint32_t s = 0; s < got; ++s) {  // :594-598
              Splits::Split sp;
              sp.weight = gap[s];
              sp.members.assign(perm.begin() + (size_t)dim[s] * n, perm.begin() + (size_t)dim[s] * n + idx[s] + 1);
              std::sort(sp.members.begin(), sp.members.end());
              splits.splits.push_back(std::move(sp));
            }
          }
          break;
        }
        case Action::EmbeddingsFromTwisted: {  // bin/KPopTwistDB.ml:494-498
          need_gpu();
          std::vector<double> mv = metric_vector(metric, T);
          if (mv.size() != twisted.cols()) throw Error("Incompatible_geometries");  // lib/Matrix.ml:81-82
          embeddings = twisted;
          if (!twisted.data.empty())
            check(kpop_embeddings(twisted.data.data(), (uint32_t)twisted.rows(), (uint32_t)twisted.cols(), mv.data(), distance.kind,
                                  distance.p, distance_normalize ? 1 : 0, embeddings.data.data()));
          break;
        }
        case Action::TablesToRegister:
          if (a.reg == Reg::Twister) load_twister_tables(T, a.s1);
          else if (a.reg == Reg::Twisted) twisted = read_table(make_filename(a.s1, "KPopTwisted", true));
          else if (a.reg == Reg::Embeddings) twisted = read_table(make_filename(a.s1, "KPopVectors", true));  // sic, bin/KPopTwistDB.ml:474-475
          else if (a.reg == Reg::Distances) distances = read_table(make_filename(a.s1, "KPopDMatrix", true));
          break;
        case Action::AddTablesToRegister:
          if (a.reg == Reg::Twisted) merge_rowwise(twisted, read_table(make_filename(a.s1, "KPopTwisted", true)));
          else if (a.reg == Reg::Distances) merge_rowwise(distances, read_table(make_filename(a.s1, "KPopDMatrix", true)));
          else if (a.reg == Reg::Embeddings) merge_rowwise(embeddings, read_table(make_filename(a.s1, "KPopVectors", true)));
          else throw Error("nothing can be added to the twister, metrics or splits registers");
          break;
        case Action::SetKmersNormalize: kmers_normalize = a.flag; break;
        case Action::AddKmersFiles: add_twisted_from_files(T, twisted, a.files, kmers_normalize, verbose); break;
        case Action::SetPrecision: precision = (int)a.num; break;
        case Action::SetDistance: distance = distance_of_string(a.s1); break;
        case Action::SetDistanceNormalize: distance_normalize = a.flag; break;
        case Action::SetMetric: metric = metric_of_string(a.s1); break;
        case Action::SetSummaryKeepAtMost: keep_at_most = a.num; break;
        case Action::RegisterToTables:
          if (a.reg == Reg::Twister) {  // Twister.to_files, lib/Twister.ml:28-30
            T.need_matrix();
            write_table(make_filename(a.s1, "KPopTwister", true), T.twister, precision);
            write_table(make_filename(a.s1, "KPopInertia", true), T.inertia, precision);
          } else if (a.reg == Reg::Twisted) {
            write_table(make_filename(a.s1, "KPopTwisted", true), twisted, precision);
          } else if (a.reg == Reg::Distances) {
            write_table(make_filename(a.s1, "KPopDMatrix", true), distances, precision);
          } else if (a.reg == Reg::Metrics) {  // get_metrics_matrix, lib/Twister.ml:210-217
            Table m;
            m.row_names = {"metrics"};
            m.col_names = T.inertia.col_names;
            {
              const std::vector<double> mv = metric_vector(metric, T);
              m.data.assign(mv.begin(), mv.end());
            }
            write_table(make_filename(a.s1, "KPopMetrics", true), m, precision);
          } else if (a.reg == Reg::Embeddings) {
            write_table(make_filename(a.s1, "KPopVectors", true), embeddings, precision);
          } else {  // Trees.Splits.to_file ~precision:!precision_splits, bin/KPopTwistDB.ml:534-535 (format declared in splits.h)
            write_splits(make_filename(a.s1, "PhyloSplits", true), splits, precision_splits);
          }
          break;
        case Action::DistancesFromTwisted: {  // bin/KPopTwistDB.ml:542-546
          Table m2 = load_twisted_operand(a.s1);
          if (twisted.col_names != m2.col_names) throw Error("Incompatible_geometries");  // lib/Matrix.ml:193-194
          need_gpu();
          std::vector<double> mv = metric_vector(metric, T);
          if (mv.size() != twisted.cols()) throw Error("Incompatible_vector_lengths");
          Table dm;
          dm.col_names = twisted.row_names;  // lib/Matrix.ml:264-266
          dm.row_names = m2.row_names;
          dm.data.assign(dm.rows() * dm.cols(), 0.);
          check((kpop_device_slots() > 1 ? kpop_sharded_distance_rowwise : kpop_distance_rowwise)(twisted.data.data(), (uint32_t)twisted.rows(), m2.data.data(), (uint32_t)m2.rows(),
                                      (uint32_t)twisted.cols(), mv.data(), distance.kind, distance.p, distance_normalize ? 1 : 0,
                                      dm.data.data()));
          distances = dm;
          break;
        }
        case Action::SummaryFromTwisted:
        case Action::SummaryFromDistances: {
          need_gpu();
          const bool from_tw = a.kind == Action::SummaryFromTwisted;
          Table m2;
          if (from_tw) {
            m2 = load_twisted_operand(a.s1);
            if (twisted.col_names != m2.col_names) throw Error("Incompatible_geometries");  // lib/Matrix.ml:698-699
            stage_mark("KPopTwistDB", "  twisted operand read");
          }
          const std::vector<std::string> &rows = from_tw ? m2.row_names : distances.row_names;
          const std::vector<std::string> &cols = from_tw ? twisted.row_names : distances.col_names;
          const uint32_t r1 = (uint32_t)cols.size(), r2 = (uint32_t)rows.size();
          uint32_t stride = keep_at_most == 0 ? r1 : (uint32_t)std::min<long long>(r1, keep_at_most + 2);  // widened below if a tie group is larger
          stride = std::max(stride, 1u);
          std::vector<double> stats((size_t)r2 * 4), dist, z;
          std::vector<uint32_t> n(r2), idx;
          for (;;) {  // widen the neighbour stride if some row has a larger tie group
            idx.assign((size_t)r2 * stride, 0);
            dist.assign((size_t)r2 * stride, 0.);
            z.assign((size_t)r2 * stride, 0.);
            if (from_tw) {
              std::vector<double> mv = metric_vector(metric, T);
              check((kpop_device_slots() > 1 ? kpop_sharded_distance_summary : kpop_distance_summary)(twisted.data.data(), r1, m2.data.data(), r2, (uint32_t)twisted.cols(), mv.data(),
                                          distance.kind, distance.p, distance_normalize ? 1 : 0, (uint32_t)keep_at_most, stride,
                                          stats.data(), n.data(), idx.data(), dist.data(), z.data()));
            } else {
              check(kpop_summarize_distances(distances.data.data(), r2, r1, (uint32_t)keep_at_most, stride, stats.data(), n.data(),
                                             idx.data(), dist.data(), z.data()));
            }
            uint32_t mx = 0;
            for (uint32_t v : n) mx = std::max(mx, v);
            if (mx <= stride) break;
            stride = mx;
          }
          stage_mark("KPopTwistDB", "  summaries computed");
          write_summary(make_filename(from_tw ? a.s2 : a.s1, "KPopSummary", true), rows, cols, stats, n, idx, dist, z, stride,
                        quote_summary);
          stage_mark("KPopTwistDB", "  summary text written");
          break;
        }
      }
    }
  } catch (const std::exception &e) {
    fprintf(stderr, "(KPopTwistDB): FATAL: Uncaught exception: %s\n", e.what());
    if (g_warm.joinable()) g_warm.join();
    T.reset();
    return 1;
  }
  stage_mark("KPopTwistDB", "last action done");
  if (g_warm.joinable()) g_warm.join();
  // Everything is written and closed.  Returning would take the registers apart value by value and the HIP runtime with
  // them (0.25-0.4 s with 4M rows in memory): the process ends here instead and the system takes it all back at once.
  fflush(stdout);
  fflush(stderr);
  _exit(0);
}

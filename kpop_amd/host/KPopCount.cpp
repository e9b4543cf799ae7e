// KPopCount -- drop-in for the reference's bin/KPopCount.ml on the MI355X hot path.
//
// Same options, same spectra text on stdout / <prefix>.KPopSpectra.txt
// (bin/KPopCount.ml:108-212, 26-63).  The per-read hashing/counting loop
// (:36-50) runs in libkpop_hip.so; this file is argument parsing, FASTA/FASTQ
// reading and printf.
//
// Differences from the reference, on purpose:
//   * k-mers inside a spectrum are printed in ascending hash order (the
//     reference prints in Hashtbl order, which is unspecified; every consumer
//     keys by name: lib/Twister.ml:151, lib/KMerDB.ml:536-562);
//   * -M is accepted but nothing is ever spilled: the -l spectrum is always
//     fully merged (the reference spills partial tables with repeated hashes
//     that consumers re-sum, bin/KPopCount.ml:39-50);
//   * a runtime failure exits with status 1 (the reference's DB tools print
//     the exception and exit 0);
//   * `-L` with stdout on a pipe whose reader is this repository's KPopTwistDB: the counting is deferred to that
//     process (fast_seq.h, "the reads stream"): the linted reads cross the pipe instead of ~9 bytes of text per k-mer,
//     and KPopTwistDB counts and twists them in one kernel.  Every other reader gets the text (KPOP_PIPE_FORMAT=text
//     forces it);
//   * protein k-mers use the residue encoding declared in csrc/kmer.h (the reference's lives in the absent BiOCamLib);
//     Sequences.Lint.proteinize is taken as dnaize's twin: upper-case, dashes and blanks dropped.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kpop_hip.h"
#include "fast_seq.h"
#include "kpop_text.h"

using namespace kpop_host;

namespace {

struct Input {
  SeqFormat fmt;
  std::string a, b;  // b non-empty: paired-end
};

struct Params {
  int k = 12;                     // bin/KPopCount.ml:88
  uint64_t max_results = 16777216;  // :89
  int content = KPOP_DNA_DS;      // :87
  std::vector<Input> inputs;
  bool have_l_or_L = false;
  std::string label;
  std::string output;
  bool verbose = false;
};

const char *kVersion = "18-hip";

void usage(FILE *f) {
  fprintf(f,
          "This is KPopCount (MI355X/HIP hot path) version %s\n"
          "Usage: KPopCount -l <output_vector_label>|-L [OPTIONS]\n"
          " -k|-K|--k-mer-size|--k-mer-length <k>   k-mer length (1..30 for DNA, 1..12 for protein; default 12)\n"
          " -M|--max-results-size <n>               accepted for compatibility (nothing is spilled)\n"
          " -C|--content DNA-ss|DNA-single-stranded|DNA-ds|DNA-double-stranded|protein   (default DNA-ds)\n"
          " -f|--fasta <file>                       FASTA input (repeatable)\n"
          " -s|--single-end <file>                  FASTQ input (repeatable)\n"
          " -p|--paired-end <file1> <file2>         paired FASTQ input (repeatable)\n"
          " -l|--label <label>                      one spectrum for all input, with this label\n"
          " -L|--one-spectrum-per-sequence          one spectrum per sequence, labelled by its name\n"
          " -o|--output <prefix>                    write <prefix>.KPopSpectra.txt (default stdout)\n"
          " -v|--verbose  -V|--version  -h|--help\n",
          kVersion);
}

[[noreturn]] void parse_error(const std::string &msg) {
  usage(stderr);
  fprintf(stderr, "(KPopCount): ERROR: %s\n", msg.c_str());
  exit(1);
}

void check(int rc) {
  if (rc != 0) throw Error(std::string("libkpop_hip: ") + kpop_last_error());
}

struct Merged {  // running -l spectrum: ascending hashes
  std::vector<uint64_t> hash;
  std::vector<uint64_t> count;
  void add(const uint64_t *h, const uint32_t *c, uint64_t n) {
    std::vector<uint64_t> nh, nc;
    nh.reserve(hash.size() + n);
    nc.reserve(hash.size() + n);
    size_t i = 0, j = 0;
    while (i < hash.size() || j < n) {
      if (j == n || (i < hash.size() && hash[i] < h[j])) {
        nh.push_back(hash[i]);
        nc.push_back(count[i]);
        ++i;
      } else if (i == hash.size() || h[j] < hash[i]) {
        nh.push_back(h[j]);
        nc.push_back(c[j]);
        ++j;
      } else {
        nh.push_back(hash[i]);
        nc.push_back(count[i] + c[j]);
        ++i;
        ++j;
      }
    }
    hash.swap(nh);
    count.swap(nc);
  }
};

struct Sink {  // where a batch of reads goes: the GPU and the text writer, or the reads stream
  const Params &P;
  FILE *out = nullptr;
  bool decided = false, stream = false, gpu = false;
  Merged merged;
  std::vector<uint64_t> offsets;
  // Per-read spectra are written as text by a thread of their own while the next block is counted: two sets of result
  // buffers, one being filled by the library, one being printed.
  struct Counted {
    std::vector<uint64_t> oo;
    std::vector<uint64_t, DefaultInitAlloc<uint64_t>> oh;  // (one slot per base: sized for the worst case, filled by the library --
    std::vector<uint32_t, DefaultInitAlloc<uint32_t>> oc;  //  no zero-fill of 12 bytes per base first)
    std::vector<std::string> labels;
  } sets[2];
  int filling = 0;
  std::thread writer;
  std::string writer_error;
  explicit Sink(const Params &p) : P(p) {}
  ~Sink() {
    if (writer.joinable()) writer.join();
  }
  void wait_for_writer() {
    if (writer.joinable()) writer.join();
    if (!writer_error.empty()) throw Error(writer_error);
  }

  void decide() {
    decided = true;
    const bool per_read = P.label.empty();
    stream = per_read && P.output.empty() && stdout_reader_is_dropin_twistdb();
    if (stream) {
#ifdef F_SETPIPE_SZ
      (void)fcntl(1, F_SETPIPE_SZ, 1 << 20);  // fewer, larger hand-overs to the reader (the default is 64 KB)
#endif
      fflush(stdout);
      ReadStreamHeader h;
      h.k = (uint32_t)P.k;
      h.content = (uint32_t)P.content;
      write_stream_header(1, h);
      if (P.verbose) fprintf(stderr, "(KPopCount): the reader of stdout is KPopTwistDB: handing it the reads, it counts them itself\n");
    }
  }
  void need_gpu() {
    if (gpu) return;
    int dev = 0;
    if (const char *e = getenv("KPOP_DEVICE")) dev = atoi(e);
    check(kpop_init(dev));
    gpu = true;
  }
  void process(const FlatBatch &b) {
    if (b.size() == 0) return;
    if (!decided) decide();
    if (stream) {
      write_stream_block(1, b);
      return;
    }
    need_gpu();
    const bool per_read = P.label.empty();
    const size_t n = b.size();
    offsets.resize(n + 1);
    offsets[0] = 0;
    for (size_t r = 0; r < n; ++r) offsets[r + 1] = offsets[r] + b.lens[r];
    const uint64_t cap = b.bases.size() + 1;
    Counted &c = sets[filling];
    c.oh.resize(cap);
    c.oc.resize(cap);
    c.oo.assign(per_read ? n + 1 : 2, 0);
    static const uint8_t dummy = 0;
    check(kpop_count_reads(b.bases.empty() ? &dummy : b.bases.data(), offsets.data(), (uint32_t)n, P.k, P.content,
                           per_read ? 1 : 0, c.oh.data(), c.oc.data(), c.oo.data(), cap));
    if (per_read) {  // bin/KPopCount.ml:44-46
      c.labels.resize(n);
      size_t at = 0;
      for (size_t r = 0; r < n; ++r) {
        c.labels[r] = strip_external_quotes_and_check(std::string(b.tags.data() + at, b.tag_lens[r]));
        at += b.tag_lens[r];
      }
      wait_for_writer();  // the block before this one is on its way out; its buffers are free again after this
      const int digits = name_digits(P.k, P.content == KPOP_PROTEIN);
      FILE *f = out;
      Counted *job = &c;
      writer = std::thread([this, f, job, digits] {
        try {
          write_spectra_parallel(f, job->labels, job->oh.data(), job->oc.data(), job->oo.data(), digits);
        } catch (const std::exception &e) {
          writer_error = e.what();
        }
      });
      filling ^= 1;
    } else {
      merged.add(c.oh.data(), c.oc.data(), c.oo[1]);
    }
  }
  void finish() {
    wait_for_writer();
    if (stream) write_stream_end(1);
  }
};

}  // namespace

int main(int argc, char **argv) {
  Params P;
  auto need = [&](int &i, const char *opt) -> std::string {
    if (i + 1 >= argc) parse_error(std::string("Option '") + opt + "' needs a parameter");
    return argv[++i];
  };
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-k" || a == "-K" || a == "--k-mer-size" || a == "--k-mer-length") {
      P.k = atoi(need(i, a.c_str()).c_str());
      if (P.k <= 0) parse_error("k-mer length must be positive");
    } else if (a == "-M" || a == "--max-results-size") {
      long long v = atoll(need(i, a.c_str()).c_str());
      if (v <= 0) parse_error("maximum results size must be positive");
      P.max_results = (uint64_t)v;
    } else if (a == "-C" || a == "--content") {
      const std::string c = need(i, a.c_str());
      if (c == "DNA-ss" || c == "DNA-single-stranded") P.content = KPOP_DNA_SS;
      else if (c == "DNA-ds" || c == "DNA-double-stranded") P.content = KPOP_DNA_DS;
      else if (c == "protein" || c == "prot") P.content = KPOP_PROTEIN;
      else parse_error("Invalid_content(\"" + c + "\")");
    } else if (a == "-f" || a == "--fasta") {
      P.inputs.push_back({SeqFormat::FASTA, need(i, a.c_str()), ""});
    } else if (a == "-s" || a == "--single-end") {
      P.inputs.push_back({SeqFormat::FASTQ, need(i, a.c_str()), ""});
    } else if (a == "-p" || a == "--paired-end") {
      std::string f1 = need(i, a.c_str());
      std::string f2 = need(i, a.c_str());
      P.inputs.push_back({SeqFormat::FASTQ, f1, f2});
    } else if (a == "-l" || a == "--label") {
      P.have_l_or_L = true;
      try {
        P.label = strip_external_quotes_and_check(need(i, a.c_str()));
      } catch (const Error &) {
        parse_error("Spectrum labels must not contain quotes");  // bin/KPopCount.ml:171
      }
    } else if (a == "-L" || a == "--one-spectrum-per-sequence") {
      P.have_l_or_L = true;
    } else if (a == "-o" || a == "--output") {
      P.output = make_filename(need(i, a.c_str()), "KPopSpectra", true);
    } else if (a == "-v" || a == "--verbose") {
      P.verbose = true;
    } else if (a == "-V" || a == "--version") {
      printf("%s\n", kVersion);
      return 0;
    } else if (a == "-h" || a == "--help") {
      usage(stdout);
      return 1;  // bin/KPopCount.ml:211
    } else {
      parse_error("Unknown option '" + a + "'");
    }
  }
  if (!P.have_l_or_L) parse_error("One of options '-l' and '-L' is mandatory");  // :213-214
  if (P.content == KPOP_PROTEIN ? P.k > 12 : P.k > 30) parse_error("k-mer length must be <= 30 for DNA or <= 12 for protein");  // :113
  for (size_t i = 1; i < P.inputs.size(); ++i)
    if (P.inputs[i].fmt != P.inputs[0].fmt) parse_error("You cannot process FASTA and FASTQ inputs together");  // :236
  if (P.inputs.empty()) return 0;  // :218

  try {
    FILE *out = P.output.empty() ? stdout : fopen(P.output.c_str(), "wb");
    if (!out) throw Error("cannot write '" + P.output + "'");
    std::vector<char> iobuf(1 << 22);
    setvbuf(out, iobuf.data(), _IOFBF, iobuf.size());
    if (!P.label.empty()) fprintf(out, "\t%s\n", P.label.c_str());  // :33-34
    stage_mark("KPopCount", "start");
    Sink sink(P);
    sink.out = out;
    Merged &merged = sink.merged;
    FlatBatch batch;
    uint64_t n_reads = 0;
    for (const Input &in : P.inputs) {
      if (in.b.empty()) {
        FastSeqReader rd(in.a, in.fmt);
        while (rd.next(batch)) {
          stage_mark("KPopCount", "block parsed");
          n_reads += batch.size();
          sink.process(batch);
          stage_mark("KPopCount", "block handed on");
        }
      } else {  // mates alternate: segment 0, segment 1, ... (bin/KPopCount.ml:36-54)
        // both files through the block reader (records linted by the threads), their records dealt alternately: a record at
        // a time through the line reader, as this was, paired-end input ran at 0.9 M reads/s
        FastSeqReader r1(in.a, in.fmt), r2(in.b, in.fmt);
        struct Side {
          FlatBatch b;
          size_t rec = 0, base = 0, tag = 0;  // the next record and where its bases and tag start
          bool eof = false;
          size_t left() const { return b.size() - rec; }
        } s1, s2;
        auto refill = [](FastSeqReader &r, Side &s) {
          if (s.left() || s.eof) return;
          s.rec = s.base = s.tag = 0;
          if (!r.next(s.b)) {
            s.b.clear();
            s.eof = true;
          }
        };
        batch.clear();
        for (;;) {
          refill(r1, s1);
          refill(r2, s2);
          if (s1.eof != s2.eof) throw Error("paired-end files '" + in.a + "' and '" + in.b + "' have different numbers of reads");
          if (s1.eof) break;
          const size_t n = std::min(s1.left(), s2.left());
          for (size_t i = 0; i < n; ++i)
            for (Side *s : {&s1, &s2}) {
              const uint32_t bl = s->b.lens[s->rec], tl = s->b.tag_lens[s->rec];
              batch.bases.insert(batch.bases.end(), s->b.bases.begin() + (long)s->base, s->b.bases.begin() + (long)(s->base + bl));
              batch.lens.push_back(bl);
              batch.tags.insert(batch.tags.end(), s->b.tags.begin() + (long)s->tag, s->b.tags.begin() + (long)(s->tag + tl));
              batch.tag_lens.push_back(tl);
              s->base += bl;
              s->tag += tl;
              ++s->rec;
            }
          n_reads += n;
          if (batch.size() >= (1u << 20) || batch.bases.size() >= (256ull << 20)) {
            sink.process(batch);
            batch.clear();
          }
        }
        sink.process(batch);
        batch.clear();
      }
    }
    sink.finish();
    if (!P.label.empty()) {  // final dump, bin/KPopCount.ml:60
      std::vector<uint32_t> c32(merged.count.size());
      for (size_t i = 0; i < c32.size(); ++i) {
        if (merged.count[i] > 0x7FFFFFFFull) throw Error("k-mer count exceeds 2^31-1");
        c32[i] = (uint32_t)merged.count[i];
      }
      write_spectrum_body(out, merged.hash.data(), c32.data(), c32.size(), name_digits(P.k, P.content == KPOP_PROTEIN));
    }
    if (P.verbose) fprintf(stderr, "(KPopCount): Added %llu reads.\n", (unsigned long long)n_reads);
    if (out != stdout) {
      if (fclose(out) != 0) throw Error("cannot write '" + P.output + "'");
    } else if (fflush(out) != 0)
      throw Error("cannot write to the standard output");
    stage_mark("KPopCount", "output closed");
    // everything is written: end here rather than take the HIP runtime and the buffers apart piece by piece (0.3 s)
    fflush(stderr);
    _exit(0);
  } catch (const std::exception &e) {
    fprintf(stderr, "(KPopCount): FATAL: %s\n", e.what());
    return 1;
  }
  return 0;
}

// fast_seq.h -- block-wise FASTA/FASTQ input for the drop-ins (Files.ReadsIterate + Sequences.Lint.dnaize,
// bin/KPopCount.ml:36,242-245), and the reads stream KPopCount hands to KPopTwistDB when the two drop-ins sit
// on either end of a pipe.
//
// Same record and linting rules as SeqReader (kpop_text.h), which stays as the line-by-line statement of them and
// as the reader of paired-end mates; tests/host/seq_diff.cpp holds the two to identical output.  What differs is
// the mechanics: the file is taken in blocks of tens of MB, cut at record boundaries, and the records of a block
// are linted by several threads into flat arrays (no std::string per read).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "kpop_text.h"

namespace kpop_host {

struct FlatBatch {
  std::vector<uint8_t> bases;      // linted, concatenated
  std::vector<uint32_t> lens;      // bases per read
  std::vector<char> tags;          // concatenated, no separators
  std::vector<uint32_t> tag_lens;  // bytes per tag
  size_t size() const { return lens.size(); }
  void clear() {
    bases.clear();
    lens.clear();
    tags.clear();
    tag_lens.clear();
  }
  void append(const FlatBatch &o);
};

class FastSeqReader {
 public:
  FastSeqReader(const std::string &path, SeqFormat fmt);
  ~FastSeqReader();
  // Replaces the contents of `out` with the records of the next block (about block_bytes of text, always whole
  // records, at least one).  false at end of file.
  bool next(FlatBatch &out, size_t block_bytes = 64u << 20, unsigned threads = 0);

 private:
  struct Impl;
  Impl *p_;
};

// ---------------------------------------------------------------------------------------------------------------
// The reads stream.  In `KPopCount -L ... | KPopTwistDB -i T x -k /dev/stdin ...` (README.md:606) the text spectra
// exist only to cross the pipe: ~9 bytes per k-mer, written by one tool and parsed back by the other.  When the
// reader of KPopCount's stdout is this repository's KPopTwistDB, KPopCount defers the counting to it: it sends the
// linted reads and their names, and KPopTwistDB counts and twists them in one fused kernel (kpop_count_twist).
// The spectra that would have crossed are a pure function of (read, k, content), so the twisted rows are the ones
// the text path gives -- tests/test_gpu_cli.py compares the two byte for byte.  Any other reader gets text.
//
//   stream := magic "\0KPopRd1" | u32 k | u32 content | block* | u32 0 u32 0
//   block  := u32 n_reads (>0) | u32 0 | u64 n_bases | u64 tag_bytes | u32 len[n] | u32 tag_len[n] | tags | bases
// Little endian.  The leading NUL can start no line of a spectra file (names hold no \000, README.md:744).
constexpr char kReadStreamMagic[8] = {'\0', 'K', 'P', 'o', 'p', 'R', 'd', '1'};

struct ReadStreamHeader {
  uint32_t k = 0, content = 0;
};
void write_stream_header(int fd, const ReadStreamHeader &h);
void write_stream_block(int fd, const FlatBatch &b);
void write_stream_end(int fd);

class ReadStreamReader {  // the consuming side, on a descriptor whose first 8 bytes (the magic) have been taken off
 public:
  explicit ReadStreamReader(int fd);
  ReadStreamHeader header;
  bool next(FlatBatch &out);  // false after the end marker
 private:
  int fd_;
};

// Is the process reading our stdout this repository's KPopTwistDB (same directory as this executable)?  Looks the
// pipe up in /proc; waits a little for a reader that has been forked but has not exec'ed yet.  KPOP_PIPE_FORMAT=text|
// reads overrides the answer.
bool stdout_reader_is_dropin_twistdb();

void write_all(int fd, const void *p, size_t n);
bool read_all(int fd, void *p, size_t n);  // false on a clean EOF at the first byte; throws on a short read

}  // namespace kpop_host

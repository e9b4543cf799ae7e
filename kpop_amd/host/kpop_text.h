// kpop_text.h -- the reference's text formats and sequence input, host side.
//
// Everything here is byte/text plumbing around the C ABI of include/kpop_hip.h;
// no arithmetic of the hot path lives in this directory.  file:line citations
// are into the reference checkout.
#pragma once
#include <stdint.h>

#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace kpop_host {

struct Error : std::runtime_error {
  explicit Error(const std::string &m) : std::runtime_error(m) {}
};

void *huge_block_alloc(size_t bytes);         // nullptr: not available, use the ordinary allocator
bool huge_block_free(void *p, size_t bytes);  // false: p did not come from huge_block_alloc

// A vector of doubles whose resize() leaves the new elements uninitialised: the big matrices (a million twisted rows are
// half a GB) are filled by whoever sized them, and a zero-fill first would touch every page one more time.
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = DefaultInitAlloc<U>;
  };
  DefaultInitAlloc() = default;
  template <class U>
  DefaultInitAlloc(const DefaultInitAlloc<U> &) {}
  template <class U, class... A>
  void construct(U *p, A &&...a) {
    if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
    else ::new ((void *)p) U(std::forward<A>(a)...);
  }
  // Large blocks are mapped on their own, 2 MB-aligned, and offered to transparent huge pages: whoever fills them (a copy
  // from the device, the threads laying rows) takes one page fault per 2 MB instead of one per 4 KB, and giving them back is
  // as much cheaper.  KPOP_HUGE_PAGES=0 turns it off.
  T *allocate(size_t n) {
    if (n * sizeof(T) >= kHugeBlock) {
      if (void *p = huge_block_alloc(n * sizeof(T))) return static_cast<T *>(p);
    }
    return std::allocator<T>::allocate(n);
  }
  void deallocate(T *p, size_t n) {
    if (n * sizeof(T) >= kHugeBlock && huge_block_free(p, n * sizeof(T))) return;
    std::allocator<T>::deallocate(p, n);
  }
  static constexpr size_t kHugeBlock = 32u << 20;
};

using DVec = std::vector<double, DefaultInitAlloc<double>>;

// KPOP_TIMING=1: "<tool>: <stage> +<seconds since the previous mark> (<seconds since the first>)" on stderr
void stage_mark(const char *tool, const char *stage);

// fn(lo, hi) over [0, n) cut into contiguous pieces, one per host thread (KPOP_HOST_THREADS overrides the count)
void parallel_for(size_t n, size_t min_per_thread, const std::function<void(size_t, size_t)> &fn);

// ---- names ---------------------------------------------------------------
// Matrix.Base.strip_external_quotes_and_check (BiOCamLib; call sites bin/KPopCount.ml:45,169,
// lib/Twister.ml:110): drop one surrounding pair of double quotes, then refuse any remaining quote.
std::string strip_external_quotes_and_check(const std::string &s);

// automatic file naming: prefix + "." + type [+ ".txt"], "/dev/..." verbatim
// (lib/Matrix.ml:309-320, lib/Twister.ml:219-221, lib/KMerDB.ml:28-30)
std::string make_filename(const std::string &prefix, const std::string &type_name, bool table);

// k-mer hash <-> hex name (bin/KPopCount.ml:46; encoding declared in csrc/kmer.h)
std::string hash_to_hex(uint64_t h, int k);
bool hex_to_hash(const std::string &s, uint64_t *h);

// ---- sequence input (Files.ReadsIterate + Sequences.Lint.dnaize; bin/KPopCount.ml:36,242-245) ----
struct ReadBatch {
  std::vector<uint8_t> bases;     // linted, concatenated
  std::vector<uint64_t> offsets;  // n+1
  std::vector<std::string> tags;  // n
  void clear() {
    bases.clear();
    offsets.assign(1, 0);
    tags.clear();
  }
  size_t size() const { return tags.size(); }
};

enum class SeqFormat { FASTA, FASTQ };

// Streams records of one file into batches.  dnaize ~keep_lowercase:false ~keep_dashes:false:
// letters are upper-cased, '-' and white space are dropped; anything that is not ACGT stays in place
// (and breaks the k-mer window in the kernels).
class SeqReader {
 public:
  SeqReader(const std::string &path, SeqFormat fmt);
  ~SeqReader();
  // appends up to max_bases / max_reads to `out`; returns false at end of file with nothing appended
  bool next_batch(ReadBatch &out, uint64_t max_bases, uint64_t max_reads);
  // one record (for interleaving paired-end mates); false at EOF
  bool next_record(std::string &tag, std::string &seq);

 private:
  struct Impl;
  Impl *p_;
};

// ---- spectra text (Appendix A.1; writer bin/KPopCount.ml:34,45-46,60; parser lib/Twister.ml:91-145) ----
struct Spectra {  // CSR over spectra, lines in file order
  std::vector<std::string> labels;
  std::vector<uint64_t> offsets{0};
  std::vector<std::string> names;  // k-mer names, opaque to every consumer
  std::vector<double> values;      // float_of_string (lib/Twister.ml:155)
};
// Parses one file (or /dev/stdin) appending to `out`.  Errors mirror the reference's exceptions:
// Wrong_number_of_columns (:103-104), Header_expected (:106-107), Float_expected (:155-157).
void read_spectra_file(const std::string &path, Spectra &out);
// hex digits of a k-mer name: ceil(2k/4) for DNA, ceil(5k/4) for protein (encodings declared in csrc/kmer.h)
inline int name_digits(int k, bool protein) { return protein ? (5 * k + 3) / 4 : (k + 1) / 2; }
void write_spectrum(FILE *f, const std::string &label, const uint64_t *hash, const uint32_t *count, uint64_t n, int digits);

// The same parser for the GPU path: k-mer names are turned into hashes while parsing (a name that is not `name_len`
// hexadecimal digits can match no twister column and becomes `absent`), no per-line strings are made, and the file
// is cut at line boundaries and parsed by several threads.  Same errors, same line numbers.
struct HashedSpectra {
  std::vector<std::string> labels;
  std::vector<uint64_t> offsets{0};
  std::vector<uint64_t, DefaultInitAlloc<uint64_t>> hash;  // (sized, then filled by threads: no zero-fill first -- it
  DVec values;                                             //  was half of the parsing time of a block)
};
// The twister's columns as the parser needs them (lib/Twister.ml:71-76, a Hashtbl from column name to column).  Given to the
// parsers below it makes them say exactly what the reference says: a name is looked up as a STRING (so "0A" is not "0a"), and a
// value that is not a float is an error only on a line whose name IS a column (float_of_string sits inside `Some idx`, :153-157).
// Two forms.  hex: every column name is name_len lowercase hexadecimal digits (what KPopCount writes), a k-mer's number is the
// value of its name and `columns` holds the columns' numbers in ascending order.  opaque: names are any strings, a k-mer's number
// is its column's index through `index` (the last of several columns of one name, Hashtbl.add shadows).
struct KmerLookup {
  bool opaque = false;
  const std::vector<uint64_t> *columns = nullptr;
  const std::unordered_map<std::string_view, uint64_t> *index = nullptr;
};
void read_spectra_hashed(const std::string &path, size_t name_len, uint64_t absent, HashedSpectra &out, unsigned threads = 0);
// the same on an open descriptor whose first head_len bytes the caller has already taken off (to look at them)
void read_spectra_hashed_fd(int fd, const char *head, size_t head_len, size_t name_len, uint64_t absent, HashedSpectra &out,
                            unsigned threads = 0);

// Text spectra from a descriptor a block at a time, so that reading, parsing and twisting overlap (a million read spectra
// are 1.26 GB of text: read whole, then parsed, then twisted was 2.1 s of which none overlapped).  A block ends where a
// spectrum ends -- before the last line that begins with a tab, a header (lib/Twister.ml:106-111) -- and is block_bytes
// long or as much longer as its last spectrum needs.
using TextBlock = std::vector<char, DefaultInitAlloc<char>>;
class SpectraTextStream {
 public:
  SpectraTextStream(int fd, const char *head, size_t head_len, size_t block_bytes = 64u << 20);
  bool next(TextBlock &block);  // false once the input is exhausted
 private:
  int fd_;
  size_t block_bytes_;
  TextBlock carry_;
  bool eof_ = false;
};
// One block of that stream -> hashed spectra (`out` is overwritten).  first_block: the stream's first line must be a header
// (Header_expected otherwise); lines_before: the lines of the blocks before this one, for the line numbers of
// Wrong_number_of_columns; *n_lines receives this block's.  The same errors, in the same order, as the whole-file parser.
void parse_spectra_block(const char *data, size_t size, size_t name_len, uint64_t absent, bool first_block, uint64_t lines_before,
                         HashedSpectra &out, uint64_t *n_lines, unsigned threads = 0, bool *plain = nullptr,
                         const KmerLookup *lookup = nullptr);
// *plain: every data line of the block was name_len LOWERCASE hexadecimal digits, a tab, 1-15 decimal digits and a newline, and
// no line carried a CR -- what KPopCount writes; KPopCountDB takes such blocks through this parser and the others line by line.

// "\t<label>\n" + "<hex>\t<count>\n"... for reads [0, n) of a CSR result, formatted by several threads and written in
// order (bin/KPopCount.ml:44-46).  labels must already be checked.
void write_spectra_parallel(FILE *f, const std::vector<std::string> &labels, const uint64_t *hash, const uint32_t *count,
                            const uint64_t *offsets, int digits, unsigned threads = 0);
void write_spectrum_body(FILE *f, const uint64_t *hash, const uint32_t *count, uint64_t n, int digits);

// ---- matrix tables (Appendix A.2; README.md:618-626,643-650; src/KPopTwist:100,108,116) ----
struct Table {
  std::vector<std::string> col_names, row_names;
  DVec data;  // row-major rows x cols
  size_t rows() const { return row_names.size(); }
  size_t cols() const { return col_names.size(); }
  bool empty() const { return row_names.empty() && col_names.empty(); }
};
Table read_table(const std::string &path);
void write_table(const std::string &path, const Table &t, int precision);
// Matrix.merge_rowwise (BiOCamLib; call site lib/Matrix.ml:331-334): same columns, rows appended
void merge_rowwise(Table &into, const Table &add);

// Row bookkeeping of Twister.add_twisted_from_files (lib/Twister.ml:78-82,189-204) without a tree of a million strings:
// `labels` are the row names in arrival order, the first n_existing of them rows the register already held (among those a
// repeated name keeps its last row, StringMap.add); a later label that is already present is the reference's
// Duplicate_label, raised for the first such row in arrival order.  Returns the arrival numbers of the surviving rows in
// bytewise label order.
std::vector<uint32_t> order_rows_by_label(const std::vector<std::string> &labels, size_t n_existing);

std::string format_g(double x, int precision);  // "%.*g"
// The same characters printf("%.*g") gives, appended to `out`: std::to_chars (shortest-path digit generation, about 4x
// the speed of glibc's multi-precision printf; equal on every value tried, tests/host/format_g_check.cpp).
void append_g(std::string &out, double x, int precision);

// Rows [0, n) of a text file: formatted by the host threads a slab of rows at a time (fmt appends row r's text, newline
// included, to `out`) and written in row order by a thread of their own while the next slab is being formatted.
void write_rows_parallel(FILE *f, const std::string &path, size_t n, size_t reserve_per_row,
                         const std::function<void(size_t row, std::string &out)> &fmt);

}  // namespace kpop_host

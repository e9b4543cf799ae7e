// counter_db.h -- the k-mer database register of KPopCountDB (lib/KMerDB.ml), host side.
//
// Name tables, metadata, selection logic and text output live here; every loop that touches all counts
// (statistics, class combination, table transformations, spectral distances) is a call into libkpop_hip.so
// (kpop_counter_*, kpop_distance_rowwise).  file:line citations are into the reference checkout.
#pragma once
#include <stdint.h>

#include <set>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "kpop_text.h"
#include "ocaml_marshal.h"

namespace kpop_host {

// Transformation.parameters_t (lib/KMerDB.ml:145-163)
struct Transform {
  std::string which = "power";
  double threshold = 1., power = 1.;
  int code() const;  // KPOP_TRANSF_*; throws Unknown_transformation
};

// TableFilter.t (lib/KMerDB.ml:985-1008)
struct TableFilter {
  bool print_row_names = true, print_col_names = true, print_metadata = false, transpose = false;
  Transform transform;
  bool print_zero_rows = false;
  std::set<std::string> filter_columns;
  int precision = 15;
};

// (metadata field, Str regexp) pairs; an empty field matches the label (bin/KPopCountDB.ml:81-93)
using RegexpSelector = std::vector<std::pair<std::string, std::string>>;
RegexpSelector parse_regexp_selector(const std::string &s);
// Str.string_match re s 0 (anchored at the start, not at the end)
bool str_string_match(const std::string &str_regexp, const std::string &s);

class CounterDB {
 public:
  CounterCore core;  // storage columns may be physically shorter than n_rows (implicit trailing zeros) between calls
  bool verbose = false;

  size_t n_cols() const { return core.col_names.size(); }
  size_t n_rows() const { return core.row_names.size(); }
  size_t n_meta() const { return core.meta_names.size(); }

  static CounterDB of_binary(const std::string &prefix);                 // lib/KMerDB.ml:414-430
  void to_binary(const std::string &prefix);                             // :395-413
  void add_meta(const std::string &fname);                               // :433-504
  void add_files(const std::vector<std::string> &prefixes);              // :507-588
  std::set<std::string> selected_from_regexps(const RegexpSelector &re) const;  // :590-623
  std::set<std::string> selected_negate(const std::set<std::string> &sel) const;  // :624-625
  void add_combined_selected(const std::string &new_label, const std::set<std::string> &selection, int criterion);  // :639-736
  void split_spectra(const std::string &classes_label, int criterion);   // :787-813
  void remove_selected(const std::set<std::string> &selected);           // :766-785
  void output_summary() const;                                           // :241-266
  void to_table(const TableFilter &filter, const std::string &prefix);   // :1012-1169
  void to_spectra(const TableFilter &filter, const std::string &prefix); // :1170-1236
  // :1237-1279; kind/p as in kpop_distance_rowwise
  void to_distances(int kind, double p, bool normalise, const std::set<std::string> &sel1, const std::set<std::string> &sel2,
                    const std::string &prefix);

  // every spectrum padded to n_rows; returns one pointer per spectrum
  std::vector<const int32_t *> columns();

 private:
  std::unordered_map<std::string, uint32_t> col_idx_, row_idx_, meta_idx_;
  // k-mer names that are 1..15 lowercase hexadecimal digits (what KPopCount writes) are indexed by (length << 60 | value)
  // instead of by string: one integer hash per spectrum line instead of a string allocation and a string hash.  Every
  // other name lives in row_idx_; a name is in exactly one of the two.
  // (open addressing: a look-up per spectrum line, tens of millions of them for a thousand genomes)
  struct HexIndex {
    std::vector<uint64_t> keys;  // 0 = empty slot (a key is never 0: the length sits in its top bits)
    std::vector<uint32_t> rows;
    size_t used = 0;
    void clear() {
      keys.clear();
      rows.clear();
      used = 0;
    }
    void reserve(size_t n) {
      size_t cap = 64;
      while (cap < 2 * n + 2) cap <<= 1;
      if (cap > keys.size()) rehash(cap);
    }
    static size_t slot_of(uint64_t k, size_t mask) { return (size_t)((k * 0x9E3779B97F4A7C15ull) >> 20) & mask; }
    void rehash(size_t cap) {
      std::vector<uint64_t> ok;
      std::vector<uint32_t> orow;
      ok.swap(keys);
      orow.swap(rows);
      keys.assign(cap, 0);
      rows.assign(cap, 0);
      const size_t mask = cap - 1;
      for (size_t i = 0; i < ok.size(); ++i)
        if (ok[i]) {
          size_t s = slot_of(ok[i], mask);
          while (keys[s]) s = (s + 1) & mask;
          keys[s] = ok[i];
          rows[s] = orow[i];
        }
    }
    // the row of `key`, or, when it is new, `next_row` entered under it (and *added set)
    uint32_t find_or_add(uint64_t key, uint32_t next_row, bool *added) {
      if (2 * (used + 1) > keys.size()) rehash(keys.empty() ? 64 : keys.size() * 2);
      const size_t mask = keys.size() - 1;
      size_t s = slot_of(key, mask);
      while (keys[s] && keys[s] != key) s = (s + 1) & mask;
      if (keys[s]) {
        *added = false;
        return rows[s];
      }
      keys[s] = key;
      rows[s] = next_row;
      ++used;
      *added = true;
      return next_row;
    }
    static constexpr uint32_t kNone = 0xFFFFFFFFu;
    uint32_t find(uint64_t key) const {  // (read-only: safe from several threads while nobody adds)
      if (keys.empty()) return kNone;
      const size_t mask = keys.size() - 1;
      size_t s = slot_of(key, mask);
      while (keys[s] && keys[s] != key) s = (s + 1) & mask;
      return keys[s] ? rows[s] : kNone;
    }
    void set(uint64_t key, uint32_t row) {  // Hashtbl.add: the last one entered under a name is the one found
      bool added;
      (void)find_or_add(key, row, &added);
      if (!added) {
        const size_t mask = keys.size() - 1;
        size_t s = slot_of(key, mask);
        while (keys[s] != key) s = (s + 1) & mask;
        rows[s] = row;
      }
    }
  } hex_row_idx_;
  void add_spectra_text(int fd, const std::string &fname);  // one file of add_files
  static bool hex_key(const char *s, size_t n, uint64_t *key);
  uint32_t row_of(const char *name, size_t len);  // the row of a k-mer name, appended if new
  void rebuild_indices();
  uint32_t add_empty_column_if_needed(const std::string &label);  // :376-391
};

}  // namespace kpop_host

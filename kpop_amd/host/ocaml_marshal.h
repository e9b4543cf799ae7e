// ocaml_marshal.h -- reading and writing the OCaml Marshal (output_value) stream of the reference's binary
// registers: '.KPopTwister', '.KPopTwisted', '.KPopDMatrix' (lib/Matrix.ml:812-845, lib/Twister.ml:222-246).
//
// A binary register is a sequence of independent marshalled values:
//     "KPop<Type>" (string), "2022-04-03" (string), Matrix.Base.t (record)
// and '.KPopTwister' holds two such triples (twister, then inertia).  Matrix.Base.t is declared in BiOCamLib
// (not part of the reference checkout); its field order is taken as { col_names; row_names; data } with
// data : Float.Array.t array -- the order of every record literal in the reference (lib/Matrix.ml:188-190,
// 264-266; lib/Twister.ml:205-206).  UNPINNED: no OCaml toolchain is available here to produce a real file.
//
// Wire format (OCaml runtime, extern.c / intern.c; identical in 4.x and 5.x for these codes): 20-byte header
// {magic 0x8495A6BE, data_len, num_objects, size_32, size_64} (or the 32-byte 0x8495A6BF form), then a
// pre-order stream of small-int / small-block / small-string prefixes and CODE_* bytes; shared references
// (CODE_SHARED8/16/32) index the objects read so far and are honoured on input.  Output never shares.
#pragma once
#include <stdio.h>

#include <stdint.h>

#include <string>
#include <vector>

#include "kpop_text.h"

namespace kpop_host {

constexpr const char *kArchiveVersion = "2022-04-03";  // lib/Matrix.ml:812

// Matrix.to_channel / of_channel (lib/Matrix.ml:814-817,828-833).  type_name is "KPopTwisted" etc.
void marshal_write_matrix(FILE *f, const std::string &type_name, const Table &t);
// returns false at a clean end of file; throws on anything malformed
bool marshal_read_matrix(FILE *f, std::string *type_name, Table *t);

// whole-file helpers with the reference's checks (Unexpected_type, Incompatible_archive_version)
Table read_binary_matrix(const std::string &path, const std::string &expect_type);
void write_binary_matrix(const std::string &path, const std::string &type_name, const Table &t);
void read_binary_twister(const std::string &path, Table *twister, Table *inertia);
void read_binary_twister_inertia(const std::string &path, Table *inertia);  // skips over the twister matrix
void write_binary_twister(const std::string &path, const Table &twister, const Table &inertia);

// ---- '.KPopCounter' (lib/KMerDB.ml:54-63,389-430): "KPopCounter", "2022-04-03", then the record
//   { n_cols; n_rows; n_meta; idx_to_col_names; idx_to_row_names; idx_to_meta_names; meta; storage }
// with storage : I32BAVector.t array.  BAVector is BiOCamLib's (absent); it is taken as a plain one-dimensional
// int32 c_layout Bigarray -- UNPINNED like the matrix field order.  A Bigarray travels as a custom block
// (identifier "_bigarr02": dimension count, kind|layout flags, dimensions, big-endian elements).
struct CounterCore {
  std::vector<std::string> col_names, row_names, meta_names;
  std::vector<std::vector<std::string>> meta;  // n_cols x n_meta
  std::vector<std::vector<int32_t>> storage;   // n_cols x n_rows
};
void write_binary_counter(const std::string &path, const CounterCore &db);
CounterCore read_binary_counter(const std::string &path);

}  // namespace kpop_host

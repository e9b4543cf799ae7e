/*
 * kpop_oracle.h -- CPU restatement of KPop's count -> twist -> distance path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under kpop_amd/ may include, link or
 * dlopen this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - distance / norms / summary : follow in-repo reference source line by
 *     line (lib/Space.ml:150-205, lib/Matrix.ml:42-76,191-266,632-690) and
 *     are PINNED by the README known-answer test (README.md:649 -> :660).
 *   - twist : lib/Twister.ml:146-188 is followed literally; the inner sparse
 *     mat-vec lives in BiOCamLib (github PaoloRibeca/BiOCamLib, un-vendored
 *     submodule, no pinned commit: .gitmodules:1-3) -- restated from its
 *     published algorithm (row-outer, IntMap-ascending inner loop).
 *     Mathematically determined (t = T.x); order of summation unpinned.
 *   - count : the k-mer encoding lives in BiOCamLib KMers (absent).
 *     PARITY UNPINNED: the encoding below is declared by this repository
 *      and is swappable in one place (kpo_base_code).
 *     Only structural pin: README.md:106 (DNA-ds collapses a k-mer with its
 *     reverse complement: 512 = 4^5/2 rows per spectrum at k=5).
 *   - metric "powers" : lib/Space.ml:88-105 delegates to BiOCamLib
 *     Numbers.Frequencies.Vector (absent); semantics inferred from the
 *     comment at lib/Space.ml:98-102.  PARITY UNPINNED.
 */
#ifndef KPOP_ORACLE_H
#define KPOP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* content modes: bin/KPopCount.ml:66-82 */
#define KPO_DNA_DS 0
#define KPO_DNA_SS 1
#define KPO_PROTEIN 2 /* counting only; k <= 12; encoding declared in kpop_amd/csrc/kmer.h (PARITY UNPINNED) */

/* distance kinds: lib/Space.ml:140-143 */
#define KPO_EUCLIDEAN 0
#define KPO_COSINE 1
#define KPO_MINKOWSKI 2

/* ---- synthetic inputs: SplitMix64, counter-addressable ---- */
uint64_t kpo_mix64(uint64_t z);
/* n-th output (n >= 0) of the SplitMix64 stream started at `seed` */
uint64_t kpo_splitmix_at(uint64_t seed, uint64_t n);
/* read r, base i = "ACGT"[stream(r*read_len+i) >> 62]; bases is n_reads*read_len */
void kpo_synth_reads(uint64_t seed, uint64_t n_reads, uint32_t read_len, uint8_t *bases);
/* twister coefficient for dimension d and k-mer hash h, in [-1,1) */
double kpo_synth_twister_coeff(uint64_t seed, uint32_t d, uint64_t h);
/* inertia w_d ~ 2^(-d/8), sum 1 */
void kpo_synth_inertia(uint32_t n_dims, double *w);
/* all canonical (content=DS) or all (SS) k-mer hashes, ascending; returns count.
   out may be NULL to query the count. */
uint64_t kpo_enumerate_kmers(int k, int content, uint64_t *out);
/* dims-major synthetic twister for the given column hashes */
void kpo_synth_twister(uint64_t seed, uint32_t n_dims, const uint64_t *col_hash, uint64_t n_cols,
                       double *T_dims_major);

/* ---- count: restates KIH.iterc + KIHF (bin/KPopCount.ml:36-50) ---- */
/* 0..3 for ACGT (either case), -1 otherwise */
int kpo_base_code(uint8_t c);
int kpo_protein_code(uint8_t c);
void kpo_to_hex_protein(uint64_t hash, int k, char *out /* >= 17 bytes */);
/* hex name of a hash for k: zero padded lowercase, ceil(k/2) digits (bin/KPopCount.ml:46) */
void kpo_to_hex(uint64_t hash, int k, char *out /* >= 17 bytes */);
/* one read -> unique (hash,count) ascending by hash.  Returns n_unique, or -1 if cap too small. */
int64_t kpo_count_read(const uint8_t *seq, uint64_t len, int k, int content,
                       uint64_t *out_hash, uint32_t *out_count, uint64_t cap);
/* CSR over reads (per_read=1, the -L mode) or one merged spectrum (per_read=0, -l mode).
   out_offsets has n_reads+1 entries (per_read=1) or 2 (per_read=0). Returns 0 / -1 (capacity). */
int kpo_count_reads(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content,
                    int per_read, uint64_t *out_hash, uint32_t *out_count, uint64_t *out_offsets,
                    uint64_t out_capacity);

/* ---- twist: restates lib/Twister.ml:146-188 ---- */
/* spectra given as CSR of (hash, value) in file line order; T dims-major [n_dims][n_cols];
   col_hash[c] = k-mer of twister column c.  out is n_spectra x n_dims row-major. */
int kpo_twist(const double *T_dims_major, uint64_t n_cols, uint32_t n_dims, const uint64_t *col_hash,
              const uint64_t *hash, const double *value, const uint64_t *offsets, uint32_t n_spectra,
              int normalize, double *out);

/* ---- metric: lib/Space.ml:88-105 ---- */
void kpo_metric_flat(uint32_t n, double *out);
void kpo_metric_powers(const double *inertia, uint32_t n, double power_int, double threshold,
                       double power_ext, double *out);

/* ---- distance: lib/Space.ml:150-205, lib/Matrix.ml:42-76,191-266 ---- */
double kpo_norm(int kind, double p, const double *metric, const double *v, uint32_t n);
void kpo_normalizations(int kind, double p, const double *metric, const double *m, uint32_t rows,
                        uint32_t n_dims, double *out);
double kpo_distance(int kind, double p, const double *metric, const double *a, double na,
                    const double *b, double nb, uint32_t n);
/* out is r2 x r1 row-major: out[j*r1+i] = d(m1[i], m2[j])  (lib/Matrix.ml:253) */
void kpo_distance_rowwise(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                          const double *metric, int kind, double p, int normalize, double *out);

/* ---- embeddings: Base.get_embeddings, lib/Matrix.ml:78-128; out is rows x n_dims ---- */
void kpo_embeddings(const double *m, uint32_t rows, uint32_t n_dims, const double *metric, int kind, double p,
                    int normalize, double *out);

/* ---- summary: lib/Matrix.ml:632-690 ---- */
/* row of n distances -> stats[4] = mean, sd, median, MAD; neighbours (ties extend).
   out_idx/out_dist/out_z need room for n entries. Returns eff_len. */
uint32_t kpo_summarize_row(const double *row, uint32_t n, uint32_t req_len, double *stats,
                           uint32_t *out_idx, double *out_dist, double *out_z);
/* lib/Matrix.ml:691-766: never materialises r2 x r1.  out_stats r2x4; neighbours CSR. */
int kpo_distance_summary(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                         const double *metric, int kind, double p, int normalize, uint32_t keep_at_most,
                         double *out_stats, uint64_t *out_offsets /* r2+1 */, uint32_t *out_idx,
                         double *out_dist, double *out_z, uint64_t capacity);

/* ---- whole pipeline on `threads` OpenMP threads, for bench.py's cpu_baseline ----
   count (-L) -> twist -> rowwise distance vs classes.  twisted: n_reads x n_dims,
   dist: n_reads x n_classes.  Returns seconds of wall time spent. */
double kpo_pipeline(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content,
                    const double *T_dims_major, uint64_t n_cols, uint32_t n_dims, const uint64_t *col_hash,
                    const double *classes, uint32_t n_classes, const double *metric, int kind, double p,
                    int normalize_counts, int normalize_distance, int threads, double *twisted,
                    double *dist);

/* ---- k-mer database (KPopCountDB): lib/KMerDB.ml ----
   storage = n_cols spectra ("columns") of n_rows int32 counts each, one pointer per spectrum
   (`storage: I32BAVector.t array`, lib/KMerDB.ml:54-63).
   PARITY UNPINNED where it leans on the absent BiOCamLib: FreqVector.sum is taken as the running sum in
   insertion order, FreqVector.median as the upper median sorted[n/2] (the convention of lib/Matrix.ml:660-668),
   0 for an empty histogram. */
#define KPO_TRANSF_BINARY 0
#define KPO_TRANSF_POWER 1
#define KPO_TRANSF_CLR 2
#define KPO_TRANSF_PSEUDO 3
/* stats_table_of_core_db.compute_one (lib/KMerDB.ml:174-215) for one vector of n counts read with `stride`:
   stats[4] = non_zero, max, sum, sum_log */
void kpo_counter_vector_stats(const int32_t *v, uint64_t n, uint64_t stride, double threshold, double power,
                              double *stats);
/* column statistics (col_stats n_cols x 4) and row statistics (row_stats n_rows x 4; may be NULL) */
void kpo_counter_stats(const int32_t *const *columns, uint32_t n_cols, uint64_t n_rows, double threshold,
                       double power, double *col_stats, double *row_stats);
/* Transformation.compute (lib/KMerDB.ml:96-144) */
double kpo_counter_transform_one(int which, double threshold, double power, const double *col_stats,
                                 int32_t counts);
/* add_combined_selected (lib/KMerDB.ml:628-736): columns[sel[0..n_sel)] visited in that order; col_sum[c] =
   the linear statistic sum of column c (power 1, threshold 1); criterion 0 = mean, 1 = median.
   out[n_rows] receives Int32.of_float of the combination; returns the accumulated norm (:715). */
double kpo_counter_combine(const int32_t *const *columns, uint64_t n_rows, const uint32_t *sel, uint32_t n_sel,
                           const double *col_sum, int criterion, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif

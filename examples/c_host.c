/* c_host.c -- the C ABI of libkpop_hip.so from a plain C host (no Python, no C++): the calls an OCaml/ctypes binding would
 * make, in the order the reference's tools make them (bin/KPopCount.ml:36-50 -> lib/Twister.ml:146-188 ->
 * lib/Matrix.ml:191-266,691-766).  Built and run by tests/test_gpu_abi_c.py:
 *     gcc -O2 -std=c99 -Iinclude examples/c_host.c -Lkpop_amd -lkpop_hip -Wl,-rpath,$PWD/kpop_amd -lm -o c_host
 * Prints the counts of one read, its twisted row and its distances to two reference rows, deterministically. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kpop_hip.h"

#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc_ = (call);                                                                \
    if (rc_ != 0) {                                                                  \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, kpop_last_error());      \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

int main(void) {
  CHECK(kpop_init(0));
  /* three reads, concatenated; k = 3, DNA-ds */
  const char *reads[3] = {"ACGTACGTTGCA", "TTTTTTTT", "ACGNNACGTAC"};
  uint8_t bases[64];
  uint64_t offsets[4] = {0, 0, 0, 0};
  size_t at = 0;
  for (int r = 0; r < 3; ++r) {
    memcpy(bases + at, reads[r], strlen(reads[r]));
    at += strlen(reads[r]);
    offsets[r + 1] = at;
  }
  const int k = 3;
  uint64_t hash[64], out_off[4];
  uint32_t count[64];
  CHECK(kpop_count_reads(bases, offsets, 3, k, KPOP_DNA_DS, 1, hash, count, out_off, 64));
  for (int r = 0; r < 3; ++r) {
    printf("read %d:", r);
    for (uint64_t i = out_off[r]; i < out_off[r + 1]; ++i) printf(" %02llx=%u", (unsigned long long)hash[i], count[i]);
    printf("\n");
  }
  /* a twister over every canonical 3-mer (32 of them), 2 dimensions, dims-major as the reference holds it */
  uint64_t cols[64];
  int n_cols = 0;
  for (uint64_t h = 0; h < 64; ++h) {
    uint64_t rc = 0, x = h;
    for (int j = 0; j < k; ++j) {
      rc = (rc << 2) | (3 - (x & 3));
      x >>= 2;
    }
    if (h <= rc) cols[n_cols++] = h;
  }
  double *T = (double *)malloc(sizeof(double) * 2 * (size_t)n_cols);
  for (int c = 0; c < n_cols; ++c) {
    T[c] = (double)(c + 1) / 8.0;            /* dimension 0 */
    T[n_cols + c] = (double)(n_cols - c) / 4.0; /* dimension 1 */
  }
  kpop_twister *tw = NULL;
  CHECK(kpop_twister_load(T, (uint64_t)n_cols, 2, cols, k, &tw));
  double twisted[6], fused[6], exact[6];
  double values[64];
  for (uint64_t i = 0; i < out_off[3]; ++i) values[i] = (double)count[i];
  CHECK(kpop_twist(tw, hash, values, out_off, 3, 1, twisted));
  CHECK(kpop_count_twist(tw, bases, offsets, 3, KPOP_DNA_DS, 1, fused));
  CHECK(kpop_spectra_twist(tw, bases, offsets, 3, k, KPOP_DNA_DS, 1, exact));
  for (int r = 0; r < 3; ++r)
    printf("twisted %d: %.15g %.15g | fused %.15g %.15g | spectra_twist %s\n", r, twisted[2 * r], twisted[2 * r + 1], fused[2 * r],
           fused[2 * r + 1], memcmp(exact + 2 * r, twisted + 2 * r, 16) == 0 ? "identical" : "DIFFERENT");
  /* distances of the three twisted rows (the operand) to two reference rows (the register) */
  const double inertia[2] = {0.75, 0.25};
  double metric[2], ref[4] = {1.0, 2.0, 3.0, 1.0}, dist[6];
  CHECK(kpop_metric_compute(KPOP_METRIC_POWERS, inertia, 2, 1.0, 1.0, 2.0, metric));
  CHECK(kpop_distance_rowwise(ref, 2, twisted, 3, 2, metric, KPOP_EUCLIDEAN, 2.0, 1, dist));
  for (int r = 0; r < 3; ++r) printf("distances %d: %.15g %.15g\n", r, dist[2 * r], dist[2 * r + 1]);
  double stats[12], nd[6], nz[6];
  uint32_t nn[3], idx[6];
  CHECK(kpop_distance_summary(ref, 2, twisted, 3, 2, metric, KPOP_EUCLIDEAN, 2.0, 1, 1, 2, stats, nn, idx, nd, nz));
  for (int r = 0; r < 3; ++r) printf("summary %d: mean %.15g closest %u at %.15g\n", r, stats[4 * r], idx[2 * r], nd[2 * r]);
  /* the same three reads through the streaming pipeline: page-locked buffers, one call, distances and summary only
   * (the twisted rows never cross the bus); must agree with the separate calls above to the last bit */
  {
    uint8_t *pb = NULL;
    uint64_t *po = NULL;
    double *pd = NULL, *ps = NULL, *pnd = NULL, *pnz = NULL;
    uint32_t *pn = NULL, *pi = NULL;
    CHECK(kpop_host_alloc((void **)&pb, sizeof bases));
    CHECK(kpop_host_alloc((void **)&po, sizeof offsets));
    CHECK(kpop_host_alloc((void **)&pd, sizeof dist));
    CHECK(kpop_host_alloc((void **)&ps, sizeof stats));
    CHECK(kpop_host_alloc((void **)&pn, sizeof nn));
    CHECK(kpop_host_alloc((void **)&pi, sizeof idx));
    CHECK(kpop_host_alloc((void **)&pnd, sizeof nd));
    CHECK(kpop_host_alloc((void **)&pnz, sizeof nz));
    memcpy(pb, bases, sizeof bases);
    memcpy(po, offsets, sizeof offsets);
    kpop_pipeline_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.content = KPOP_DNA_DS;
    cfg.normalize_counts = 1;
    cfg.kind = KPOP_EUCLIDEAN;
    cfg.p = 2.0;
    cfg.normalize_distances = 1;
    cfg.outputs = KPOP_OUT_DISTANCES | KPOP_OUT_SUMMARY;
    cfg.keep_at_most = 1;
    cfg.max_neighbours = 2;
    cfg.chunk_reads = 2; /* two chunks for three reads: the ring is exercised */
    kpop_pipeline *pl = NULL;
    CHECK(kpop_pipeline_create(tw, ref, 2, metric, &cfg, &pl));
    kpop_pipeline_outputs po_out;
    memset(&po_out, 0, sizeof po_out);
    po_out.distances = pd;
    po_out.stats = ps;
    po_out.n_neighbours = pn;
    po_out.nb_index = pi;
    po_out.nb_distance = pnd;
    po_out.nb_z = pnz;
    uint64_t ticket = 0;
    CHECK(kpop_pipeline_submit(pl, pb, po, 3, &po_out, &ticket));
    CHECK(kpop_pipeline_collect(pl, ticket));
    uint32_t chunks = 0;
    int pinned = 0;
    CHECK(kpop_pipeline_stats(pl, &chunks, &pinned, NULL));
    /* `fused` rows are what the pipeline twists (kpop_count_twist); its distances are checked by the caller of this program */
    int same = 1;
    for (int r = 0; r < 3; ++r) same = same && pn[r] == nn[r] && pi[2 * r] == idx[2 * r];
    printf("pipeline: %u chunks, pinned %d, neighbours %s\n", chunks, pinned, same ? "identical" : "DIFFERENT");
    for (int r = 0; r < 3; ++r) printf("pipeline distances %d: %.15g %.15g\n", r, pd[2 * r], pd[2 * r + 1]);
    CHECK(kpop_pipeline_destroy(pl));
    CHECK(kpop_host_free(pb));
    CHECK(kpop_host_free(po));
    CHECK(kpop_host_free(pd));
    CHECK(kpop_host_free(ps));
    CHECK(kpop_host_free(pn));
    CHECK(kpop_host_free(pi));
    CHECK(kpop_host_free(pnd));
    CHECK(kpop_host_free(pnz));
  }
  {
    /* the same three reads handed over PACKED (2 bits a base + a bit a base that is no base: include/kpop_hip.h, "packed bases"):
       the host packs once, 2.25 bits a base cross the bus, the rows are the byte entry point's bit for bit */
    const uint64_t n_bases = offsets[3];
    uint32_t *codes = (uint32_t *)calloc(kpop_packed_code_words(n_bases) + 1, 4), *invalid = (uint32_t *)calloc(kpop_packed_mask_words(n_bases) + 1, 4);
    double packed_rows[6];
    CHECK(kpop_pack_bases(bases, n_bases, codes, invalid, 1));
    CHECK(kpop_count_twist_packed(tw, codes, invalid, offsets, 3, KPOP_DNA_DS, 1, packed_rows));
    printf("packed: %llu bases in %llu + %llu words, rows %s\n", (unsigned long long)n_bases, (unsigned long long)kpop_packed_code_words(n_bases),
           (unsigned long long)kpop_packed_mask_words(n_bases), memcmp(packed_rows, fused, sizeof fused) == 0 ? "identical" : "DIFFERENT");
    free(codes);
    free(invalid);
  }
  CHECK(kpop_twister_free(tw));
  free(T);
  CHECK(kpop_shutdown());
  return 0;
}

/*
 * kpop_oracle.c -- CPU restatement of KPop's count -> twist -> distance path.
 * TEST INFRASTRUCTURE ONLY (see kpop_oracle.h for the pinning status of each part).
 *
 * Compiled with -ffp-contract=off: the reference is OCaml, whose float
 * arithmetic never fuses a multiply into an add.
 *
 * All file:line citations are into /root/reference.
 */
#include "kpop_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* synthetic inputs                                     */
/* ------------------------------------------------------------------ */

uint64_t kpo_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

uint64_t kpo_splitmix_at(uint64_t seed, uint64_t n) {
  /* SplitMix64: state_{n} = seed + (n+1)*gamma, output = mix(state) */
  return kpo_mix64(seed + (n + 1) * 0x9E3779B97F4A7C15ULL);
}

void kpo_synth_reads(uint64_t seed, uint64_t n_reads, uint32_t read_len, uint8_t *bases) {
  static const char acgt[4] = {'A', 'C', 'G', 'T'};
  uint64_t total = n_reads * (uint64_t)read_len;
  for (uint64_t i = 0; i < total; ++i) bases[i] = (uint8_t)acgt[kpo_splitmix_at(seed, i) >> 62];
}

double kpo_synth_twister_coeff(uint64_t seed, uint32_t d, uint64_t h) {
  uint64_t z = kpo_mix64(seed ^ ((uint64_t)d << 40) ^ h);
  /* 53 random bits -> [0,1) -> [-1,1) */
  return (double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

void kpo_synth_inertia(uint32_t n_dims, double *w) {
  double s = 0.;
  for (uint32_t d = 0; d < n_dims; ++d) {
    w[d] = exp2(-(double)d / 8.0);
    s += w[d];
  }
  for (uint32_t d = 0; d < n_dims; ++d) w[d] /= s;
}

static uint64_t revcomp_hash(uint64_t h, int k) {
  uint64_t r = 0;
  for (int i = 0; i < k; ++i) {
    r = (r << 2) | (3 - (h & 3));
    h >>= 2;
  }
  return r;
}

uint64_t kpo_enumerate_kmers(int k, int content, uint64_t *out) {
  uint64_t n = 1ULL << (2 * k), cnt = 0;
  for (uint64_t h = 0; h < n; ++h) {
    if (content == KPO_DNA_DS && revcomp_hash(h, k) < h) continue;
    if (out) out[cnt] = h;
    ++cnt;
  }
  return cnt;
}

void kpo_synth_twister(uint64_t seed, uint32_t n_dims, const uint64_t *col_hash, uint64_t n_cols,
                       double *T) {
#pragma omp parallel for schedule(static)
  for (int64_t d = 0; d < (int64_t)n_dims; ++d)
    for (uint64_t c = 0; c < n_cols; ++c)
      T[(uint64_t)d * n_cols + c] = kpo_synth_twister_coeff(seed, (uint32_t)d, col_hash[c]);
}

/* ------------------------------------------------------------------ */
/* count (declared encoding; SURVEY.md Appendix B -- parity unpinned)  */
/* ------------------------------------------------------------------ */

int kpo_base_code(uint8_t c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

void kpo_to_hex_protein(uint64_t hash, int k, char *out) {
  int digits = (5 * k + 3) / 4;
  static const char hx[] = "0123456789abcdef";
  for (int i = digits - 1; i >= 0; --i) {
    out[i] = hx[hash & 15];
    hash >>= 4;
  }
  out[digits] = 0;
}

void kpo_to_hex(uint64_t hash, int k, char *out) {
  int digits = (k + 1) / 2; /* ceil(2k/4) */
  static const char hx[] = "0123456789abcdef";
  for (int i = digits - 1; i >= 0; --i) {
    out[i] = hx[hash & 15];
    hash >>= 4;
  }
  out[digits] = 0;
}

static int cmp_u64(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return (x > y) - (x < y);
}

/* KIH.iterc res seq (bin/KPopCount.ml:38): every k-window free of non-ACGT
   symbols contributes one increment of its (canonical) hash. Returns number
   of keys written into keys (cap must be >= len). */
/* residue code of the protein hash declared by this repository (kpop_amd/csrc/kmer.h): the 20 standard amino acids
   in alphabetical order of their one-letter codes, either case; -1 breaks the window */
int kpo_protein_code(uint8_t c) {
  static const char order[] = "ACDEFGHIKLMNPQRSTVWY";
  if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 'a' + 'A');
  for (int i = 0; i < 20; ++i)
    if ((char)c == order[i]) return i;
  return -1;
}

static uint64_t iterc_protein(const uint8_t *seq, uint64_t len, int k, uint64_t *keys) {
  const uint64_t mask = (1ULL << (5 * k)) - 1;
  uint64_t fwd = 0, n = 0;
  int run = 0;
  for (uint64_t i = 0; i < len; ++i) {
    int c = kpo_protein_code(seq[i]);
    if (c < 0) {
      run = 0;
      fwd = 0;
      continue;
    }
    fwd = ((fwd << 5) | (uint64_t)c) & mask;
    if (++run >= k) keys[n++] = fwd;
  }
  return n;
}

static uint64_t iterc(const uint8_t *seq, uint64_t len, int k, int content, uint64_t *keys) {
  if (content == KPO_PROTEIN) return iterc_protein(seq, len, k, keys);
  uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  uint64_t fwd = 0, rc = 0, n = 0;
  int run = 0;
  for (uint64_t i = 0; i < len; ++i) {
    int c = kpo_base_code(seq[i]);
    if (c < 0) {
      run = 0;
      fwd = rc = 0;
      continue;
    }
    fwd = ((fwd << 2) | (uint64_t)c) & mask;
    rc = (rc >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
    if (++run >= k) keys[n++] = (content == KPO_DNA_DS && rc < fwd) ? rc : fwd;
  }
  return n;
}

static int64_t rle(uint64_t *keys, uint64_t n, uint64_t *out_hash, uint32_t *out_count, uint64_t cap) {
  qsort(keys, n, sizeof(uint64_t), cmp_u64);
  uint64_t u = 0;
  for (uint64_t i = 0; i < n;) {
    uint64_t j = i;
    while (j < n && keys[j] == keys[i]) ++j;
    if (u >= cap) return -1;
    out_hash[u] = keys[i];
    out_count[u] = (uint32_t)(j - i);
    ++u;
    i = j;
  }
  return (int64_t)u;
}

int64_t kpo_count_read(const uint8_t *seq, uint64_t len, int k, int content, uint64_t *out_hash,
                       uint32_t *out_count, uint64_t cap) {
  if (k < 1 || k > (content == KPO_PROTEIN ? 12 : 30)) return -2;
  uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (len ? len : 1));
  uint64_t n = iterc(seq, len, k, content, keys);
  int64_t u = rle(keys, n, out_hash, out_count, cap);
  free(keys);
  return u;
}

int kpo_count_reads(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content,
                    int per_read, uint64_t *out_hash, uint32_t *out_count, uint64_t *out_offsets,
                    uint64_t out_capacity) {
  if (k < 1 || k > (content == KPO_PROTEIN ? 12 : 30)) return -2;
  if (per_read) {
    /* -L: dump + clear after every read (bin/KPopCount.ml:39-50) */
    uint64_t pos = 0;
    out_offsets[0] = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
      int64_t u = kpo_count_read(bases + offsets[r], offsets[r + 1] - offsets[r], k, content,
                                 out_hash + pos, out_count + pos, out_capacity - pos);
      if (u < 0) return -1;
      pos += (uint64_t)u;
      out_offsets[r + 1] = pos;
    }
    return 0;
  }
  /* -l: one table for all reads, dumped at the end (bin/KPopCount.ml:60) */
  uint64_t total = offsets[n_reads] - offsets[0];
  uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (total ? total : 1));
  uint64_t n = 0;
  for (uint32_t r = 0; r < n_reads; ++r)
    n += iterc(bases + offsets[r], offsets[r + 1] - offsets[r], k, content, keys + n);
  int64_t u = rle(keys, n, out_hash, out_count, out_capacity);
  free(keys);
  if (u < 0) return -1;
  out_offsets[0] = 0;
  out_offsets[1] = (uint64_t)u;
  return 0;
}

/* ------------------------------------------------------------------ */
/* twist (lib/Twister.ml:146-188)                                      */
/* ------------------------------------------------------------------ */

typedef struct {
  uint64_t hash;
  uint64_t col;
} hc_t;

static int cmp_hc(const void *a, const void *b) {
  const hc_t *x = (const hc_t *)a, *y = (const hc_t *)b;
  if (x->hash != y->hash) return (x->hash > y->hash) - (x->hash < y->hash);
  return (x->col > y->col) - (x->col < y->col);
}

/* name -> column table (lib/Twister.ml:71-76). Hashtbl.add shadows earlier
   bindings, so find_opt returns the LAST column carrying a given name. */
static hc_t *build_index(const uint64_t *col_hash, uint64_t n_cols) {
  hc_t *ix = (hc_t *)malloc(sizeof(hc_t) * (n_cols ? n_cols : 1));
  int sorted = 1;
  for (uint64_t c = 0; c < n_cols; ++c) {
    ix[c].hash = col_hash[c];
    ix[c].col = c;
    if (c && col_hash[c] <= col_hash[c - 1]) sorted = 0;
  }
  if (!sorted) qsort(ix, n_cols, sizeof(hc_t), cmp_hc);
  return ix;
}

static int64_t find_col(const hc_t *ix, uint64_t n, uint64_t h) {
  uint64_t lo = 0, hi = n; /* upper bound of h */
  while (lo < hi) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (ix[mid].hash <= h) lo = mid + 1; else hi = mid;
  }
  if (lo == 0 || ix[lo - 1].hash != h) return -1;
  return (int64_t)ix[lo - 1].col;
}

typedef struct {
  uint64_t idx;
  uint64_t seq; /* arrival order, for a stable sort */
  double v;
} sv_t;

static int cmp_sv(const void *a, const void *b) {
  const sv_t *x = (const sv_t *)a, *y = (const sv_t *)b;
  if (x->idx != y->idx) return (x->idx > y->idx) - (x->idx < y->idx);
  return (x->seq > y->seq) - (x->seq < y->seq);
}

/* One spectrum: the `worker` closure of lib/Twister.ml:146-188. Lines arrive
   REVERSED (`rev_lines`, :142-144), so accumulation runs last line first. */
static void twist_one(const double *T, uint64_t n_cols, uint32_t n_dims, const hc_t *ix,
                      const uint64_t *hash, const double *value, uint64_t n_lines, int normalize,
                      double *out, sv_t *scratch) {
  double acc = 0.;
  uint64_t m = 0;
  for (uint64_t r = 0; r < n_lines; ++r) {
    uint64_t l = n_lines - 1 - r;
    int64_t c = find_col(ix, n_cols, hash[l]);
    if (c < 0) continue; /* :167-169 discarded, and excluded from acc */
    acc = acc + value[l]; /* :158 */
    scratch[m].idx = (uint64_t)c;
    scratch[m].seq = m;
    scratch[m].v = value[l];
    ++m;
  }
  qsort(scratch, m, sizeof(sv_t), cmp_sv);
  /* IntMap with repeated k-mers accumulated, vv +. v (:160-163) */
  uint64_t u = 0;
  for (uint64_t i = 0; i < m;) {
    uint64_t j = i + 1;
    double vv = scratch[i].v;
    while (j < m && scratch[j].idx == scratch[i].idx) {
      vv = vv + scratch[j].v;
      ++j;
    }
    scratch[u].idx = scratch[i].idx;
    scratch[u].v = vv;
    ++u;
    i = j;
  }
  if (normalize && acc != 0.) /* :177-178 */
    for (uint64_t i = 0; i < u; ++i) scratch[i].v = scratch[i].v / acc;
  /* BiOCamLib Matrix.multiply_matrix_sparse_vector_single_threaded (:183):
     for every matrix row d, fold the sparse vector in ascending index order. */
  for (uint32_t d = 0; d < n_dims; ++d) {
    const double *row = T + (uint64_t)d * n_cols;
    double a = 0.;
    for (uint64_t i = 0; i < u; ++i) a = a + row[scratch[i].idx] * scratch[i].v;
    out[d] = a;
  }
}

int kpo_twist(const double *T, uint64_t n_cols, uint32_t n_dims, const uint64_t *col_hash,
              const uint64_t *hash, const double *value, const uint64_t *offsets, uint32_t n_spectra,
              int normalize, double *out) {
  hc_t *ix = build_index(col_hash, n_cols);
  uint64_t maxl = 1;
  for (uint32_t s = 0; s < n_spectra; ++s)
    if (offsets[s + 1] - offsets[s] > maxl) maxl = offsets[s + 1] - offsets[s];
  sv_t *scratch = (sv_t *)malloc(sizeof(sv_t) * maxl);
  for (uint32_t s = 0; s < n_spectra; ++s)
    twist_one(T, n_cols, n_dims, ix, hash + offsets[s], value + offsets[s], offsets[s + 1] - offsets[s],
              normalize, out + (uint64_t)s * n_dims, scratch);
  free(scratch);
  free(ix);
  return 0;
}

/* ------------------------------------------------------------------ */
/* metric (lib/Space.ml:88-105)                                        */
/* ------------------------------------------------------------------ */

void kpo_metric_flat(uint32_t n, double *out) {
  /* :89-95 */
  for (uint32_t i = 0; i < n; ++i) out[i] = 1. / (double)n;
}

void kpo_metric_powers(const double *inertia, uint32_t n, double power_int, double threshold,
                       double power_ext, double *out) {
  /* :104-105  pow_abs pi |> threshold_accum_abs thr |> pow_abs pe |> normalize_abs
     (BiOCamLib Numbers.Frequencies.Vector: absent; inferred semantics, PARITY UNPINNED) */
  double total = 0.;
  for (uint32_t i = 0; i < n; ++i) {
    out[i] = pow(fabs(inertia[i]), power_int);
    total += fabs(out[i]);
  }
  /* keep leading elements until the accumulated |x| reaches threshold * total;
     everything after that point is zeroed (threshold 1 keeps all) */
  double run = 0., limit = threshold * total;
  for (uint32_t i = 0; i < n; ++i) {
    double a = fabs(out[i]);
    if (run >= limit) out[i] = 0.;
    run += a;
  }
  double s = 0.;
  for (uint32_t i = 0; i < n; ++i) {
    out[i] = pow(fabs(out[i]), power_ext);
    s += fabs(out[i]);
  }
  if (s != 0.)
    for (uint32_t i = 0; i < n; ++i) out[i] = out[i] / s;
}

/* ------------------------------------------------------------------ */
/* distance (lib/Space.ml:150-205)                                     */
/* ------------------------------------------------------------------ */

static double scale(int kind, double p, double x) {
  /* lib/Space.ml:159-165 */
  switch (kind) {
    case KPO_EUCLIDEAN: return sqrt(x);
    case KPO_COSINE: return x / 2.;
    default: return pow(x, 1. / p);
  }
}

double kpo_norm(int kind, double p, const double *metric, const double *v, uint32_t n) {
  /* compute_norm_unscaled :166-179 then scale :180-181 */
  double acc = 0.;
  if (kind == KPO_MINKOWSKI)
    for (uint32_t i = 0; i < n; ++i) acc = acc + (pow(fabs(v[i]), p) * metric[i]);
  else
    for (uint32_t i = 0; i < n; ++i) acc = acc + (v[i] * v[i] * metric[i]);
  return scale(kind, p, acc);
}

void kpo_normalizations(int kind, double p, const double *metric, const double *m, uint32_t rows,
                        uint32_t n_dims, double *out) {
  /* lib/Matrix.ml:42-76; zero norm becomes 1 (:67) */
  for (uint32_t i = 0; i < rows; ++i) {
    double nv = kpo_norm(kind, p, metric, m + (uint64_t)i * n_dims, n_dims);
    out[i] = (nv == 0.) ? 1. : nv;
  }
}

double kpo_distance(int kind, double p, const double *metric, const double *a, double na,
                    const double *b, double nb, uint32_t n) {
  /* compute_unscaled with adaptors a/na, b/nb (lib/Space.ml:182-203; lib/Matrix.ml:247-249) */
  double acc = 0.;
  if (kind == KPO_MINKOWSKI) {
    for (uint32_t i = 0; i < n; ++i) {
      double diff = fabs(a[i] / na - b[i] / nb);
      acc = acc + (pow(diff, p) * metric[i]);
    }
  } else {
    for (uint32_t i = 0; i < n; ++i) {
      double diff = a[i] / na - b[i] / nb;
      acc = acc + (diff * diff * metric[i]);
    }
  }
  return scale(kind, p, acc);
}

void kpo_distance_rowwise(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                          const double *metric, int kind, double p, int normalize, double *out) {
  /* lib/Matrix.ml:191-266 */
  double *n1 = (double *)malloc(sizeof(double) * (r1 ? r1 : 1));
  double *n2 = (double *)malloc(sizeof(double) * (r2 ? r2 : 1));
  if (normalize) {
    kpo_normalizations(kind, p, metric, m1, r1, n_dims, n1);
    kpo_normalizations(kind, p, metric, m2, r2, n_dims, n2);
  } else {
    for (uint32_t i = 0; i < r1; ++i) n1[i] = 1.;
    for (uint32_t j = 0; j < r2; ++j) n2[j] = 1.;
  }
  for (uint32_t j = 0; j < r2; ++j)
    for (uint32_t i = 0; i < r1; ++i)
      out[(uint64_t)j * r1 + i] = kpo_distance(kind, p, metric, m1 + (uint64_t)i * n_dims, n1[i],
                                               m2 + (uint64_t)j * n_dims, n2[j], n_dims);
  free(n1);
  free(n2);
}

/* ------------------------------------------------------------------ */
/* summary (lib/Matrix.ml:632-690)                                     */
/* ------------------------------------------------------------------ */

typedef struct {
  double d;
  uint32_t i;
} di_t;

static int cmp_di(const void *a, const void *b) {
  const di_t *x = (const di_t *)a, *y = (const di_t *)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return (x->i > y->i) - (x->i < y->i);
}

static int cmp_f64(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

uint32_t kpo_summarize_row(const double *row, uint32_t n, uint32_t req_len, double *stats,
                           uint32_t *out_idx, double *out_dist, double *out_z) {
  di_t *distr = (di_t *)malloc(sizeof(di_t) * (n ? n : 1));
  for (uint32_t c = 0; c < n; ++c) { /* :636-639 FloatIntMultimap.add dist col_idx */
    distr[c].d = row[c];
    distr[c].i = c;
  }
  qsort(distr, n, sizeof(di_t), cmp_di);
  uint32_t eff_len = 0;
  int64_t median_pos = n / 2;
  double median = 0., acc = 0.;
  for (uint32_t s = 0; s < n;) { /* :641-650 iter_set: one call per distinct distance */
    uint32_t e = s;
    while (e < n && distr[e].d == distr[s].d) ++e;
    int64_t set_len = e - s;
    double dist = distr[s].d;
    acc = acc + ((double)set_len * dist);
    if (median_pos >= 0 && median_pos - set_len < 0) median = dist;
    median_pos -= set_len;
    if (eff_len < req_len) eff_len += (uint32_t)set_len;
    s = e;
  }
  double mean = (n > 0) ? acc / (double)n : 0.; /* :651-655 */
  acc = 0.;
  double *dd = (double *)malloc(sizeof(double) * (n ? n : 1));
  for (uint32_t c = 0; c < n; ++c) { /* :659-670 */
    double d = row[c] - mean;
    acc = acc + (d * d);
    dd[c] = fabs(row[c] - median);
  }
  qsort(dd, n, sizeof(double), cmp_f64);
  median_pos = n / 2; /* :671-678 */
  double mad = 0.;
  for (uint32_t s = 0; s < n;) {
    uint32_t e = s;
    while (e < n && dd[e] == dd[s]) ++e;
    int64_t occs = e - s;
    if (median_pos >= 0 && median_pos - occs < 0) mad = dd[s];
    median_pos -= occs;
    s = e;
  }
  double stddev = (n > 1) ? sqrt(acc / ((double)n - 1.)) : 0.; /* :679-683 */
  stats[0] = mean;
  stats[1] = stddev;
  stats[2] = median;
  stats[3] = mad;
  for (uint32_t i = 0; i < eff_len && i < n; ++i) { /* :685-689 */
    out_idx[i] = distr[i].i;
    out_dist[i] = distr[i].d;
    out_z[i] = (distr[i].d - mean) / stddev;
  }
  free(dd);
  free(distr);
  return eff_len < n ? eff_len : n;
}

int kpo_distance_summary(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                         const double *metric, int kind, double p, int normalize, uint32_t keep_at_most,
                         double *out_stats, uint64_t *out_offsets, uint32_t *out_idx, double *out_dist,
                         double *out_z, uint64_t capacity) {
  /* lib/Matrix.ml:691-766 */
  double *n1 = (double *)malloc(sizeof(double) * (r1 ? r1 : 1));
  double *n2 = (double *)malloc(sizeof(double) * (r2 ? r2 : 1));
  double *row = (double *)malloc(sizeof(double) * (r1 ? r1 : 1));
  uint32_t *ti = (uint32_t *)malloc(sizeof(uint32_t) * (r1 ? r1 : 1));
  double *td = (double *)malloc(sizeof(double) * (r1 ? r1 : 1));
  double *tz = (double *)malloc(sizeof(double) * (r1 ? r1 : 1));
  if (normalize) {
    kpo_normalizations(kind, p, metric, m1, r1, n_dims, n1);
    kpo_normalizations(kind, p, metric, m2, r2, n_dims, n2);
  } else {
    for (uint32_t i = 0; i < r1; ++i) n1[i] = 1.;
    for (uint32_t j = 0; j < r2; ++j) n2[j] = 1.;
  }
  uint32_t req_len = keep_at_most ? keep_at_most : r1; /* :723-726 */
  uint64_t pos = 0;
  int rc = 0;
  out_offsets[0] = 0;
  for (uint32_t j = 0; j < r2; ++j) {
    for (uint32_t i = 0; i < r1; ++i) /* :744-749 */
      row[i] = kpo_distance(kind, p, metric, m1 + (uint64_t)i * n_dims, n1[i], m2 + (uint64_t)j * n_dims,
                            n2[j], n_dims);
    uint32_t e = kpo_summarize_row(row, r1, req_len, out_stats + (uint64_t)j * 4, ti, td, tz);
    if (pos + e > capacity) {
      rc = -1;
      break;
    }
    memcpy(out_idx + pos, ti, sizeof(uint32_t) * e);
    memcpy(out_dist + pos, td, sizeof(double) * e);
    memcpy(out_z + pos, tz, sizeof(double) * e);
    pos += e;
    out_offsets[j + 1] = pos;
  }
  free(n1); free(n2); free(row); free(ti); free(td); free(tz);
  return rc;
}

/* ------------------------------------------------------------------ */
/* whole pipeline, OpenMP over reads, for the CPU baseline             */
/* ------------------------------------------------------------------ */

static double now_s(void) {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.;
#endif
}

double kpo_pipeline(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content,
                    const double *T, uint64_t n_cols, uint32_t n_dims, const uint64_t *col_hash,
                    const double *classes, uint32_t n_classes, const double *metric, int kind, double p,
                    int normalize_counts, int normalize_distance, int threads, double *twisted,
                    double *dist) {
  hc_t *ix = build_index(col_hash, n_cols);
  uint64_t maxl = 1;
  for (uint32_t r = 0; r < n_reads; ++r)
    if (offsets[r + 1] - offsets[r] > maxl) maxl = offsets[r + 1] - offsets[r];
  double *n1 = (double *)malloc(sizeof(double) * (n_classes ? n_classes : 1));
  if (normalize_distance) kpo_normalizations(kind, p, metric, classes, n_classes, n_dims, n1);
  else for (uint32_t i = 0; i < n_classes; ++i) n1[i] = 1.;
  if (threads < 1) threads = 1;
  double t0 = now_s();
#pragma omp parallel num_threads(threads)
  {
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * maxl);
    uint64_t *uh = (uint64_t *)malloc(sizeof(uint64_t) * maxl);
    uint32_t *uc = (uint32_t *)malloc(sizeof(uint32_t) * maxl);
    double *uv = (double *)malloc(sizeof(double) * maxl);
    sv_t *scratch = (sv_t *)malloc(sizeof(sv_t) * maxl);
#pragma omp for schedule(dynamic, 64)
    for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
      /* count (-L): one spectrum per read */
      uint64_t n = iterc(bases + offsets[r], offsets[r + 1] - offsets[r], k, content, keys);
      int64_t u = rle(keys, n, uh, uc, maxl);
      /* the spectrum crosses a text pipe in the reference (counts printed %d,
         parsed float_of_string): integers, so the value is exact */
      for (int64_t i = 0; i < u; ++i) uv[i] = (double)uc[i];
      double *t = twisted + (uint64_t)r * n_dims;
      twist_one(T, n_cols, n_dims, ix, uh, uv, (uint64_t)u, normalize_counts, t, scratch);
      double n2 = 1.;
      if (normalize_distance) {
        n2 = kpo_norm(kind, p, metric, t, n_dims);
        if (n2 == 0.) n2 = 1.;
      }
      for (uint32_t i = 0; i < n_classes; ++i)
        dist[(uint64_t)r * n_classes + i] =
            kpo_distance(kind, p, metric, classes + (uint64_t)i * n_dims, n1[i], t, n2, n_dims);
    }
    free(keys); free(uh); free(uc); free(uv); free(scratch);
  }
  double t1 = now_s();
  free(n1);
  free(ix);
  return t1 - t0;
}

/* ------------------------------------------------------------------ k-mer database (lib/KMerDB.ml) */

void kpo_counter_vector_stats(const int32_t *v, uint64_t n, uint64_t stride, double threshold, double power,
                              double *stats) {
  /* lib/KMerDB.ml:182-195: the relative threshold is taken against the plain sum of v^power */
  double sum = 0.;
  for (uint64_t i = 0; i < n; ++i) sum += pow((double)v[i * stride], power);
  if (threshold < 1.) threshold *= sum;
  /* :196-215 */
  double non_zero = 0., max = 0., s = 0., sl = 0.;
  for (uint64_t i = 0; i < n; ++i) {
    const int32_t c = v[i * stride];
    const double f = (double)c;
    if (f >= threshold) {
      non_zero += 1.;
      if ((double)c > max) max = (double)c;
      s += pow(f, power);
      sl += log(f) * power;
    }
  }
  stats[0] = non_zero; stats[1] = max; stats[2] = s; stats[3] = sl;
}

void kpo_counter_stats(const int32_t *const *columns, uint32_t n_cols, uint64_t n_rows, double threshold,
                       double power, double *col_stats, double *row_stats) {
  for (uint32_t c = 0; c < n_cols; ++c)
    kpo_counter_vector_stats(columns[c], n_rows, 1, threshold, power, col_stats + 4 * (uint64_t)c);
  if (!row_stats) return;
  int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (n_cols ? n_cols : 1));
  for (uint64_t r = 0; r < n_rows; ++r) {
    for (uint32_t c = 0; c < n_cols; ++c) tmp[c] = columns[c][r];
    kpo_counter_vector_stats(tmp, n_cols, 1, threshold, power, row_stats + 4 * r);
  }
  free(tmp);
}

/* Stdlib.max on floats: `if a >= b then a else b` -- a NaN second argument comes back out */
static double ocaml_max(double a, double b) { return a >= b ? a : b; }

double kpo_counter_transform_one(int which, double threshold, double power, const double *cs, int32_t icounts) {
  const double counts = (double)icounts, non_zero = cs[0], cmax = cs[1], csum = cs[2], csum_log = cs[3];
  const double thr0 = threshold;
  if (threshold < 1.) threshold *= csum; /* lib/KMerDB.ml:98-104 */
  switch (which) {
    case KPO_TRANSF_BINARY: return counts >= threshold ? 1. : 0.;
    case KPO_TRANSF_POWER:
      if (power == 1.) return counts >= threshold ? counts : 0.;
      return counts >= threshold ? pow(counts, power) : 0.;
    case KPO_TRANSF_CLR: {
      double v = counts >= threshold ? counts : 0.;
      v = ocaml_max(v, 0.1); /* epsilon, :95 */
      return log(v) * power - csum_log / non_zero;
    }
    default: { /* pseudocounts, :130-144; a negative power raises Invalid_transformation there */
      (void)thr0;
      double v;
      if (power == 0.) v = cmax * log((counts + 1.) / threshold);
      else {
        const double red = ocaml_max(0., threshold - 1.), c_p = pow(red, power);
        if (power < 1.) v = (pow(counts, power) - c_p) * pow(cmax, 1. - power) / power;
        else v = (pow(counts, power) - c_p) / (pow(threshold, power) - c_p);
      }
      return ocaml_max(0., floor(v) / csum);
    }
  }
}

static int cmp_double(const void *a, const void *b) {
  const double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

/* Int32.of_float on x86-64: truncate to the native int, keep the low 32 bits */
static int32_t int32_of_float(double x) {
  if (!(x > -9.2e18 && x < 9.2e18)) return 0;
  return (int32_t)(uint32_t)(uint64_t)(int64_t)x;
}

double kpo_counter_combine(const int32_t *const *columns, uint64_t n_rows, const uint32_t *sel, uint32_t n_sel,
                           const double *col_sum, int criterion, int32_t *out) {
  double max_norm = 0.; /* lib/KMerDB.ml:646-660 */
  for (uint32_t s = 0; s < n_sel; ++s)
    if (col_sum[sel[s]] > max_norm) max_norm = col_sum[sel[s]];
  double *vals = (double *)malloc(sizeof(double) * (n_sel ? n_sel : 1));
  double norm = 0.;
  for (uint64_t i = 0; i < n_rows; ++i) {
    uint32_t m = 0;
    double sum = 0.;
    for (uint32_t s = 0; s < n_sel; ++s) { /* :687-696 */
      const double nrm = col_sum[sel[s]];
      if (nrm > 0.) {
        const double v = (double)columns[sel[s]][i] * max_norm / nrm;
        vals[m++] = v;
        sum += v;
      }
    }
    double res;
    if (criterion == 0) res = sum; /* RescaledMean -> FVF.sum, :703-704 */
    else {                         /* RescaledMedian -> FVF.median * n, :705-706 */
      qsort(vals, m, sizeof(double), cmp_double);
      res = (m ? vals[m / 2] : 0.) * (double)n_sel;
    }
    norm += res;               /* :715 */
    out[i] = int32_of_float(res); /* :716 */
  }
  free(vals);
  return norm;
}

/* ------------------------------------------------------------------ embeddings (lib/Matrix.ml:78-128) */
void kpo_embeddings(const double *m, uint32_t rows, uint32_t n_dims, const double *metric, int kind, double p,
                    int normalize, double *out) {
  const double inv_power = kind == KPO_MINKOWSKI ? 1. / p : 0.5; /* :84-87 */
  double *w = (double *)malloc(sizeof(double) * (n_dims ? n_dims : 1));
  for (uint32_t c = 0; c < n_dims; ++c) w[c] = pow(metric[c], inv_power); /* :88 */
  for (uint32_t i = 0; i < rows; ++i) {
    double *v = out + (uint64_t)i * n_dims;
    for (uint32_t c = 0; c < n_dims; ++c) v[c] = m[(uint64_t)i * n_dims + c] * w[c]; /* :104 */
    if (normalize) {
      const double norm = kpo_norm(kind, p, metric, v, n_dims); /* :106 */
      if (norm != 0.)
        for (uint32_t c = 0; c < n_dims; ++c) v[c] = v[c] / norm;
    }
  }
  free(w);
}

/*
 * kpop_hip.h -- C ABI of libkpop_hip.so: KPop's count -> twist -> distance hot
 * path on AMD MI355X (gfx950).
 *
 * The reference (PaoloRibeca/KPop) is pure OCaml and has no FFI of its own
 * (SURVEY.md F1); these are the entry points an OCaml `ctypes`/stub binding
 * would bind to replace the three OCaml call sites on the hot path.  Each
 * entry point cites the reference code it replaces (file:line under the
 * reference checkout).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C, no exceptions/callbacks across the boundary, no torch types;
 *   - every function returns 0 on success or a negative kpop_status;
 *     kpop_last_error() returns a thread-local message for the last failure;
 *   - host entry points (kpop_*) take caller-owned host buffers and do
 *     H2D / compute / D2H internally;
 *   - device entry points (kpop_dev_*) take pointers that already live in HBM
 *     plus a hipStream_t passed as void* (NULL = default stream); they only
 *     enqueue work and never synchronise;
 *   - call from the PARENT process only (a HIP context does not survive
 *     fork(); the reference's fork()ed workers at lib/Twister.ml:90 are
 *     replaced wholesale, not per-worker).  Host entry points of one device
 *     run one at a time (the library serialises them); device entry points on
 *     different streams may run concurrently (the library's scratch is per
 *     stream);
 *   - kpop_init(device) drives one GPU; kpop_init_devices(devices, n) drives
 *     n of them from ONE process (SURVEY.md 8b "multi-GPU sharding is
 *     internal"): handles (twisters, pipelines) belong to the device slot that
 *     was current when they were made, and kpop_sharded_* spread a batch over
 *     all slots.
 *
 * k-mer encoding (declared by this repository, see kpop_amd/csrc/kmer.h; the
 * reference keeps it in the absent BiOCamLib): A0 C1 G2 T3 either case,
 * big-endian 2-bit packing, DNA-ds key = min(fwd, reverse complement), any
 * other byte breaks the window.  Protein k-mers (KMers.ProteinHash): the 20 standard amino acids in alphabetical
 * order of their one-letter codes are 0..19, 5 bits per residue big-endian, any other byte breaks the window;
 * names are ceil(5k/4) hex digits.  Twisting takes hashes of either kind (they are just column keys).
 */
#ifndef KPOP_HIP_H
#define KPOP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  KPOP_OK = 0,
  KPOP_ERR_INVALID = -1,     /* bad argument */
  KPOP_ERR_CAPACITY = -2,    /* caller buffer too small */
  KPOP_ERR_UNSUPPORTED = -3, /* valid request the HIP path does not cover yet */
  KPOP_ERR_HIP = -4,         /* HIP runtime error (no device, OOM, launch failure) */
  KPOP_ERR_NOT_INIT = -5
} kpop_status;

/* bin/KPopCount.ml:66-82 (Content.t) */
#define KPOP_DNA_DS 0
#define KPOP_DNA_SS 1
#define KPOP_PROTEIN 2 /* k <= 12; counting, and since round 6 the fused count->twist calls too (kpop_count_twist, kpop_dev_count_twist,
                          kpop_spectra_twist, the pipeline: five bits a residue, a twister loaded with 2 k_load >= 5 k bits); not the packed form */
/* lib/Space.ml:140-143 (Distance.t) */
#define KPOP_EUCLIDEAN 0
#define KPOP_COSINE 1
#define KPOP_MINKOWSKI 2
/* lib/Space.ml:81-84 (Metric.t) */
#define KPOP_METRIC_FLAT 0
#define KPOP_METRIC_POWERS 1

/* ---------------------------------------------------------------- runtime */
int kpop_init(int device);          /* select the GPU this process drives (= kpop_init_devices(&device, 1)) */
/* Several GPUs in one process: devices[i] becomes device SLOT i (the same GPU may be named twice: two independent
   slots on it -- how the multi-device code is exercised on a one-GPU box).  Replaces `-T` / Parallel.get_nproc,
   bin/KPopTwistDB.ml:103, and the fork()ed workers of lib/Twister.ml:90-196.  The calling thread works on slot 0
   afterwards; kpop_use_device(slot) moves it (thread-local, like hipSetDevice).  Peer access between distinct GPUs is
   enabled where the hardware allows it (xGMI).                                                                       */
int kpop_init_devices(const int *devices, int n);
int kpop_use_device(int slot);
int kpop_device_slots(void);        /* slots initialised by the last kpop_init / kpop_init_devices */
int kpop_shutdown(void);            /* release workspaces */
int kpop_device_count(void);        /* >=0, or negative kpop_status */
const char *kpop_last_error(void);
const char *kpop_version(void);
int kpop_synchronize(void *stream);
/* performance knobs for A/B measurements (results are identical for every setting):
   "unroll" 8|16 row loads in flight per wave; "nt" row loads 0 plain | 1 non-temporal | 2 chosen by the size of the
   twister (default); "seg" windows per segment of the genome kernel, 0 = sized to an XCD's L2 (default); "hist" 1 (default) | 0: the
   merged spectrum of kpop_count_reads(per_read = 0) by atomic histogram where the hashes fit 26 bits, or always by sort;
   "histlds" 1 (default) | 0 | 2 | 3 | 4: that histogram staged through LDS as the batch suggests (private tables up to k = 7;
   (hash, count) tables over the same stretch of 64 assemblies of one organism; for what does not repeat -- a read set,
   unrelated genomes -- the hashes partitioned by their top bits and every bucket counted in LDS, the table written without
   a global atomic, and -- round 5 -- the (hash, count) pairs of every bucket written straight into the spectrum at an offset taken
   by a decoupled look-back: the 4^k-counter table is never written), direct global atomics as in round 2, chunks always
   combined, always partitioned, or (4) partitioned with the dense table and its compaction as in round 4;
   "histguess" 1 (default) | 0: that partition sizes its buckets from a sample of the batch (one piece in 32; a bucket that
   overflows its room sends the call back to the exact count) or always counts every window first;
   "summary2" 1 (default) | 3 | 0 | 2: summaries against more than 4,096 rows by brackets from a sample and ONE pass over the
   distance rows, the same in two passes, round 2's one block per row, or (131,072 rows and more) with the distances
   computed and reduced in one kernel and no distance rows in memory -- same results, 1, 3 and 2 level;
   "distance_mfma" 1 (default) | 0: kpop_dev_distance_rowwise of 2^32 products and more (rows x rows x dimensions), euclidean / cosine: the
   pairs' dot products as f64 MFMAs (a tiled contraction, any number of dimensions), pairs whose d^2 is a small part of |a|^2 + |b|^2
   recomputed with the reference's chain -- <= 1e-12 relative (3e-15 measured), not bit for bit; 0: the chain for every pair (the reference's bits);
   "summary_mfma" 1 (default) | 2 | 0: summaries against 65,536 rows and more, euclidean / cosine, 4 dimensions and more (up to 128 the query rows
   stay in registers, beyond a tiled contraction; 2: up to 128 dimensions the summary's pass runs inside the contraction and no approximate row
   is written -- same results, measured level to slower: 2.77 against 2.71 ms for 256 x 1M x 64, 9.1 against 8.6 for 1,024): the distances
   as f64 MFMAs (|a|^2 + |b|^2 - 2 a.b) that only LOCATE the neighbours, the median and the MAD's edges; everything reported is
   recomputed with the reference's chain, rows the refinement cannot vouch for are redone from exact distance rows
   (distance_mfma.hip).  Same medians, MADs and neighbour lists bit for bit; mean and standard deviation are sums of the
   approximate values (1e-13 relative; values cancellation would show in are replaced by exact ones).  0: the chain for every pair.
   "summary_mfma_lists" 1 (default) | 0: that refinement reads the candidate lists the summary's one pass left, or scans the rows;
   "summary_lanes" 1 (default) | 2: 512 query rows and more of such a summary in batches of 256 on two streams (measured level);
   "summary_audit" 0 (default) | 1: count the rows left to the exact fall-back (kpop_debug_summary_fallbacks);
   "summary_sample" 1 (default) | 0: such a summary's brackets and bands from the query rows' distances to a sample of the REFERENCE ROWS at even
   spacing (a small contraction of its own; whatever the layout of the database), or from 64 runs of 1,024 consecutive elements of the distance
   rows (a database laid out lineage by lineage makes those runs speak for a few lineages: the brackets miss, the rows take the slow kernel);
   "summary_rawref" 1 (default) | 0: such a summary takes its reference set as it is (no normalised copy of it is made: the norms' pass keeps the
   rows' sums of squares, dot products are scaled where they come out, the exact chains divide as they go -- the same bits), or makes the copy;
   "summary_pass" 1 (default) | 0: the pass over such a summary's approximate rows written for rows the library made itself, or the general one;
   "tilepipe" 1 (default) | 0: the tile route's kernel with producer and consumer wavefronts (tile_pipe.h; beyond 64 dimensions its three-stage
   form: producers / MFMA wavefronts / gather wavefronts), or round 4's; "tilewide" 0 (default) | 1: that three-stage form at any number of
   dimensions (up to 64: the same bits); "tilecap_mb" 0 (default: 4 GiB per 64 columns, a quarter of the device at most) | MiB: the per-slot tables
   a call of that route may take before the batch goes through in sub-batches of sequences; "pipeprio" 1 (default) | 0..3: issue priority of its MFMA wavefronts;
   "dense" 2 (default) | 1 | 0: the matrix-core routes of the twist.  2: chosen by the batch -- sequences of more than 512
   windows (assemblies, k <= 15) go through count_twist_tile_kernel: the CONSENSUS rows of a stretch of 64 sequences x 512
   windows (the rows of four seed sequences, an LDS set) are multiplied on the f64 matrix cores, the rows private to one
   sequence are gathered per sequence (tile_residual_kernel), stretches that share little with their seeds stay with the
   streaming kernel; kpop_count_twist's batches of assemblies at small k go through the dense image of their
   counts (kpop_dev_count_twist_dense); kpop_twist takes the dense contraction when the spectra are dense enough.  1: kpop_twist
   always by the dense contraction.  0: opt out -- the sparse mat-vec in the reference's order of additions everywhere.  The
   one knob that changes results, in the last bits (<= 2e-15 relative measured; north_star allows 1e-5); "ldspad" bytes of
   extra LDS per block of the fused reads kernel (an occupancy probe); "dbg" development switches (also KPOP_TUNE_DBG)  */
int kpop_tune(const char *key, int value);
/* development: the phase clocks count_twist_tile_kernel adds up under kpop_tune("dbg", 16 << 24) (s_memtime ticks of thread 0 of
   every block, eight phases in out[0..7]; tools/probes/ab_tile_kernel.py prints them) and, under kpop_tune("dbg", 32 << 24), the
   chunks it took (out[14]) and the rows of their consensus sets as multiplied on the matrix cores (out[15]: 2 x 64 x n_dims
   flops each; bench.py's MFMA roofline); read and cleared; synchronises the device */
int kpop_debug_counters(uint64_t *out, int n);
/* development: under kpop_tune("summary_audit", 1) the large-reference summaries (kpop_dev_distance_summary against 65,536 rows and
   more) count the query rows they leave to a fall-back -- rows whose sample-based brackets missed (the ten-pass kernel) or whose
   certificates failed (exact distance rows); the
   results are the same either way, the fall-back costs milliseconds.  Read and cleared.                                         */
int kpop_debug_summary_fallbacks(uint64_t *rows);

/* plain device-memory helpers so a non-HIP host language can stage buffers */
int kpop_dev_malloc(void **ptr, uint64_t bytes);
int kpop_dev_free(void *ptr);
int kpop_memcpy_h2d(void *dst, const void *src, uint64_t bytes);
int kpop_memcpy_d2h(void *dst, const void *src, uint64_t bytes);
int kpop_dev_memset(void *dst, int value, uint64_t bytes);
/* Page-locked host memory.  The streaming pipeline below copies straight from / to the caller's buffers; when they are
   page-locked the copy engines run beside the kernels and kpop_pipeline_submit returns without waiting.  An OCaml
   binding wraps kpop_host_alloc'ed memory as a Bigarray (Ctypes.bigarray_of_ptr) once and reuses it for every batch;
   kpop_host_register pins memory the caller already owns (costly per call: do it once per buffer, not per batch). */
int kpop_host_alloc(void **ptr, uint64_t bytes);
int kpop_host_free(void *ptr);
int kpop_host_register(void *ptr, uint64_t bytes);
int kpop_host_unregister(void *ptr);

/* ------------------------------------------------------------------ count
 * Replaces the per-read loop of KMerCounter.compute, bin/KPopCount.ml:36-50:
 *   KIH.iterc res read.seq (:38) + KIHF.iter dump (:46,:60) + KIHF.clear (:49).
 * bases: concatenated, already linted sequences (read r = bases[offsets[r] ..
 * offsets[r+1])).  per_read=1 is -L (one spectrum per read, :39-50); per_read=0
 * is -l (one spectrum for all reads, :60).  Output: CSR of unique
 * (hash, count), hashes ascending inside a spectrum (the reference's Hashtbl
 * order is unspecified; consumers key by name).  out_offsets has n_reads+1
 * entries (per_read=1) or 2 (per_read=0).  k: 1..30.  Integer, bit-exact.    */
int kpop_count_reads(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content,
                     int per_read, uint64_t *out_hash, uint32_t *out_count, uint64_t *out_offsets,
                     uint64_t out_capacity);

/* ---------------------------------------------------------------- twister
 * Device-resident twister: replaces Twister.t.twister (lib/Twister.ml:22-25)
 * plus the inverted name->column Hashtbl (lib/Twister.ml:71-76).
 * T_dims_major is the reference layout: n_dims rows (one Float.Array per
 * dimension) of n_cols coefficients; col_hash[c] = hash of the k-mer naming
 * column c (hex name parsed by the binding).  On device it is re-laid k-mer-
 * major so one k-mer's coefficients are one contiguous row.                  */
typedef struct kpop_twister kpop_twister;
int kpop_twister_load(const double *T_dims_major, uint64_t n_cols, uint32_t n_dims, const uint64_t *col_hash,
                      int k, kpop_twister **out);
/* synthetic twister generated on device: columns = every
   canonical (DNA-ds) / every (DNA-ss) k-mer ascending, coefficient(d,h) from
   SplitMix64 -- bench/test tooling, same function as the oracle's.           */
int kpop_twister_synth(uint64_t seed, int k, int content, uint32_t n_dims, kpop_twister **out);
/* One rank's slice of the same twister when its k-mer rows are sharded over GPUs (k = 15 / large D):
   only the k-mers with hash in [hash_lo, hash_hi).  acc_dim = 1 appends a dimension of ones, so an un-normalised
   twist also carries the shard's part of the normaliser `acc` (lib/Twister.ml:158): the ranks all-reduce
   [n x (n_dims+1)] partials and divide by the last column (kpop_amd/shard.py).                                  */
int kpop_twister_synth_slice(uint64_t seed, int k, int content, uint32_t n_dims, uint64_t hash_lo,
                             uint64_t hash_hi, int acc_dim, kpop_twister **out);
/* The fused count -> twist entry points hash the reads with the k the twister was loaded with.  A twister read from
   a file only shows the WIDTH of its k-mer names, ceil(k/2) hex digits (bin/KPopCount.ml:46), so a loader that
   had to infer k loads with the even candidate and, once the producer of the reads has said which k it counts
   with, sets it here (k <= the loaded k: the name -> row index of the larger k also holds every smaller hash). */
int kpop_twister_set_count_k(kpop_twister *tw, int k);
int kpop_twister_free(kpop_twister *tw);
int kpop_twister_info(const kpop_twister *tw, uint64_t *n_cols, uint32_t *n_dims, int *k,
                      uint64_t *device_bytes);
/* Bytes (part of device_bytes) of the twister's second copy of its rows AT THEIR HASHES, 0 when it keeps none.  A nearly
   complete twister of k = 13..15 and up to 32 dimensions keeps one when it fits the free memory with room to spare
   (k = 15, 16 dimensions: 137 GB beside 69 GB): the fused count -> twist of reads then needs no name -> row look-up, which
   for such a twister is a cache line of index for every cache line of row (lib/Twister.ml:151 is a Hashtbl.find_opt per
   k-mer).  kpop_tune("direct", 0 | 1 | 2) before loading: never | whenever it fits | by that rule (default).  Same results. */
int kpop_twister_direct_bytes(const kpop_twister *tw, uint64_t *bytes);

/* ------------------------------------------------------------------ twist
 * Replaces the worker closure of Twister.add_twisted_from_files,
 * lib/Twister.ml:146-188: name->column lookup (:151), acc over found k-mers
 * only (:158), duplicate k-mers summed (:160-163), normalise by acc when
 * normalize && acc<>0 (:177-178), sparse mat-vec (:183).
 * Spectra are CSR (hash, value) lines; out is n_spectra x n_dims row-major
 * (the "transposed" product of :52,:190).                                    */
int kpop_twist(const kpop_twister *tw, const uint64_t *hash, const double *value, const uint64_t *offsets,
               uint32_t n_spectra, int normalize, double *out);

/* Fused count -> twist (bin/KPopCount.ml:36-50 piped into lib/Twister.ml:146-188,
 * the README.md:606 pipeline) without materialising text spectra.            */
int kpop_count_twist(const kpop_twister *tw, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                     int content, int normalize, double *out);

/* ------------------------------------------------------------ packed bases
 * BASELINE north_star's "packed bases" at the boundary: 16 bases a 32-bit word of 2-bit codes (A 0, C 1, G 2, T 3, either case; base i
 * of the batch in bits 2 (i % 16).. of word i / 16) and one bit a base that is none of ACGTacgt (32 bases a word; such a base starts
 * no k-mer, as after Sequences.Lint.dnaize, bin/KPopCount.ml:242-245): 2.25 bits a base where every other entry point takes 8.  A host
 * that keeps its sequences packed sends 3.56 x fewer bytes over the bus -- BASELINE config 3's 1.5 GB of bases cost 26 ms of bus
 * against 9.6 ms of kernels.  kpop_pack_bases packs on the host (threads <= 0: the library chooses); on the device the words are
 * spread back to one byte a base at HBM's rate and the kernels run as they are: the ASCII entry points' results bit for bit.
 * offsets are in bases, from base 0 of the packed arrays.  Replaces the sequence side of bin/KPopCount.ml:36-50.                  */
uint64_t kpop_packed_code_words(uint64_t n_bases);   /* (n_bases + 15) / 16 */
uint64_t kpop_packed_mask_words(uint64_t n_bases);   /* (n_bases + 31) / 32 */
int kpop_pack_bases(const uint8_t *bases, uint64_t n_bases, uint32_t *codes, uint32_t *invalid, int threads);
int kpop_count_twist_packed(const kpop_twister *tw, const uint32_t *codes, const uint32_t *invalid, const uint64_t *offsets, uint32_t n_reads,
                            int content, int normalize, double *out);

/* The same pipeline with the count spelled out: the rows that kpop_count_reads(k, content, per_read = 1) followed
 * by kpop_twist would give, bit for bit, for sequences of any length, with the spectra never leaving the device.
 * k is the caller's (k <= the k the twister was loaded with: a twister file shows only the width of its k-mer names).
 * This is what KPopTwistDB runs when KPopCount hands it reads instead of text spectra (kpop_amd/host/fast_seq.h). */
int kpop_spectra_twist(const kpop_twister *tw, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                       int k, int content, int normalize, double *out);

/* ------------------------------------------------------- streaming pipeline
 * Reads in host memory -> twisted rows / distances to a set of class vectors / per-read summary in host memory: the
 * README.md:606 + :641 / :656 chain (bin/KPopCount.ml:36-50 -> lib/Twister.ml:58-206 -> lib/Matrix.ml:191-266 or
 * :691-766) as one call, with the upload of chunk c+1, the kernels of chunk c and the download of chunk c-1 running at
 * the same time on three streams (SURVEY.md 7 "pinned-memory double-buffered ingest ... never materialise").
 * `outputs` selects what crosses the bus back: a caller after -d or -s never pulls the twisted rows.  Every output
 * row depends on its own read only; results are bit-identical to kpop_dev_count_twist + kpop_dev_distance_rowwise /
 * kpop_dev_distance_summary on the whole batch.
 * A pipeline belongs to the device slot that was current at creation; twister, classes (n_classes x n_dims row-major)
 * and metric are those of kpop_count_twist / kpop_distance_rowwise (first operand = classes, as `-d` has the register
 * on the column side, lib/Matrix.ml:253).                                                                            */
#define KPOP_OUT_TWISTED 1
#define KPOP_OUT_DISTANCES 2
#define KPOP_OUT_SUMMARY 4
typedef struct kpop_pipeline kpop_pipeline;
typedef struct {
  uint32_t struct_size;      /* = sizeof(kpop_pipeline_config) */
  int content;               /* KPOP_DNA_DS | KPOP_DNA_SS */
  int normalize_counts;      /* --counts-normalize, lib/Twister.ml:177 */
  int kind;                  /* KPOP_EUCLIDEAN | KPOP_COSINE | KPOP_MINKOWSKI */
  double p;                  /* Minkowski power */
  int normalize_distances;   /* --distance-normalize, lib/Matrix.ml:197-202 */
  int outputs;               /* KPOP_OUT_* ored */
  uint32_t keep_at_most;     /* summary: --summary-keep-at-most, 0 = all */
  uint32_t max_neighbours;   /* summary: stride of the neighbour outputs */
  uint32_t chunk_reads;      /* reads per chunk, 0 = chosen from the batch (a quarter of it, 16,384..131,072, after a short first chunk) */
  uint32_t depth;            /* chunks in flight (device slots), 0 = 4 */
  uint64_t chunk_bases;      /* bases per chunk, 0 = 256 MiB (a longer sequence gets a chunk of its own) */
  int record_timeline;       /* 1: timing events around every chunk's upload, kernels and download (kpop_pipeline_timeline) */
} kpop_pipeline_config;
typedef struct {
  double *twisted;           /* n_reads x n_dims            (KPOP_OUT_TWISTED)   */
  double *distances;         /* n_reads x n_classes         (KPOP_OUT_DISTANCES) */
  double *stats;             /* n_reads x 4: mean, sd, median, MAD (KPOP_OUT_SUMMARY, as kpop_distance_summary) */
  uint32_t *n_neighbours;    /* n_reads */
  uint32_t *nb_index;        /* n_reads x max_neighbours */
  double *nb_distance;       /* n_reads x max_neighbours */
  double *nb_z;              /* n_reads x max_neighbours */
} kpop_pipeline_outputs;
int kpop_pipeline_create(const kpop_twister *tw, const double *classes, uint32_t n_classes, const double *metric,
                         const kpop_pipeline_config *cfg, kpop_pipeline **out);
/* Enqueue one batch.  bases / offsets / the output buffers must stay valid and untouched until the ticket is
   collected.  With page-locked buffers the call only enqueues (several batches may be in flight: submit the next
   before collecting the previous and the bus never idles); with pageable ones it may wait for copies.            */
int kpop_pipeline_submit(kpop_pipeline *pl, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                         const kpop_pipeline_outputs *out, uint64_t *ticket);
int kpop_pipeline_submit_packed(kpop_pipeline *pl, const uint32_t *codes, const uint32_t *invalid, const uint64_t *offsets, uint32_t n_reads,
                                const kpop_pipeline_outputs *out, uint64_t *ticket);  /* the batch in 2.25 bits a base (see "packed bases") */
int kpop_pipeline_collect(kpop_pipeline *pl, uint64_t ticket);  /* returns once that batch's outputs are in host memory */
int kpop_pipeline_run(kpop_pipeline *pl, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                      const kpop_pipeline_outputs *out);        /* submit + collect */
/* of the last submit: chunks it was cut into, whether every buffer was page-locked, slots in the ring */
int kpop_pipeline_stats(const kpop_pipeline *pl, uint32_t *chunks, int *pinned, uint32_t *depth);
/* The device-side timeline of the last submit of a pipeline created with record_timeline (call after its collect): per
   chunk six times in milliseconds from the start of the first upload -- upload start, end; kernels start, end; download
   start, end -- taken with events on the three streams.  ms holds max_chunks x 6 doubles; *n_chunks = chunks written. */
int kpop_pipeline_timeline(kpop_pipeline *pl, uint32_t max_chunks, double *ms, uint32_t *n_chunks);
int kpop_pipeline_destroy(kpop_pipeline *pl);

/* ------------------------------------------------------------ several GPUs
 * "multi-GPU sharding is internal".  After kpop_init_devices(devices, n) these calls spread one batch
 * over all n device slots from inside the library -- one host thread per device, started per call -- replacing `-T`
 * (bin/KPopTwistDB.ml:103) and the fork()ed workers of lib/Twister.ml:90-196 and lib/Matrix.ml:212-266,712-766.
 * Every sequence is independent through count and twist: reads are cut into contiguous shards
 * (kpop_shard_bounds), twister / classes / metric are replicated, results land in the caller's rows.                 */
/* rows [lo, hi) of n_items that `rank` of `world` owns: balanced, the first n_items % world ranks get one more.
   Pure host arithmetic (no GPU).                                                                                    */
int kpop_shard_bounds(uint64_t n_items, int rank, int world, uint64_t *lo, uint64_t *hi);
/* a twister on another device slot: copied device to device (xGMI where peers are enabled); onto a slot of the SAME
   GPU it is a second handle on the same arrays.  Free with kpop_twister_free.                                       */
int kpop_twister_replicate(const kpop_twister *src, int slot, kpop_twister **out);
typedef struct kpop_sharded kpop_sharded;
/* One streaming pipeline (above) per device slot; `tw` may live on any slot and is replicated to the others.
   classes / metric may be NULL for a twisted-rows-only job (metric alone is enough for the all-vs-all summary).     */
int kpop_sharded_create(const kpop_twister *tw, const double *classes, uint32_t n_classes, const double *metric,
                        const kpop_pipeline_config *cfg, kpop_sharded **out);
int kpop_sharded_slots(const kpop_sharded *sh);
/* kpop_pipeline_run over all devices: host memory to host memory, rows of shard s written by device s.  Distances
   against a reference set need no exchange between devices.  Bit-identical to one pipeline on one device.           */
int kpop_sharded_run(kpop_sharded *sh, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                     const kpop_pipeline_outputs *out);
/* kpop_spectra_twist with the reads cut over the devices (any sequence length; bit for bit the rows of kpop_count_reads
   + kpop_twist)                                                                                                      */
int kpop_sharded_spectra_twist(kpop_sharded *sh, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k,
                               int content, int normalize, double *out);
/* BASELINE config 4 with the reads already resident: slot s holds n_reads[s] reads in ITS HBM (d_bases[s],
   d_offsets[s] = n_reads[s] + 1 offsets into d_bases[s]; n_bases[s] bytes; max_len = longest read anywhere).  Every
   device twists its shard in `chunks` pieces and -- gather != 0 -- pushes each finished piece into every peer's copy
   of the full n_total x n_dims matrix (hipMemcpyPeerAsync on one stream per destination: the all-gather of
   SURVEY.md 8e over point-to-point xGMI links) while it twists the next; then the distances of its rows to the
   classes.  Returns when all devices are done and every copy of the matrix is complete.                             */
int kpop_sharded_resident_step(kpop_sharded *sh, const uint8_t *const *d_bases, const uint64_t *const *d_offsets,
                               const uint32_t *n_reads, const uint64_t *n_bases, uint32_t max_len, int chunks, int gather);
/* device pointers of slot `slot` after a resident step: the full matrix (read order), the slot's first row and row
   count in it, its distances to the classes (n_rows x n_classes)                                                    */
int kpop_sharded_resident_buffers(const kpop_sharded *sh, int slot, double **d_full, uint64_t *first_row, uint64_t *n_rows,
                                  double **d_distances);
/* of the last resident step on that slot: host time until its kernels were done, and the time it then still waited
   for its pushes to land (the exposed part of the exchange)                                                         */
int kpop_sharded_timings(const kpop_sharded *sh, int slot, double *ms_compute, double *ms_exposed_comm);
/* ... and its fused count->twist launches chunk by chunk (HIP events on the slot's compute stream; what a roofline of the
   in-process path is computed from): ms[0 .. *n_chunks), at most max_chunks                                          */
int kpop_sharded_chunk_timings(const kpop_sharded *sh, int slot, double *ms, int max_chunks, int *n_chunks);
/* After a resident step with gather: slot s summarises the first min(queries_per_slot, n_s) rows of its shard (0 =
   all) against ALL n_total twisted vectors (Matrix.summarize_rowwise, lib/Matrix.ml:691-766; N x N is never formed).
   One output row per query, slots in order; out_query = the query's global read number; neighbour indices are global
   read numbers.  capacity = rows the output arrays hold.                                                            */
int kpop_sharded_all_vs_all_summary(kpop_sharded *sh, uint32_t queries_per_slot, uint32_t keep_at_most, uint32_t max_neighbours,
                                    uint64_t capacity, uint64_t *n_queries_out, uint64_t *out_query, double *out_stats,
                                    uint32_t *out_n, uint32_t *out_idx, double *out_dist, double *out_z);
int kpop_sharded_destroy(kpop_sharded *sh);
/* kpop_distance_rowwise / kpop_distance_summary with the rows of the second operand cut over the device slots        */
int kpop_sharded_distance_rowwise(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                                  const double *metric, int kind, double p, int normalize, double *out);
int kpop_sharded_distance_summary(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                                  const double *metric, int kind, double p, int normalize, uint32_t keep_at_most,
                                  uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx,
                                  double *out_dist, double *out_z);

/* ------------------------------------------------------ twister generation
 * Replaces the R stage of src/KPopTwist:93-116 (library `ca`): correspondence analysis of a k-mers x spectra
 * count table.  counts is n_kmers x n_spectra row-major (what `KPopCountDB -t` exports, src/KPopTwist:38-44);
 * normalize = divide every column by its sum first (:93-94).  n_dims = min(n_kmers, n_spectra) - 1.  Outputs:
 * twisted (n_spectra x n_dims: the class positions, :98-100), inertia (n_dims, :105) and the twister
 * (n_dims x n_kmers, dims-major: exactly what kpop_twister_load takes, :110-116).  Dimension signs are
 * arbitrary, as they are in R.  The dense S'S and S*W contractions run on the f64 matrix cores.            */
int kpop_ca(const double *counts, uint64_t n_kmers, uint32_t n_spectra, int normalize, uint32_t *n_dims_out,
            double *twisted, double *inertia, double *twister);

/* ----------------------------------------------------------------- metric
 * Replaces Space.Distance.Metric.compute, lib/Space.ml:88-105 (called from
 * Twister.get_metrics_vector, lib/Twister.ml:208-209).  O(n_dims) host code. */
int kpop_metric_compute(int metric_kind, const double *inertia, uint32_t n_dims, double power_int,
                        double threshold, double power_ext, double *out);

/* --------------------------------------------------------------- distance
 * Replaces Base.get_normalizations + Base.get_distance_rowwise,
 * lib/Matrix.ml:42-76,191-266 (Space.Distance.compute, lib/Space.ml:182-205).
 * out is r2 x r1 row-major: out[j*r1+i] = d(m1 row i, m2 row j) (:253).      */
int kpop_distance_rowwise(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                          const double *metric, int kind, double p, int normalize, double *out);

/* Replaces Matrix.summarize_rowwise + summarize_distance_matrix_row,
 * lib/Matrix.ml:691-766,632-690: per m2 row the mean, sample sd, upper median
 * and MAD of its r1 distances (out_stats r2 x 4), then the keep_at_most
 * closest m1 rows, whole tie groups included (:648-649).  Neighbour outputs
 * have a fixed stride max_neighbours per row; out_n[j] is the reference's
 * eff_len (may exceed max_neighbours: entries beyond the stride are dropped).
 * keep_at_most=0 means "all" (:723-726).  The r2 x r1 matrix is never formed.
 * Lists of any length: against a first operand of more than 4,096 rows the summary kernels themselves return at most
 * 2,048 neighbours per row; a longer list (keep_at_most = all, :723-726; a tie group of thousands, :648-649) is completed
 * by the host entry points -- the row's distances sorted by (distance, column) with the device-wide radix sort.  The
 * device entry points (kpop_dev_*) report out_n and fill at most 2,048 entries of such a row.
 * Cost of a long row: its distances recomputed, an 8-pass radix sort of r1 keys, two copies and a stream synchronisation --
 * about 30 launches and ~0.3 ms a row (r1 = 100,000), one row after the other under the slot's lock.  keep_at_most = 0 against
 * more than 4,096 rows makes EVERY query row long: meant for the reference's use (a few rows inspected in full), not for
 * 100,000 queries (tens of seconds). */
int kpop_distance_summary(const double *m1, uint32_t r1, const double *m2, uint32_t r2, uint32_t n_dims,
                          const double *metric, int kind, double p, int normalize, uint32_t keep_at_most,
                          uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx,
                          double *out_dist, double *out_z);

/* Replaces Base.get_embeddings, lib/Matrix.ml:78-128 (KPopTwistDB -e): every row times metric ** (1/2, or 1/p for
 * Minkowski), then -- if normalize -- divided by its own norm under the same distance and metric (left as is when
 * that norm is 0).  out is rows x n_dims.                                                                         */
int kpop_embeddings(const double *m, uint32_t rows, uint32_t n_dims, const double *metric, int kind, double p,
                    int normalize, double *out);

/* Replaces Matrix.get_splits with SplitsAlgorithm.Gaps, lib/Matrix.ml:524-600 (KPopTwistDB -p, the default algorithm):
 * per dimension the rows sorted by coordinate and the gaps between consecutive ones; all gaps by decreasing size, then
 * dimension, then position; the s-th (s < *n_splits <= max_splits) is the split { perm[out_dim[s]][0 .. out_idx[s]] }
 * with weight out_gap[s].  perm is n_dims x rows (the row order of every dimension); rows with equal coordinates stay in
 * ascending row order.  The container the reference puts splits in (BiOCamLib Trees.Splits) is declared host-side.     */
int kpop_splits_gaps(const double *embeddings, uint32_t rows, uint32_t n_dims, uint32_t max_splits, uint32_t *n_splits,
                     double *out_gap, uint32_t *out_dim, uint32_t *out_idx, uint32_t *perm);

/* Replaces Matrix.summarize_distance, lib/Matrix.ml:767-810 (KPopTwistDB -S): the same per-row summary
 * over a distance matrix that already exists (r2 rows x r1 columns, row-major).                      */
int kpop_summarize_distances(const double *dist, uint32_t r2, uint32_t r1, uint32_t keep_at_most,
                             uint32_t max_neighbours, double *out_stats, uint32_t *out_n, uint32_t *out_idx,
                             double *out_dist, double *out_z);

/* --------------------------------------------------- device-resident path
 * Same operations on buffers already in HBM; enqueue only.                   */
int kpop_dev_synth_reads(uint64_t seed, uint64_t n_reads, uint32_t read_len, uint64_t first_read,
                         uint8_t *d_bases, uint64_t *d_offsets, void *stream);
/* -L counting with everything resident (bin/KPopCount.ml:36-50), reads of up to 512 windows.  The CSR lands
   in d_out_hash / d_out_count (the caller sizes them for the worst case, one entry per window) and
   d_out_offsets (n_reads+1); d_scratch needs kpop_dev_count_reads_scratch_bytes().                        */
uint64_t kpop_dev_count_reads_scratch_bytes(uint32_t n_reads, uint32_t max_len, int k);
int kpop_dev_count_reads(const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads, uint32_t max_len,
                         int k, int content, void *d_scratch, uint64_t *d_out_hash, uint32_t *d_out_count,
                         uint64_t *d_out_offsets, void *stream);
/* The library-owned workspace (segment partials here; distance rows of a chunk in kpop_dev_distance_summary
   against a large first operand) is one PER STREAM (per device slot), so calls on different streams may overlap;
   the first call on a stream that needs more of it than any before synchronises the device to grow it.
   kpop_dev_workspace_reserve[_stream](bytes) grows it ahead of time, after which those calls only enqueue.
   n_bases = offsets[n_reads] (size of d_bases), max_len = longest read in the batch: the host
   knows both from the offsets it uploaded (a read longer than max_len says comes back as a row of NaNs).  Reads of up to 512 windows take the one-wavefront-
   per-read kernel; longer sequences (genomes) the streaming kernel, whose segment partials
   live in a library-owned workspace (grown with hipMalloc on the first call that needs more). */
int kpop_dev_workspace_reserve(uint64_t bytes);                       /* the null stream's */
int kpop_dev_workspace_reserve_stream(uint64_t bytes, void *stream);
int kpop_dev_count_twist(const kpop_twister *tw, const uint8_t *d_bases, const uint64_t *d_offsets,
                         uint32_t n_reads, uint64_t n_bases, uint32_t max_len, int content, int normalize,
                         double *d_out, void *stream);
/* ... from the packed form (see "packed bases"): d_codes / d_invalid as kpop_pack_bases writes them, resident on the device; the bases are
   spread back to bytes into a second library-owned block of the stream (n_bases bytes), then kpop_dev_count_twist.  kpop_dev_unpack_bases
   is that first step by itself (d_bases: n_bases bytes; A C G T, N for a base marked invalid).                                        */
int kpop_dev_unpack_bases(const uint32_t *d_codes, const uint32_t *d_invalid, uint64_t n_bases, uint8_t *d_bases, void *stream);
int kpop_dev_count_twist_packed(const kpop_twister *tw, const uint32_t *d_codes, const uint32_t *d_invalid, const uint64_t *d_offsets,
                                uint32_t n_reads, uint64_t n_bases, uint32_t max_len, int content, int normalize, double *d_out, void *stream);
/* max_lines = lines of the longest spectrum of the batch, or 0 when the caller does not know: up to 512 the kernel keeps
   the columns it finds while summing the counts (lib/Twister.ml:158) for the products (:183) instead of looking them up
   twice.  A spectrum longer than a non-zero max_lines says comes back as a row of NaNs.                              */
int kpop_dev_twist(const kpop_twister *tw, const uint64_t *d_hash, const double *d_value,
                   const uint64_t *d_offsets, uint32_t n_spectra, uint64_t max_lines, int normalize,
                   double *d_out, void *stream);
/* The same twist as a dense contraction on the f64 matrix cores (v_mfma_f64_16x16x4_f64): the spectra of a batch tile
   are laid out as X[tile x n_kmers] and multiplied by the twister's rows.  2 n_kmers D flops per spectrum whatever it
   holds: worth it for many dense spectra of a small k only.  Equal to kpop_dev_twist up to rounding
   (the order of additions is the GEMM's).  d_work needs kpop_dev_twist_dense_workspace_bytes().                    */
uint64_t kpop_dev_twist_dense_workspace_bytes(const kpop_twister *tw, uint32_t n_spectra);
int kpop_dev_twist_dense(const kpop_twister *tw, const uint64_t *d_hash, const double *d_value,
                         const uint64_t *d_offsets, uint32_t n_spectra, int normalize, void *d_work, double *d_out,
                         void *stream);
/* The same for spectra whose lines ASCEND BY HASH (the order kpop_count_reads / kpop_dev_count_reads produce; repeated
   and unknown k-mers allowed): the spectra are densified inside the contraction -- 64 spectra x 128 twister rows at a
   time in LDS, v_mfma_f64_16x16x4_f64 against the twister's rows from L2 -- with no dense image in HBM.  acc
   (lib/Twister.ml:158) is applied once to the finished sums.  A spectrum whose lines do not ascend comes back as a row of
   NaNs.  Same workspace.                                                                                            */
int kpop_dev_twist_dense_sorted(const kpop_twister *tw, const uint64_t *d_hash, const double *d_value,
                                const uint64_t *d_offsets, uint32_t n_spectra, int normalize, void *d_work,
                                double *d_out, void *stream);
/* Sequences -> twisted rows with the counts handed to the contraction DENSE: for a twister of at most 36,864 k-mers
   (every canonical k-mer up to k = 8) and 256 dimensions, one block per sequence counts into an LDS table with one
   counter per twister row and writes it out as a row of u32; the contraction loads those rows (4 bytes per sequence and
   k-mer), divides by acc as it stages them (lib/Twister.ml:177-178, element by element) and multiplies on the f64 matrix
   cores.  What kpop_dev_count_twist gives, to rounding (the order of additions is the GEMM's), for batches of ASSEMBLIES
   at small k, where every spectrum holds most of the columns.  Any sequence length.                                  */
uint64_t kpop_dev_count_twist_dense_workspace_bytes(const kpop_twister *tw, uint32_t n_reads);
int kpop_dev_count_twist_dense(const kpop_twister *tw, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n_reads,
                               int content, int normalize, void *d_work, double *d_out, void *stream);
/* kpop_ca on device-resident data: d_counts (n_kmers x n_spectra, row-major, not modified) in, d_twisted (n_spectra x
   n_dims), d_inertia (n_dims) and d_twister (n_dims x n_kmers, dims-major) out, all device pointers; *n_dims_out is a
   host word.  d_work needs kpop_dev_ca_workspace_bytes() (the standardised copy of the table).  The weights of the
   columns and the order of the eigenvalues are host work between launches: the call synchronises `stream` on the way
   and has completed when it returns.                                                                                */
uint64_t kpop_dev_ca_workspace_bytes(uint64_t n_kmers, uint32_t n_spectra);
/* (d_work == d_counts is allowed: the table is then standardised where it stands and is lost) */
int kpop_dev_ca(const double *d_counts, uint64_t n_kmers, uint32_t n_spectra, int normalize, void *d_work,
                uint32_t *n_dims_out, double *d_twisted, double *d_inertia, double *d_twister, void *stream);
/* What KPopTwist does to the table between the transformation and the analysis (src/KPopTwist:76-91: a keep list, a
   sample, a threshold on the row sums), on a row-major k-mers x spectra table that stays on the device: sums of the
   rows (lane-strided partial sums and a fixed tree per row) and of the columns (slabs of rows added in order), and the
   rows d_rows[0 .. n_sel) copied, in that order, into d_out (n_sel x n_cols).                                          */
int kpop_dev_table_row_sums(const double *d_table, uint64_t n_rows, uint32_t n_cols, double *d_out, void *stream);
int kpop_dev_table_col_sums(const double *d_table, uint64_t n_rows, uint32_t n_cols, double *d_out, void *stream);
int kpop_dev_table_gather_rows(const double *d_table, uint32_t n_cols, const uint64_t *d_rows, uint64_t n_sel,
                               double *d_out, void *stream);
/* Distance workspace: row norms and the pre-normalised copies a/n_i, b/n_j of
   both operands (the per-element divisions of lib/Matrix.ml:247-249, done once).
   kpop_dev_distance_workspace_bytes gives the size d_work must have.          */
uint64_t kpop_dev_distance_workspace_bytes(uint32_t r1, uint32_t r2, uint32_t n_dims);
int kpop_dev_distance_rowwise(const double *d_m1, uint32_t r1, const double *d_m2, uint32_t r2,
                              uint32_t n_dims, const double *d_metric, int kind, double p, int normalize,
                              void *d_work, double *d_out, void *stream);
/* Base.get_normalizations, lib/Matrix.ml:42-76: the norms of the rows of one operand under the distance and metric
   (0 -> 1, :67).  A caller that measures many batches against ONE first operand (the class vectors) computes its norms
   once and hands them to kpop_dev_distance_rowwise_norms with every batch (d_norms1 = NULL: computed inside, = plain
   kpop_dev_distance_rowwise; they are used where the kernel divides while staging -- fewer than 128 rows in the first
   operand, normalize set -- and recomputed otherwise).                                                              */
int kpop_dev_row_norms(const double *d_m, uint32_t rows, uint32_t n_dims, const double *d_metric, int kind, double p,
                       double *d_norms, void *stream);
int kpop_dev_distance_rowwise_norms(const double *d_m1, uint32_t r1, const double *d_norms1, const double *d_m2, uint32_t r2,
                                    uint32_t n_dims, const double *d_metric, int kind, double p, int normalize,
                                    void *d_work, double *d_out, void *stream);
int kpop_dev_distance_summary(const double *d_m1, uint32_t r1, const double *d_m2, uint32_t r2,
                              uint32_t n_dims, const double *d_metric, int kind, double p, int normalize,
                              uint32_t keep_at_most, uint32_t max_neighbours, void *d_work,
                              double *d_out_stats, uint32_t *d_out_n, uint32_t *d_out_idx,
                              double *d_out_dist, double *d_out_z, void *stream);

int kpop_dev_summarize_distances(const double *d_dist, uint32_t r2, uint32_t r1, uint32_t keep_at_most,
                                 uint32_t max_neighbours, double *d_out_stats, uint32_t *d_out_n,
                                 uint32_t *d_out_idx, double *d_out_dist, double *d_out_z, void *stream);

/* ------------------------------------------------- k-mer database (KPopCountDB)
 * the operations of lib/KMerDB.ml that touch every count.  A database is the reference's
 * `storage: I32BAVector.t array` (lib/KMerDB.ml:54-63): n_cols spectra ("columns"), each a vector of n_rows int32
 * counts -- passed as one pointer per spectrum, exactly the shape an OCaml binding has in hand (Bigarray data
 * lives outside the OCaml heap and does not move).  Statistics are {non_zero, max, sum, sum_log} per vector
 * (Transformation.statistics_t, :73-79; `min` is always 0 there and unused).                                   */
#define KPOP_TRANSF_BINARY 0 /* lib/KMerDB.ml:88-92 (Transformation.t) */
#define KPOP_TRANSF_POWER 1
#define KPOP_TRANSF_CLR 2
#define KPOP_TRANSF_PSEUDO 3
#define KPOP_COMBINE_MEAN 0 /* lib/KMerDB.ml:616-626 (CombinationCriterion.t) */
#define KPOP_COMBINE_MEDIAN 1

/* Replaces stats_table_of_core_db, lib/KMerDB.ml:171-271: col_stats is n_cols x 4, row_stats n_rows x 4 (either
 * may be NULL).  A threshold below 1 is relative to the vector's own sum of count^power (:190-195).             */
int kpop_counter_stats(const int32_t *const *columns, uint32_t n_cols, uint64_t n_rows, double threshold,
                       double power, double *col_stats, double *row_stats);

/* Replaces the worker + consumer of add_combined_selected, lib/KMerDB.ml:661-722: combines the spectra
 * columns[sel[0..n_sel)] (visited in that order; the reference's order is its `found_cols`, :646-660) into
 * out[n_rows] = Int32.of_float(sum or median * n_sel of count * max_norm / norm) (:693-716).  col_sum[c] is the
 * linear statistic `sum` of column c (kpop_counter_stats with threshold 1, power 1).  *out_norm (may be NULL)
 * receives the accumulated norm the reference prints in verbose mode (:715,725).                                */
int kpop_counter_combine(const int32_t *const *columns, uint64_t n_rows, const uint32_t *sel, uint32_t n_sel,
                         const double *col_sum, int criterion, int32_t *out, double *out_norm);

/* Replaces the Transformation.compute loops of to_table / to_spectra, lib/KMerDB.ml:96-144,1040-1050,1140-1160,
 * 1203-1212: out[r*n_cols + c] (kmer_major = 1, the default table) or out[c*n_rows + r] (kmer_major = 0:
 * transposed table and spectra).  col_stats is n_cols x 4 from kpop_counter_stats with the same threshold/power. */
int kpop_counter_transform(const int32_t *const *columns, uint32_t n_cols, uint64_t n_rows, int which,
                           double threshold, double power, const double *col_stats, int kmer_major, double *out);

/* device-resident forms: storage is [n_cols][ld] int32 with ld = kpop_dev_counter_ld(n_rows) */
uint64_t kpop_dev_counter_ld(uint64_t n_rows);
uint64_t kpop_dev_counter_workspace_bytes(uint32_t n_cols, uint64_t n_rows);
int kpop_dev_counter_stats(const int32_t *d_storage, uint64_t ld, uint32_t n_cols, uint64_t n_rows,
                           double threshold, double power, void *d_workspace, double *d_col_stats,
                           double *d_row_stats, void *stream);
/* d_sel / d_norm list the n_valid selected spectra whose norm is positive, in visiting order; d_workspace needs
   kpop_dev_counter_workspace_bytes(n_valid, n_rows) */
int kpop_dev_counter_combine(const int32_t *d_storage, uint64_t ld, uint64_t n_rows, const uint32_t *d_sel,
                             const double *d_norm, uint32_t n_valid, uint32_t n_sel, double max_norm,
                             int criterion, void *d_workspace, int32_t *d_out, double *d_out_norm, void *stream);
int kpop_dev_counter_transform(const int32_t *d_storage, uint64_t ld, uint32_t n_cols, uint64_t n_rows,
                               int which, double threshold, double power, const double *d_col_stats,
                               int kmer_major, double *d_out, void *stream);
/* test hook: d_fast[i] = a[i] / b[i] by the reciprocal + FMA route the combination kernels use, d_exact[i] by
   the hardware division; the two must be bit-identical */
int kpop_dev_division_probe(const double *d_a, const double *d_b, uint64_t n, double *d_fast, double *d_exact,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif

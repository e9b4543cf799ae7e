// sort_count.h -- the sort path of k-mer counting (sort_count.hip) as the other translation units see it
#pragma once
#include "common.h"

namespace kpop {

struct SortedSpectra {  // device-resident CSR of a batch, plus the scratch it was built in (all from the caller's arena)
  DevBuf d_bases, d_off, d_woff, d_ka, d_kb, d_scr, d_start, d_sums, d_oh, d_oc, d_oo;
  uint64_t nu = 0;        // distinct (spectrum, k-mer) pairs = entries of d_oh / d_oc
  uint32_t n_spectra = 0;  // d_oo holds n_spectra + 1 offsets
};

int sorted_count_device(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content, int per_read,
                        uint64_t cap, SortedSpectra &S, hipStream_t st);
int sorted_count_batch(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int k, int content,
                       int per_read, uint64_t *out_hash, uint32_t *out_count, uint64_t *out_offsets, uint64_t cap,
                       uint64_t *n_written);

}  // namespace kpop

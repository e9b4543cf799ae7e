// gpu_workers.h -- several GPUs behind KPopTwistDB: one worker PROCESS per GPU.
//
// A HIP context does not survive fork(), so the workers are forked before the parent has made any HIP call; each one
// selects its GPU (kpop_init(i)), loads the twister archive itself, and then serves blocks of reads: the parent cuts
// every block of the reads stream (fast_seq.h) into contiguous shares, one per worker, and puts the twisted rows back
// together in the order of the reads -- after which the usual label bookkeeping applies (lib/Twister.ml:189-206), so
// the output is the single-GPU output.  Reads go to a worker over a pipe (150 bytes each); rows come back through a
// POSIX shared-memory segment (512 bytes each at 64 dimensions), named in the worker's reply.
// No collective: this stage shards with nothing to exchange.  KPOP_DEVICES=<n>|all turns it on.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "fast_seq.h"
#include "kpop_text.h"

namespace kpop_host {

struct TwisterSource {  // what a worker loads: the archive or tables the parent's -i T / -I T named
  std::string prefix;
  bool binary = true;
};

class GpuWorkers {
 public:
  // forks n workers; worker i drives GPU (i % n_visible) -- more workers than GPUs only happens in test rigs
  // (KPOP_DEVICES_SHARE=1).  Must be called before the calling process has touched HIP.
  GpuWorkers(int n, const TwisterSource &src);
  ~GpuWorkers();
  int size() const { return (int)w_.size(); }
  // rows of b (n x n_dims, in the order of the reads) into `out`; n_dims is learnt from the first reply
  void twist_block(const FlatBatch &b, int k, int content, bool normalize, size_t n_dims, double *out);

 private:
  struct Worker {
    int pid = -1, to = -1, from = -1;
  };
  std::vector<Worker> w_;
};

// how many worker processes the environment asks for (0 = stay in-process): KPOP_DEVICES = n | all
int devices_requested();

}  // namespace kpop_host

// gpu_workers.cpp -- see gpu_workers.h
#include "gpu_workers.h"

#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <thread>

#include "../../include/kpop_hip.h"
#include "ocaml_marshal.h"

namespace kpop_host {

int devices_requested() {
  const char *e = getenv("KPOP_DEVICES");
  if (!e || !*e) return 0;
  if (!strcmp(e, "all")) return -1;  // resolved by the first worker-free call that may touch HIP: see KPopTwistDB
  const int n = atoi(e);
  return n > 1 ? n : 0;
}

namespace {

struct BlockHeader {
  uint32_t magic, k, content, normalize, n_reads, pad;
  uint64_t n_bases;
};
constexpr uint32_t kBlockMagic = 0x4B57424Bu, kQuitMagic = 0x4B575154u;
struct Reply {
  int32_t status;
  uint32_t n_reads, n_dims, name_len;
};

[[noreturn]] void worker_main(int index, int from_parent, int to_parent, const TwisterSource &src) {
  // a process of its own: errors go back as a status + message, never as an exception across the pipe
  auto fail = [&](const std::string &msg) {
    Reply r{-1, 0, 0, (uint32_t)msg.size()};
    write_all(to_parent, &r, sizeof r);
    write_all(to_parent, msg.data(), msg.size());
  };
  kpop_twister *tw = nullptr;
  size_t name_len = 0;
  uint32_t n_dims = 0;
  std::string load_error;
  try {
    int n_dev = kpop_device_count();
    if (n_dev <= 0) throw Error(std::string("libkpop_hip: ") + kpop_last_error());
    if (kpop_init(index % n_dev) != 0) throw Error(std::string("libkpop_hip: ") + kpop_last_error());
    Table T, inertia;
    if (src.binary) read_binary_twister(make_filename(src.prefix, "KPopTwister", false), &T, &inertia);
    else T = read_table(make_filename(src.prefix, "KPopTwister", true));
    const size_t n = T.cols();
    std::vector<uint64_t> col_hash(n);
    name_len = n ? T.col_names[0].size() : 0;
    for (size_t c = 0; c < n; ++c)
      if (T.col_names[c].size() != name_len || !hex_to_hash(T.col_names[c], &col_hash[c]))
        throw Error("twister column '" + T.col_names[c] + "' is not a fixed-width hexadecimal k-mer hash");
    if (name_len > 15) throw Error("k-mer names longer than 15 hex digits");
    n_dims = (uint32_t)T.rows();
    const int k = (int)std::min<size_t>(2 * name_len, 30);
    if (kpop_twister_load(T.data.data(), n, n_dims, col_hash.data(), std::max(k, 1), &tw) != 0)
      throw Error(std::string("libkpop_hip: ") + kpop_last_error());
  } catch (const std::exception &e) {
    load_error = e.what();
  }
  FlatBatch b;
  std::vector<uint64_t> offsets;
  uint64_t serial = 0;
  for (;;) {
    BlockHeader h;
    try {
      if (!read_all(from_parent, &h, sizeof h) || h.magic == kQuitMagic) break;
      if (h.magic != kBlockMagic) break;
      b.lens.resize(h.n_reads);
      b.bases.resize(h.n_bases);
      if (h.n_reads && !read_all(from_parent, b.lens.data(), (size_t)h.n_reads * 4)) break;
      if (h.n_bases && !read_all(from_parent, b.bases.data(), h.n_bases)) break;
      if (!load_error.empty()) {
        fail(load_error);
        continue;
      }
      offsets.assign((size_t)h.n_reads + 1, 0);
      for (uint32_t r = 0; r < h.n_reads; ++r) offsets[r + 1] = offsets[r] + b.lens[r];
      char name[64];
      snprintf(name, sizeof name, "/kpop_rows_%d_%llu", (int)getpid(), (unsigned long long)serial++);
      const size_t bytes = std::max<size_t>(8, (size_t)h.n_reads * n_dims * 8);
      const int fd = shm_open(name, O_RDWR | O_CREAT | O_EXCL, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) {
        fail(std::string("shared memory for the twisted rows: ") + strerror(errno));
        if (fd >= 0) {
          close(fd);
          shm_unlink(name);
        }
        continue;
      }
      double *rows = (double *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (rows == MAP_FAILED) {
        shm_unlink(name);
        fail(std::string("mmap of the twisted rows: ") + strerror(errno));
        continue;
      }
      int rc = 0;
      const bool can_match = (size_t)name_digits((int)h.k, false) == name_len;
      if (h.n_reads && can_match) {
        static const uint8_t dummy = 0;
        rc = kpop_spectra_twist(tw, b.bases.empty() ? &dummy : b.bases.data(), offsets.data(), h.n_reads, (int)h.k, (int)h.content,
                                (int)h.normalize, rows);
      } else {
        memset(rows, 0, bytes);
      }
      munmap(rows, bytes);
      if (rc != 0) {
        shm_unlink(name);
        fail(std::string("libkpop_hip: ") + kpop_last_error());
        continue;
      }
      Reply r{0, h.n_reads, n_dims, (uint32_t)strlen(name)};
      write_all(to_parent, &r, sizeof r);
      write_all(to_parent, name, r.name_len);
    } catch (const std::exception &) {
      break;  // the parent is gone or the pipe broke
    }
  }
  if (tw) kpop_twister_free(tw);
  _exit(0);
}

}  // namespace

GpuWorkers::GpuWorkers(int n, const TwisterSource &src) {
  fflush(stdout);
  fflush(stderr);
  for (int i = 0; i < n; ++i) {
    int down[2], up[2];
    if (pipe(down) != 0 || pipe(up) != 0) throw Error(std::string("pipe: ") + strerror(errno));
    const pid_t pid = fork();
    if (pid < 0) throw Error(std::string("fork: ") + strerror(errno));
    if (pid == 0) {
      close(down[1]);
      close(up[0]);
      for (const Worker &o : w_) {  // the descriptors of the workers forked before this one
        close(o.to);
        close(o.from);
      }
      worker_main(i, down[0], up[1], src);
    }
    close(down[0]);
    close(up[1]);
    if (fcntl(down[1], F_SETPIPE_SZ, 1 << 20) < 0) {
    }
    Worker w;
    w.pid = pid;
    w.to = down[1];
    w.from = up[0];
    w_.push_back(w);
  }
  signal(SIGPIPE, SIG_IGN);  // a dead worker shows up as an error return, not as a signal
}

GpuWorkers::~GpuWorkers() {
  for (Worker &w : w_) {
    BlockHeader h{};
    h.magic = kQuitMagic;
    if (write(w.to, &h, sizeof h) < 0) {
    }
    close(w.to);
    close(w.from);
  }
  for (Worker &w : w_) waitpid(w.pid, nullptr, 0);
}

void GpuWorkers::twist_block(const FlatBatch &b, int k, int content, bool normalize, size_t n_dims, double *out) {
  const size_t n = b.size(), W = w_.size();
  std::vector<size_t> lo(W + 1), base_lo(W + 1, 0);
  for (size_t i = 0; i <= W; ++i) lo[i] = n * i / W;
  {
    size_t r = 0, at = 0;
    for (size_t i = 0; i <= W; ++i) {
      for (; r < lo[i]; ++r) at += b.lens[r];
      base_lo[i] = at;
    }
  }
  std::vector<std::string> err(W);
  std::vector<std::thread> senders;
  for (size_t i = 0; i < W; ++i)
    senders.emplace_back([&, i] {
      try {
        BlockHeader h{kBlockMagic, (uint32_t)k, (uint32_t)content, normalize ? 1u : 0u, (uint32_t)(lo[i + 1] - lo[i]), 0,
                      (uint64_t)(base_lo[i + 1] - base_lo[i])};
        write_all(w_[i].to, &h, sizeof h);
        if (h.n_reads) write_all(w_[i].to, b.lens.data() + lo[i], (size_t)h.n_reads * 4);
        if (h.n_bases) write_all(w_[i].to, b.bases.data() + base_lo[i], h.n_bases);
        Reply r;
        if (!read_all(w_[i].from, &r, sizeof r)) throw Error("worker " + std::to_string(i) + " has gone away");
        std::string text(r.name_len, '\0');
        if (r.name_len && !read_all(w_[i].from, &text[0], r.name_len)) throw Error("worker " + std::to_string(i) + " has gone away");
        if (r.status != 0) throw Error("GPU worker " + std::to_string(i) + ": " + text);
        if (r.n_reads != h.n_reads || r.n_dims != n_dims) {
          shm_unlink(text.c_str());
          throw Error("GPU worker " + std::to_string(i) + " returned a block of another shape");
        }
        const size_t bytes = std::max<size_t>(8, (size_t)r.n_reads * n_dims * 8);
        const int fd = shm_open(text.c_str(), O_RDONLY, 0);
        if (fd < 0) throw Error("cannot open the rows of worker " + std::to_string(i));
        void *m = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
        close(fd);
        shm_unlink(text.c_str());
        if (m == MAP_FAILED) throw Error("cannot map the rows of worker " + std::to_string(i));
        if (r.n_reads) memcpy(out + lo[i] * n_dims, m, (size_t)r.n_reads * n_dims * 8);
        munmap(m, bytes);
      } catch (const std::exception &e) {
        err[i] = e.what();
      }
    });
  for (std::thread &t : senders) t.join();
  for (const std::string &e : err)
    if (!e.empty()) throw Error(e);
}

}  // namespace kpop_host

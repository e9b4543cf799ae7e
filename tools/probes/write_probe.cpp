// write_probe -- how fast can 0.5 GB reach a file on this box?  (development probe behind ocaml_marshal.cpp's writer)
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <thread>
#include <vector>

static double now() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv) {
  const char *path = argc > 1 ? argv[1] : "/dev/shm/write_probe.bin";
  const size_t total = 522ull << 20;
  std::vector<char> src(total);
  for (size_t i = 0; i < total; i += 4096) src[i] = (char)i;
  auto par = [&](unsigned T, auto fn) {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; ++t) pool.emplace_back([&, t] { fn(total * t / T, total * (t + 1) / T); });
    for (auto &th : pool) th.join();
  };
  for (int rep = 0; rep < 2; ++rep) {
    {  // plain write
      double t0 = now();
      int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
      for (size_t o = 0; o < total;) o += (size_t)write(fd, src.data() + o, std::min<size_t>(8 << 20, total - o));
      close(fd);
      printf("write() 8 MB chunks, 1 thread:      %.3f s\n", now() - t0);
    }
    for (unsigned T : {4u, 16u}) {  // pwrite from threads
      double t0 = now();
      int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
      if (ftruncate(fd, total)) return 1;
      par(T, [&](size_t lo, size_t hi) {
        for (size_t o = lo; o < hi;) o += (size_t)pwrite(fd, src.data() + o, std::min<size_t>(8 << 20, hi - o), (off_t)o);
      });
      close(fd);
      printf("pwrite(), %2u threads:                %.3f s\n", T, now() - t0);
    }
    for (unsigned T : {4u, 8u, 16u, 32u}) {
      for (int populate = 0; populate < 2; ++populate) {
        double t0 = now();
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0666);
        if (ftruncate(fd, total)) return 1;
        char *m = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED | (populate ? MAP_POPULATE : 0), fd, 0);
        double t1 = now();
        par(T, [&](size_t lo, size_t hi) { memcpy(m + lo, src.data() + lo, hi - lo); });
        munmap(m, total);
        close(fd);
        printf("mmap%s + %2u threads memcpy:  %.3f s (map %.3f)\n", populate ? " POPULATE" : "         ", T, now() - t0, t1 - t0);
      }
    }
  }
  unlink(path);
  return 0;
}

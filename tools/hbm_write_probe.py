#!/usr/bin/env python3
"""What this HBM delivers for write-dominated streams (torch fill / copy / cast kernels), to set beside the
transformation kernels of csrc/counter.hip which write two bytes per byte read (development aid)."""
import torch

dev = torch.device("cuda", 0)
n = 1 << 29  # 4 GiB of f64
a = torch.empty(n, dtype=torch.float64, device=dev)
b = torch.empty(n, dtype=torch.float64, device=dev)
c = torch.randint(0, 40, (n,), dtype=torch.int32, device=dev)


def timeit(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t = timeit(lambda: a.fill_(1.0))
print("fill f64 (write only)        %.3f ms  %.0f GB/s" % (t, n * 8 / t / 1e6))
t = timeit(lambda: b.copy_(a))
print("copy f64 (1 read : 1 write)  %.3f ms  %.0f GB/s" % (t, n * 16 / t / 1e6))
t = timeit(lambda: torch.sum(a))
print("sum f64 (read only)          %.3f ms  %.0f GB/s" % (t, n * 8 / t / 1e6))
t = timeit(lambda: b.copy_(c))
print("int32 -> f64 (1 read : 2 write) %.3f ms  %.0f GB/s" % (t, n * 12 / t / 1e6))

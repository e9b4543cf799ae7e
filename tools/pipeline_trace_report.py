#!/usr/bin/env python3
"""Three-way overlap of the streaming pipeline from a rocprofv3 trace.

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 tools/pipeline_probe.py --trace
    python tools/pipeline_trace_report.py DIR > profiles/rNN_pipeline_overlap.txt

Takes the last `--window-ms` of the trace (the batches in flight), splits the activity into the upload engine (SDMA
host-to-device copies), the kernels of the compute stream, and the download (SDMA device-to-host copies plus the
runtime's blit kernel when it chooses that), and reports how long each was busy and how much of that time ran beside
the others."""
import argparse
import csv
import glob
import os


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    i = j = 0
    out = []
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--window-ms", type=float, default=15.0)
    a = ap.parse_args()
    mc = glob.glob(os.path.join(a.dir, "**", "*_memory_copy_trace.csv"), recursive=True)[0]
    kt = glob.glob(os.path.join(a.dir, "**", "*_kernel_trace.csv"), recursive=True)[0]
    up, down, comp, names = [], [], [], {}
    for r in csv.DictReader(open(mc)):
        iv = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
        if "HOST_TO_DEVICE" in r["Direction"]:
            up.append(iv)
        elif "DEVICE_TO_HOST" in r["Direction"]:
            down.append(iv)
    blit = 0
    for r in csv.DictReader(open(kt)):
        iv = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
        n = r["Kernel_Name"]
        if "rocclr_copyBuf" in n:
            down.append(iv)
            blit += 1
        elif n.startswith("kpop::") or "kpop::" in n:
            comp.append(iv)
            key = n.split("(")[0].replace("void ", "")[:60]
            names.setdefault(key, []).append(iv[1] - iv[0])
    t_end = max(x[1] for x in up + down + comp)
    t0 = t_end - int(a.window_ms * 1e6)
    clip = lambda ivs: [(max(s, t0), e) for s, e in ivs if e > t0]
    U, D, K = union(clip(up)), union(clip(down)), union(clip(comp))
    span = t_end - min(x[0] for x in (U + D + K))
    busy_any = total(union([tuple(x) for x in U + D + K]))
    print("window: last %.1f ms of the trace (activity spans %.3f ms, something is running for %.3f ms of it)" % (a.window_ms, span / 1e6, busy_any / 1e6))
    print("engine                      busy ms   of span")
    for name, iv in (("upload   (SDMA H2D)", U), ("kernels  (compute stream)", K), ("download (SDMA D2H + blit)", D)):
        print("  %-26s %7.3f   %5.1f %%" % (name, total(iv) / 1e6, 100.0 * total(iv) / span))
    print("download beside kernels:  %.3f ms = %.1f %% of the download time" % (total(intersect(D, K)) / 1e6, 100.0 * total(intersect(D, K)) / max(total(D), 1)))
    print("upload beside kernels:    %.3f ms = %.1f %% of the upload time" % (total(intersect(U, K)) / 1e6, 100.0 * total(intersect(U, K)) / max(total(U), 1)))
    print("all three at once:        %.3f ms" % (total(intersect(intersect(U, K), D)) / 1e6))
    print("sum of the three busy times / span = %.2f  (1.00 = no overlap at all, 3.00 = perfect three-way overlap)" % ((total(U) + total(D) + total(K)) / span))
    print("device-to-host copies in the window: %d on the SDMA engines, %d as the runtime's blit kernel" % (len(clip(down)) - sum(1 for _ in []), blit))
    print()
    print("kernels of the compute stream (whole trace): calls, average us")
    for k, v in sorted(names.items(), key=lambda kv: -sum(kv[1])):
        print("  %-60s %6d  %9.1f" % (k, len(v), sum(v) / len(v) / 1e3))


if __name__ == "__main__":
    main()

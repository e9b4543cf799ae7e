#!/usr/bin/env python3
"""bench.py's pcie_inclusive leg, several times in one process: how steady the in-flight figures are (both outputs /
distances only), with the host threads of the process's OpenMP runtimes as they come or OMP_WAIT_POLICY=passive."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    args = bench.parse_args(["--no-cpu-baseline"])
    R = bench.Rank(args)
    R.run_headline(100000, 0, 5, 2, want_outputs=0)
    for i in range(int(os.environ.get("REPS", "4"))):
        p = bench.pcie_inclusive(R, 100000)
        print("run %d  OMP_WAIT_POLICY=%s  both %.1f M/s (single %.1f)  distances only %.1f  summary only %.1f"
              % (i, os.environ.get("OMP_WAIT_POLICY", "-"), p["value"] / 1e6, p["single_call"]["value"] / 1e6, p["distances_only"]["value"] / 1e6,
                 p["summary_only"]["value"] / 1e6), flush=True)


if __name__ == "__main__":
    main()

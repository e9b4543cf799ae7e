#!/usr/bin/env bash
# File-to-file throughput of the drop-in CLIs on 100k x 150 bp synthetic reads, k=12 (development aid).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
export PATH="$ROOT/kpop_amd/bin:$PATH"
W=$(mktemp -d)
cd "$W"
python3 - <<PY
import sys
sys.path.insert(0, "$ROOT")
from oracle import oracle as O
import numpy as np
n, L = 100000, 150
b, o = O.synth_reads(0x4B506F70, n, L)
with open("reads.fa", "w") as f:
    s = bytes(b).decode()
    f.write("".join(">r%d\n%s\n" % (i, s[i*L:(i+1)*L]) for i in range(n)))
k, d = 12, 64
PY
ls -la reads.fa | awk '{print "reads.fa", $5, "bytes"}'
T0=$(date +%s.%N); KPopCount -k 12 -L -f reads.fa -o spectra; T1=$(date +%s.%N); echo "KPopCount -L (100k reads -> spectra text): $(python3 -c "print(round($T1 - $T0, 3))") s wall"
ls -la spectra.KPopSpectra.txt | awk '{print "spectra text", $5, "bytes"}'
T0=$(date +%s.%N); KPopCount -k 12 -l all -f reads.fa -o merged; T1=$(date +%s.%N); echo "KPopCount -l (merged): $(python3 -c "print(round($T1 - $T0, 3))") s wall"
rm -rf "$W"

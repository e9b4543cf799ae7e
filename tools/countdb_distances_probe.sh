#!/usr/bin/env bash
# spectral distances between 65 class spectra over all 524,800 canonical 10-mers (KPopCountDB --distances), timed
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
export PATH="$ROOT/kpop_amd/bin:$PATH"
W=$(mktemp -d); cd "$W"
python3 - <<PY
import sys
sys.path.insert(0, "$ROOT")
from oracle import oracle as O
C, G = 65, 300000
b, o = O.synth_reads(0xC1A55, C, G)
s = bytes(b).decode()
open("all.fa", "w").write("".join(">g%02d\n%s\n" % (c, s[c*G:(c+1)*G]) for c in range(C)))
PY
KPopCount -k 10 -L -f all.fa | KPopCountDB -k /dev/stdin -o Classes --summary 2>&1 | head -c 200; echo
for i in 1 2; do T0=$(date +%s.%N); KPopCountDB -i Classes --distances '~.' '~.' D; T1=$(date +%s.%N); python3 -c "print('KPopCountDB --distances 65 x 65 over 524,800 k-mers: %.3f s' % ($T1 - $T0))"; done
KPopTwistDB -i d D -O d /dev/stdout | head -3 | cut -c1-150
rm -rf "$W"

#!/usr/bin/env bash
# README-shaped training + classification, file to file, through the drop-in CLIs (development aid):
# 65 synthetic 30 kb class genomes -> KPopCount -l | KPopCountDB -> KPopTwist (CA on the GPU) -> 100k x 150 bp reads
# -> KPopCount -L | KPopTwistDB -k -> KPopTwistDB -s.  Prints the wall time of every stage.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
export PATH="$ROOT/kpop_amd/bin:$PATH"
K=${K:-10}; NREADS=${NREADS:-100000}
W=$(mktemp -d)
cd "$W"
python3 - <<PY
import sys
sys.path.insert(0, "$ROOT")
from oracle import oracle as O
n, L = $NREADS, 150
b, o = O.synth_reads(0x4B506F70, n, L)
s = bytes(b).decode()
open("reads.fa", "w").write("".join(">r%d\n%s\n" % (i, s[i*L:(i+1)*L]) for i in range(n)))
C, G = 65, 30000
b, o = O.synth_reads(0xC1A55, C, G)
s = bytes(b).decode()
for c in range(C):
    open("class%02d.fa" % c, "w").write(">g%d\n%s\n" % (c, s[c*G:(c+1)*G]))
PY
stage() { local t0=$(date +%s.%N); "$@"; local t1=$(date +%s.%N); python3 -c "print('  %-58s %7.3f s' % ('$STAGE', $t1 - $t0))"; }
STAGE="train: 65 x (KPopCount -l) | KPopCountDB -o Classes" stage bash -c 'for f in class*.fa; do KPopCount -k '$K' -l ${f%.fa} -f $f; done | KPopCountDB -k /dev/stdin -o Classes'
STAGE="train: KPopTwist -i Classes -o Classes (CA on the GPU)" stage KPopTwist -i Classes -o Classes
ls -la Classes.KPopTwister | awk '{print "  twister file", $5, "bytes"}'
STAGE="classify: KPopCount -L -f reads.fa -o spectra" stage KPopCount -k $K -L -f reads.fa -o spectra
ls -la spectra.KPopSpectra.txt | awk '{print "  spectra text", $5, "bytes"}'
STAGE="classify: KPopTwistDB -i T Classes -k spectra -o t Test" stage KPopTwistDB -i T Classes -k spectra.KPopSpectra.txt -o t Test
STAGE="classify: KPopCount -L | KPopTwistDB -k /dev/stdin (piped)" stage bash -c "KPopCount -k $K -L -f reads.fa | KPopTwistDB -i T Classes -k /dev/stdin -o t Test2"
STAGE="classify: KPopTwistDB -i T Classes -i t Classes -s Test out" stage KPopTwistDB -i T Classes -i t Classes -s Test out
STAGE="classify: KPopTwistDB ... -d Test -o d D (65 x 100k matrix)" stage KPopTwistDB -i T Classes -i t Classes -d Test -o d D
head -c 300 out.KPopSummary.txt | head -2
cmp Test.KPopTwisted Test2.KPopTwisted && echo "  piped == file-based twisted: identical"
rm -rf "$W"

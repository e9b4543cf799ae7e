#!/bin/bash
# Stage marks (KPOP_TIMING=1) of the text-spectra route on N reads: KPopCount -L to a file, KPopTwistDB -k from that file,
# and the two joined by a pipe with KPOP_PIPE_FORMAT=text.   tools/probes/f2f_text_stages.sh [reads=1000000] [k=12]
set -e
N=${1:-1000000}; K=${2:-12}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
export PATH=$ROOT/kpop_amd/bin:$PATH
W=$(mktemp -d /dev/shm/kpop_f2t_XXXX); trap 'rm -rf $W' EXIT; cd $W
now() { python3 -c "import time; print('%.3f' % time.time())"; }
kpop_synth genomes --n 65 --len 30000 --seed 12648430 > classes.fa
KPopCount -k $K -L -f classes.fa 2>/dev/null | KPopCountDB -k /dev/stdin -o Classes 2>/dev/null
KPopTwist -i Classes -o Classes 2>/dev/null
kpop_synth reads --from classes.fa --n $N --len 150 --mutate 0.005 --seed 1263555440 > reads.fa
for rep in 1 2; do
  echo "== KPopCount -L -f reads.fa -o Reads (text spectra to a file), rep $rep"
  t0=$(now); KPOP_TIMING=1 KPopCount -k $K -L -f reads.fa -o Reads 2>&1 | grep -v "block parsed\|block handed" | tail -8; t1=$(now)
  ls -la Reads.KPopSpectra.txt | awk '{print $5, "bytes"}'
  python3 -c "print('wall %.3f s' % ($t1 - $t0))"
  echo "== KPopTwistDB -k Reads.KPopSpectra.txt, rep $rep"
  t0=$(now); KPOP_TIMING=1 KPopTwistDB -i T Classes -k Reads.KPopSpectra.txt -o t Test 2>&1 | tail -12; t1=$(now)
  python3 -c "print('wall %.3f s' % ($t1 - $t0))"
done
echo "== the two through a pipe, text spectra"
t0=$(now); KPOP_PIPE_FORMAT=text KPopCount -k $K -L -f reads.fa | KPOP_TIMING=1 KPopTwistDB -i T Classes -k /dev/stdin -o t Test2 2>&1 | tail -12; t1=$(now)
python3 -c "print('wall %.3f s' % ($t1 - $t0))"
cmp Test.KPopTwisted Test2.KPopTwisted && echo "same twisted file"

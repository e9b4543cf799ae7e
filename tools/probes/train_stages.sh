#!/bin/bash
# Stage marks of the training commands (README.md:91-93) on C class genomes of 30 kb.  tools/probes/train_stages.sh [classes=65] [k=12]
set -e
C=${1:-65}; K=${2:-12}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
export PATH=$ROOT/kpop_amd/bin:$PATH
W=$(mktemp -d /dev/shm/kpop_train_XXXX); trap 'rm -rf $W' EXIT; cd $W
now() { python3 -c "import time; print('%.3f' % time.time())"; }
kpop_synth genomes --n $C --len 30000 --seed 12648430 > classes.fa
for rep in 1 2; do
  t0=$(now); KPOP_TIMING=1 KPopCount -k $K -L -f classes.fa 2>/dev/null | KPOP_TIMING=1 KPopCountDB -k /dev/stdin -o Classes 2>&1 | tail -12 | grep -v '^$'; t1=$(now)
  python3 -c "print('KPopCount | KPopCountDB: wall %.3f s' % ($t1 - $t0))"
  t0=$(now); KPOP_TIMING=1 KPopTwist -i Classes -o Classes 2>&1 | grep -v 'kpop_ca\]' | tail -30; t1=$(now)
  python3 -c "print('KPopTwist: wall %.3f s' % ($t1 - $t0))"
done
ls -la Classes.* | awk '{print $5, $9}'

#!/bin/bash
# Stage marks (KPOP_TIMING=1) of the README's count | twist pipeline and of the summary on N reads; works in /dev/shm.
#   tools/probes/f2f_stages.sh [reads=4000000] [k=12]
set -e
N=${1:-4000000}; K=${2:-12}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
export PATH=$ROOT/kpop_amd/bin:$PATH
W=$(mktemp -d /dev/shm/kpop_f2f_XXXX); trap 'rm -rf $W' EXIT; cd $W
kpop_synth genomes --n 65 --len 30000 --seed 12648430 > classes.fa
KPopCount -k $K -L -f classes.fa 2>/dev/null | KPopCountDB -k /dev/stdin -o Classes 2>/dev/null
KPopTwist -i Classes -o Classes 2>/dev/null
kpop_synth reads --from classes.fa --n $N --len 150 --mutate 0.005 --seed 1263555440 > reads.fa
ls -la reads.fa Classes.KPopTwister | awk '{print $5, $9}'
echo "== the producer alone (reads stream into cat)"
t0=$(date +%s.%N)
KPOP_PIPE_FORMAT=reads KPopCount -k $K -L -f reads.fa | cat > /dev/null
python3 -c "import time,sys; print('wall %.3f s' % (time.time() - float(sys.argv[1])))" $t0
for rep in 1 2; do
  echo "== count | twist, rep $rep"
  t0=$(date +%s.%N)
  { KPOP_TIMING=1 KPopCount -k $K -L -f reads.fa | KPOP_TIMING=1 KPopTwistDB -i T Classes -k /dev/stdin -o t Test; } 2>&1 | grep -v "stream block\|block parsed\|block handed" | tail -30
  python3 -c "import time,sys; print('wall %.3f s' % (time.time() - float(sys.argv[1])))" $t0
done
echo "== summary"
t0=$(date +%s.%N)
KPOP_TIMING=1 KPopTwistDB -i T Classes -i t Classes -s Test Summary 2>&1 | tail -20
python3 -c "import time,sys; print('wall %.3f s' % (time.time() - float(sys.argv[1])))" $t0

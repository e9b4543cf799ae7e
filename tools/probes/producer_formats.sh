#!/bin/bash
# KPopCount as the producer of the reads stream, alone (no GPU is touched): FASTA, FASTQ and paired FASTQ, N reads.
#   tools/probes/producer_formats.sh [reads=4000000]
set -e
N=${1:-4000000}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
export PATH=$ROOT/kpop_amd/bin:$PATH
W=$(mktemp -d /dev/shm/kpop_pf_XXXX); trap 'rm -rf $W' EXIT; cd $W
kpop_synth genomes --n 20 --len 30000 --seed 1 > c.fa
kpop_synth reads --from c.fa --n $N --len 150 --mutate 0.005 --seed 3 > reads.fa
python3 - <<'PY'
with open('reads.fa') as f, open('reads.fq','w') as o, open('m_1.fq','w') as o1, open('m_2.fq','w') as o2:
    i = 0
    while True:
        h = f.readline()
        if not h: break
        s = f.readline().rstrip('\n')
        rec = '@' + h[1:] + s + '\n+\n' + 'I' * len(s) + '\n'
        o.write(rec)
        (o1 if i % 2 == 0 else o2).write(rec)
        i += 1
PY
for rep in 1 2; do
  for args in "-f reads.fa" "-s reads.fq" "-p m_1.fq m_2.fq"; do
    python3 - "$args" <<'PY'
import subprocess, sys, time
t0 = time.time()
subprocess.run("KPOP_PIPE_FORMAT=reads KPopCount -k 12 -L %s | cat > /dev/null" % sys.argv[1], shell=True, check=True)
print("%-22s %.3f s" % (sys.argv[1], time.time() - t0), flush=True)
PY
  done
done

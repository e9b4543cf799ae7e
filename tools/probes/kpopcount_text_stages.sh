set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; export PATH=$ROOT/kpop_amd/bin:$PATH
W=$(mktemp -d /dev/shm/kc_XXXX); trap 'rm -rf $W' EXIT; cd $W
kpop_synth genomes --n 65 --len 30000 --seed 12648430 > classes.fa
kpop_synth reads --from classes.fa --n 1000000 --len 150 --mutate 0.005 --seed 1263555440 > reads.fa
KPOP_TIMING=1 KPopCount -k 12 -L -f reads.fa -o Reads 2>&1 | tail -12
KPOP_TIMING=1 KPopCount -k 12 -L -f reads.fa -o Reads 2>&1 | tail -12

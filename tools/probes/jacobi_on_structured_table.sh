set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; export PATH=$ROOT/kpop_amd/bin:$PATH
W=$(mktemp -d /dev/shm/jt_XXXX); trap 'rm -rf $W' EXIT; cd $W
kpop_synth genomes --n 40 --len 30000 --seed 5 > base.fa
for i in $(seq 1 25); do kpop_synth mutants --from base.fa --n 40 --mutate 0.01 --seed $i | sed "s/^>/>m${i}_/"; done > genomes.fa
# 25 x 40 mutants of the FIRST genome only; add the 40 distinct genomes too, and mutants of each
cat base.fa >> genomes.fa
for g in $(seq 2 12); do awk -v g=$g 'BEGIN{RS=">";ORS=""} NR==g+1{print ">"$0}' base.fa > one.fa; kpop_synth mutants --from one.fa --n 50 --mutate 0.02 --seed $((100+g)) | sed "s/^>/>g${g}_/" >> genomes.fa; done
grep -c ">" genomes.fa
KPopCount -k 10 -L -f genomes.fa | KPopCountDB -k /dev/stdin -o DB
for dbg in 0 2048; do echo "KPOP_TUNE_DBG=$dbg"; KPOP_TUNE_DBG=$dbg KPOP_JACOBI_TRACE=1 KPOP_TIMING=1 KPopTwist -i DB -o TW$dbg 2>&1 | grep -E "jacobi\]|Jacobi  " | tail -6; done

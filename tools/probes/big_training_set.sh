set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; export PATH=$ROOT/kpop_amd/bin:$PATH
W=$(mktemp -d /dev/shm/bt_XXXX); trap 'rm -rf $W' EXIT; cd $W
now() { python3 -c "import time; print('%.3f' % time.time())"; }
kpop_synth genomes --n 20 --len 30000 --seed 5 > base.fa
for i in $(seq 1 50); do kpop_synth mutants --from base.fa --n 20 --mutate 0.01 --seed $i | sed "s/^>/>m${i}_/"; done > genomes.fa
grep -c ">" genomes.fa
t0=$(now); KPopCount -k 12 -L -f genomes.fa -o G; t1=$(now)
ls -la G.KPopSpectra.txt | awk '{print $5, "bytes of spectra"}'; wc -l G.KPopSpectra.txt
python3 -c "print('KPopCount -L: %.2f s' % ($t1 - $t0))"
t0=$(now); KPOP_TIMING=1 KPopCountDB -k G -o DB 2>&1 | tail -4; t1=$(now)
python3 -c "print('KPopCountDB -k -o: %.2f s' % ($t1 - $t0))"
ls -la DB.KPopCounter | awk '{print $5, "bytes of database"}'
t0=$(now); KPOP_TIMING=1 KPopTwist -i DB -o TW 2>&1 | tail -24; t1=$(now)
python3 -c "print('KPopTwist: %.2f s' % ($t1 - $t0))"
ls -la TW.KPopTwister | awk '{print $5, "bytes of twister"}'

#!/usr/bin/env bash
# how long a drop-in CLI takes to do nothing: process start + kpop_init + one tiny launch (development aid)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
export PATH="$ROOT/kpop_amd/bin:$PATH"
W=$(mktemp -d); cd "$W"
printf '>r\nACGTACGTACGTACGTACGTAGCTAGCTAGCATCGATCGATGCATGC\n' > t.fa
for i in 1 2 3 4; do
  T0=$(date +%s.%N); KPopCount -k 10 -L -f t.fa > /dev/null; T1=$(date +%s.%N)
  python3 -c "print('KPopCount tiny: %.3f s' % ($T1 - $T0))"
done
T0=$(date +%s.%N); KPopCount -V > /dev/null; T1=$(date +%s.%N); python3 -c "print('KPopCount -V (no GPU init): %.3f s' % ($T1 - $T0))"
AMD_LOG_LEVEL=0 HIP_VISIBLE_DEVICES=0 bash -c 'T0=$(date +%s.%N); KPopCount -k 10 -L -f t.fa > /dev/null; T1=$(date +%s.%N); python3 -c "print(\"with HIP_VISIBLE_DEVICES=0: %.3f s\" % ($T1 - $T0))"'
python3 - <<PY
import ctypes, time
t0 = time.perf_counter()
L = ctypes.CDLL("$ROOT/kpop_amd/libkpop_hip.so")
t1 = time.perf_counter()
rc = L.kpop_init(0)
t2 = time.perf_counter()
print("dlopen %.3f s, kpop_init %.3f s (rc %d)" % (t1 - t0, t2 - t1, rc))
PY
rm -rf "$W"

#!/bin/bash
# The host boundary (CLIs, text parsers, OCaml-Marshal readers) under AddressSanitizer + UBSan, CPU only -- never on the GPU box.
#   bash tools/sanitize_host.sh [fuzz seconds, default 60]  ->  profiles/r05_host_sanitizers.txt
set -u
cd "$(dirname "$0")/.." || exit 1
SECS=${1:-60}
OUT=${KPOP_SANITIZE_OUT:-profiles/r05_host_sanitizers.txt}
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
{
  echo "host boundary under g++ -fsanitize=address,undefined (make -C kpop_amd/host asan), $(date -u +%Y-%m-%dT%H:%MZ), $(g++ --version | head -1)"
  echo "== build"
  make -C kpop_amd/host asan -j6 2>&1 | grep -ci "warning\|error" | sed 's/^/warnings+errors: /'
  echo "== tests/test_cli.py + tests/test_host_parsers.py with KPOP_TEST_BIN=kpop_amd/bin_asan KPOP_TEST_SANITIZE=1 (the parser harnesses sanitized too)"
  KPOP_TEST_SANITIZE=1 KPOP_TEST_BIN=$PWD/kpop_amd/bin_asan python -m pytest tests/test_cli.py tests/test_host_parsers.py -q -x 2>&1 | tail -3
  echo "== mutation loop over the Marshal readers, ${SECS} s (tests/host/marshal_fuzz.cpp)"
  mkdir -p /tmp/kpop_fuzz && kpop_amd/bin_asan/marshal_fuzz /tmp/kpop_fuzz "$SECS" 2>&1 | tail -5
  echo "exit code of the fuzz loop: $?"
} | tee "$OUT"

#!/bin/bash
# The host-only pieces of libfastani_hip (fa_host.h packer + thread pool, fa_fasta.h, fa_stats.h, fa_lease.h) under
# AddressSanitizer + UndefinedBehaviorSanitizer and, separately, ThreadSanitizer -- on the CPU, g++ only (no GPU sanitizer,
# no XNACK).  Every build runs with the AVX2 packer and with FA_NO_AVX2=1 (the scalar paths), with 8 and with 3 host threads.
#   bash scripts/host_sanitize.sh [log]        (default log: profiles/r06_host_sanitizers.txt)
set -u
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r06_host_sanitizers.txt}
OUT=build/host_sanitize
mkdir -p "$OUT"
SRC=scripts/host_sanitize/driver.cpp
status=0
{
  echo "# scripts/host_sanitize.sh at $(git rev-parse --short HEAD 2>/dev/null || echo unknown), $(g++ --version | head -1)"
  for kind in asan_ubsan tsan; do
    if [ $kind = asan_ubsan ]; then FLAGS="-fsanitize=address,undefined -fno-sanitize-recover=all"; else FLAGS="-fsanitize=thread"; fi
    echo "## build: g++ -std=c++17 -O1 -g -fno-omit-frame-pointer $FLAGS -pthread $SRC"
    if ! g++ -std=c++17 -O1 -g -fno-omit-frame-pointer $FLAGS -pthread "$SRC" -o "$OUT/driver_$kind" 2>&1; then echo "BUILD FAILED"; status=1; continue; fi
    for env in "FA_HOST_THREADS=8" "FA_HOST_THREADS=8 FA_NO_AVX2=1" "FA_HOST_THREADS=3 FA_PACK_CHUNK=4096"; do
      echo "### run ($kind): $env"
      if env $env ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 TSAN_OPTIONS=halt_on_error=0 "$OUT/driver_$kind" 2>&1 | tail -40; then :; fi
      rc=${PIPESTATUS[0]}
      echo "exit status $rc"
      [ "$rc" = 0 ] || status=1
    done
  done
  echo "# overall: $([ $status = 0 ] && echo clean || echo FINDINGS)"
} | tee "$LOG"
exit $status

#!/bin/bash
# round 5: every profile behind the bench line, on the head
set -u
export TMPDIR=/tmp
for mode in default batch16 config3 config4 config5:k21f3000 config5:k16f1000; do
  echo "== $mode"; timeout 1500 bash scripts/collect_profiles.sh r05 $mode 2>&1 | tail -2
done
ls gpurun_out | grep r05 | head -40

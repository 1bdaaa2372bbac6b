#!/bin/bash
# round 5: HBM traffic of the remaining live cells of config 5 (the two slowest are in already)
set -u
export TMPDIR=/tmp
for cell in k14f1000 k14f3000 k14f5000 k16f3000 k16f5000 k21f5000; do
  echo "== $cell"; timeout 900 bash scripts/collect_profiles.sh r05 config5:$cell 2>&1 | tail -1
done

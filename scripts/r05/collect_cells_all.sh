#!/bin/bash
# round 5, at the head: kernel stats + HBM traffic of all eight live cells of config 5 again (the row formation of their passes changed)
set -u
export TMPDIR=/tmp
for cell in k14f1000 k14f3000 k14f5000 k16f1000 k16f3000 k16f5000 k21f3000 k21f5000; do
  echo "== $cell"; timeout 900 bash scripts/collect_profiles.sh r05 config5:$cell 2>&1 | tail -1
done

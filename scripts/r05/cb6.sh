#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_cb6
mkdir -p $OUT
python3 scripts/time_config3.py 20 50 5000000 3 2>/dev/null | tail -1 | tee $OUT/time_20x50.json
FA_L1_NEAR=1 timeout 900 python3 scripts/fuzz_parity.py 3000 53001 2>&1 | tail -1 | tee $OUT/fuzz.txt

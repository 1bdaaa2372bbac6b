#!/bin/bash
# round 5, fourth part of the fuzz campaign: the library of the round's last kernel change (reference flush uploading behind the hashing)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_fuzz_campaign4.txt
{
  python3 scripts/fuzz_parity.py 40000 54001 2>&1 | tail -1
  FA_K1_GENERAL=1 python3 scripts/fuzz_parity.py 5000 54002 2>&1 | tail -1 | sed 's/$/      FA_K1_GENERAL=1/'
  FA_L1_NEAR=1 FA_ROWS_EMIT_MAX=16384 python3 scripts/fuzz_parity.py 5000 54003 2>&1 | tail -1 | sed 's/$/      FA_L1_NEAR=1 FA_ROWS_EMIT_MAX=16384/'
} | tee $OUT

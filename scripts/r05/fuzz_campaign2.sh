#!/bin/bash
# round 5, second part of the fuzz campaign: at the head (one-genome workgroup order, arena reader)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_fuzz_campaign2.txt
{
  python3 scripts/fuzz_parity.py 40000 52001 2>&1 | tail -1
  FA_L1_NEAR=1 python3 scripts/fuzz_parity.py 25000 52002 2>&1 | tail -1 | sed 's/$/      FA_L1_NEAR=1/'
  python3 scripts/fuzz_parity.py 15000 52003 0 1 2>&1 | tail -1 | sed 's/$/      default cell only/'
  FA_FRAG_ORDER_ONE=0 python3 scripts/fuzz_parity.py 5000 52004 2>&1 | tail -1 | sed 's/$/      FA_FRAG_ORDER_ONE=0/'
  FA_L1_NEAR=1 FA_L1_BLOCK_SORT=0 python3 scripts/fuzz_parity.py 5000 52005 2>&1 | tail -1 | sed 's/$/      FA_L1_NEAR=1 FA_L1_BLOCK_SORT=0 (a third of the cases on 64-bit coordinates: the fuzzer draws FA_GPOS_BITS itself)/'
} | tee $OUT

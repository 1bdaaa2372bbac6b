#!/bin/bash
# round 5 differential fuzz campaign at the head (every hit, every L2 mapping, index size, threshold against the oracle)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_fuzz_campaign.txt
{
  echo "# scripts/fuzz_parity.py <cases> <seed> [time box] [default cell] at the head of round 5"
  python3 scripts/fuzz_parity.py 20000 51001 2>&1 | tail -1
  FA_L1_NEAR=1 python3 scripts/fuzz_parity.py 12000 51002 2>&1 | tail -1 | sed 's/$/      FA_L1_NEAR=1 (coordinate filter of large indices forced on)/'
  FA_L1_NEAR=1 FA_L1_BLOCK_SORT=0 python3 scripts/fuzz_parity.py 4000 51003 2>&1 | tail -1 | sed 's/$/      FA_L1_NEAR=1 FA_L1_BLOCK_SORT=0/'
  FA_L1_NEAR=1 python3 scripts/fuzz_parity.py 6000 51004 0 1 2>&1 | tail -1 | sed 's/$/      FA_L1_NEAR=1, default cell only/'
  FA_PASS_FRAGMENTS=40 python3 scripts/fuzz_parity.py 3000 51005 2>&1 | tail -1 | sed 's/$/      FA_PASS_FRAGMENTS=40/'
  FA_K1_TILE=1024 python3 scripts/fuzz_parity.py 3000 51006 2>&1 | tail -1 | sed 's/$/      FA_K1_TILE=1024 (full reference tiles)/'
  FA_POOL_MAX_GB=0 python3 scripts/fuzz_parity.py 2000 51007 2>&1 | tail -1 | sed 's/$/      FA_POOL_MAX_GB=0 (no device pool)/'
} | tee $OUT

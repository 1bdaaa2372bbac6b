#!/bin/bash
# round 5, third part of the fuzz campaign: at the head (index-build kernels on LDS tiles / block table / length histogram, rows of
# a pass formed inside k_cgi_rows only below 512 pairs)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_fuzz_campaign3.txt
{
  python3 scripts/fuzz_parity.py 30000 53001 2>&1 | tail -1
  FA_LINK_BLOCK_BITS=3 python3 scripts/fuzz_parity.py 10000 53002 2>&1 | tail -1 | sed 's/$/      FA_LINK_BLOCK_BITS=3 (blocks of 8 records: inside a contig and straddling)/'
  FA_ROWS_EMIT_MAX=1 python3 scripts/fuzz_parity.py 10000 53003 2>&1 | tail -1 | sed 's/$/      FA_ROWS_EMIT_MAX=1 (rows formed by kernels of their own)/'
  FA_FREQ_OVER_CAP=1 python3 scripts/fuzz_parity.py 5000 53004 2>&1 | tail -1 | sed 's/$/      FA_FREQ_OVER_CAP=1/'
} | tee $OUT

#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_scale
mkdir -p $OUT
FA_TRACE=1 timeout 170 python3 scripts/r05/scale_probe2.py 10 > $OUT/probe2b.txt 2>&1
echo "exit $?" >> $OUT/probe2b.txt
grep -v "sketch flush" $OUT/probe2b.txt | tail -12 | cut -c1-260
timeout 120 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "links or index or end_to_end or frequency" 2>&1 | tail -2

#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_scale
mkdir -p $OUT
FA_TRACE=1 timeout 400 python3 scripts/r05/scale_probe.py 80 50 2 > $OUT/probe_4000.txt 2>&1
echo "exit $?" >> $OUT/probe_4000.txt
tail -25 $OUT/probe_4000.txt | cut -c1-250

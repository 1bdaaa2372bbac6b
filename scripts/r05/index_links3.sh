#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_index_links
mkdir -p $OUT
for v in 0 1; do
if [ $v = 1 ]; then export FA_FREQ_SORT=1; fi
FA_TRACE=1 timeout 900 python3 scripts/time_index.py 1000 5000000 2 > $OUT/time_index_1000c$v.json 2> $OUT/trace_1000c$v.txt
echo "sort=$v"; grep "fa trace" $OUT/trace_1000c$v.txt | grep build_index | tail -8
done

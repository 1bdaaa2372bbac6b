#!/bin/bash
# round 5: where the wall clock of the 1000-genome reference build goes (host stages vs device kernels), FA_TRACE=1
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_trace_index
mkdir -p $OUT
FA_TRACE=1 python3 scripts/time_index.py 1000 5000000 2 > $OUT/time_index_1000.json 2> $OUT/trace_1000.txt
cat $OUT/time_index_1000.json; grep "fa trace" $OUT/trace_1000.txt | tail -12
FA_TRACE=1 python3 scripts/time_index.py 100 5000000 2 > $OUT/time_index_100.json 2> $OUT/trace_100.txt
cat $OUT/time_index_100.json; grep "fa trace" $OUT/trace_100.txt | tail -6

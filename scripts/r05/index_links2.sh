#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_index_links
mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "links or index or frequency or end_to_end or draft" > $OUT/pytest2.txt 2>&1
tail -4 $OUT/pytest2.txt
FA_TRACE=1 timeout 900 python3 scripts/time_index.py 1000 5000000 2 > $OUT/time_index_1000b.json 2> $OUT/trace_1000b.txt
cat $OUT/time_index_1000b.json; grep "fa trace" $OUT/trace_1000b.txt | tail -8

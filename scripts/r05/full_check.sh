#!/bin/bash
# round 5: the whole GPU suite on the head, the index-build trace with the device pool, a leak check and the default bench line
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_full
mkdir -p $OUT
timeout 3000 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
FA_TRACE=1 python3 scripts/time_index.py 1000 5000000 3 > $OUT/time_index_1000.json 2> $OUT/trace_1000.txt
cat $OUT/time_index_1000.json; grep "fa trace" $OUT/trace_1000.txt | tail -8
python3 scripts/check_leaks.py > $OUT/leaks.json 2>$OUT/leaks.err; cat $OUT/leaks.json
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 6000 $OUT/bench_default.json

#!/bin/bash
# round 5, at the head: the whole GPU suite, the step's profiles again (the one-genome workgroup order changed its traffic), the default bench line
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_final
mkdir -p $OUT
timeout 3000 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 1200 bash scripts/collect_profiles.sh r05 default 2>&1 | tail -1
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json; tail -3 $OUT/bench_default.err

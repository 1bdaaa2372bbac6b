#!/bin/bash
# round 5, at the head (index-build kernels, CGI row threshold): the whole GPU suite, the default bench line, then the profiles of
# the step, config 3 and config 4 again
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_final
mkdir -p $OUT
timeout 3000 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
timeout 1200 bash scripts/collect_profiles.sh r05 default 2>&1 | tail -1
timeout 1500 bash scripts/collect_profiles.sh r05 config4 2>&1 | tail -1
timeout 1800 bash scripts/collect_profiles.sh r05 config3 2>&1 | tail -1

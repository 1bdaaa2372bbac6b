#!/bin/bash
# round 5, the last GPU call: the whole GPU suite at the head (midpoints of the record searches), then four times config 3
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_final
mkdir -p $OUT gpurun_out/r05_scale
timeout 330 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_last.txt 2>&1; tail -2 $OUT/pytest_gpu_last.txt
timeout 230 python3 bench.py --strong --families 80 --members 50 --steps 1 --warmup 1 --no-fasta-leg > gpurun_out/r05_scale/strong_4000.json 2> gpurun_out/r05_scale/strong_4000.err
echo "exit $?"; tail -c 1500 gpurun_out/r05_scale/strong_4000.json

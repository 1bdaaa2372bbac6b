#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_zero_copy
mkdir -p $OUT
for v in 0 1 0 1; do FA_QUERY_ZERO_COPY=$v python3 scripts/time_boundary.py 2>/dev/null | head -1 | sed "s/^/FA_QUERY_ZERO_COPY=$v /" | tee -a $OUT/time_boundary.txt; done
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_binding.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt

#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_flush_overlap
mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sketch or index or minimizer or fasta or draft or protein or edge" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
timeout 900 python3 scripts/time_index.py 1000 5000000 3 > $OUT/time_index_1000.json 2> $OUT/err.txt
cat $OUT/time_index_1000.json
FA_TRACE=1 timeout 900 python3 scripts/time_index.py 1000 5000000 2 2>&1 >/dev/null | grep "fa trace" | tail -4

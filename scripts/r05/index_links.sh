#!/bin/bash
# round 5: the slide geometry of the index build (k_window_links in LDS tiles, k_link_duplicates without record gathers):
# definitions test, the parity suite around it, then the stage trace of the 1000-genome build
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_index_links
mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "links or index or end_to_end or draft or l2 or config" > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
FA_TRACE=1 timeout 900 python3 scripts/time_index.py 1000 5000000 2 > $OUT/time_index_1000.json 2> $OUT/trace_1000.txt
cat $OUT/time_index_1000.json; grep "fa trace" $OUT/trace_1000.txt | tail -8

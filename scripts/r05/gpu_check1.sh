#!/bin/bash
# round 5: the new host paths on the GPU box -- RCCL at world size 1, the files-to-table leg, the streamed ingest parity
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_check1
mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_rccl.py tests/test_gpu_parity.py -x -q -m gpu -k "rccl_at_world_size_one or through_rccl or strong_one_gpu or fasta or sharded" > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
python3 scripts/time_ingest.py 200 > $OUT/time_ingest.json 2>$OUT/time_ingest.err; cat $OUT/time_ingest.json
( cd scripts/ubench && g++ -O2 -std=c++17 -pthread -o ingest_host ingest_host.cpp && ./ingest_host 200 && ./ingest_host 1000 ) > $OUT/ingest_host.txt 2>&1; cat $OUT/ingest_host.txt

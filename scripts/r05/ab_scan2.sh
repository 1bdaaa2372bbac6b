#!/bin/bash
# round 5, A/B of the two-loci-per-lane slide (k_l2_scan2): ubench, then the resident bench step and 16 queries per launch
# with FA_SCAN2 = 0 / 1, then the parity suite with the paired form forced on; host ingest rates ride along.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_scan2
mkdir -p $OUT
( cd scripts/ubench && hipcc --offload-arch=gfx950 -O3 -std=c++17 -o slide_chain slide_chain.hip 2>/dev/null; ./slide_chain ) > $OUT/slide_chain.txt 2>&1
for nq in 1 16; do
  for v in 0 1; do
    FA_SCAN2=$v python3 scripts/time_pass.py 20 $nq 2>$OUT/time_pass_${nq}_${v}.err | tail -1 > $OUT/time_pass_${nq}_${v}.json
    cat $OUT/time_pass_${nq}_${v}.json
  done
done
( cd scripts/ubench && g++ -O2 -std=c++17 -pthread -o ingest_host ingest_host.cpp && ./ingest_host 200 && FA_HOST_THREADS=32 ./ingest_host 200 ) > $OUT/ingest_host.txt 2>&1
cat $OUT/ingest_host.txt
python3 scripts/time_ingest.py 200 > $OUT/time_ingest.json 2>$OUT/time_ingest.err; cat $OUT/time_ingest.json; tail -3 $OUT/time_ingest.err
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fasta" > $OUT/pytest_fasta.txt 2>&1; tail -5 $OUT/pytest_fasta.txt

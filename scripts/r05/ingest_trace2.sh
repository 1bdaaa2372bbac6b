#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_ingest_trace2
mkdir -p $OUT
for mode in read mmap; do
  FA_FASTA_IO=$mode FA_TRACE=1 python3 scripts/time_ingest.py 1000 > $OUT/time_ingest_1000_$mode.json 2> $OUT/trace_$mode.txt; echo "== $mode"; cat $OUT/time_ingest_1000_$mode.json; grep "fa trace" $OUT/trace_$mode.txt | grep -i "fasta\|add_fasta" | head -4
done
FA_FASTA_IO=mmap FA_FASTA_THREADS=all FA_TRACE=1 python3 scripts/time_ingest.py 1000 2> $OUT/trace_mmap_all.txt | tail -1; grep "fa trace" $OUT/trace_mmap_all.txt | grep -i "fasta\|add_fasta" | head -4
( cd scripts/ubench && g++ -O2 -std=c++17 -pthread -o ingest_host ingest_host.cpp && ./ingest_host 1000 && FA_FASTA_IO=mmap ./ingest_host 1000 && FA_FASTA_IO=mmap FA_FASTA_THREADS=all ./ingest_host 1000 ) 2>&1 | tail -3

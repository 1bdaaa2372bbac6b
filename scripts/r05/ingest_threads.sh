#!/bin/bash
set -u
OUT=gpurun_out/r05_ingest_threads.txt
{ for t in 12 24 32 48 64 96; do FA_FASTA_THREADS=$t python3 scripts/r05/ingest_threads.py 600 2>/dev/null | tail -1; done
  FA_FASTA_IO=mmap FA_FASTA_THREADS=24 python3 scripts/r05/ingest_threads.py 600 2>/dev/null | tail -1
  FA_FASTA_IO=mmap FA_FASTA_THREADS=64 python3 scripts/r05/ingest_threads.py 600 2>/dev/null | tail -1; } | tee $OUT

#!/bin/bash
# round 5: the all-vs-all at twice config 3 (2000 x 2000 genomes of 5 Mb, 8 x 10^8 index records) -- does the step keep its rate on
# an index twice the size?  (host memory checked first: the workload is 10 GB of Python bytes)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_scale
mkdir -p $OUT
free -g | tee $OUT/free.txt
avail=$(free -g | awk '/^Mem:/ {print $7}')
if [ "${avail:-0}" -lt 96 ]; then echo "less than 96 GB of host memory available: not run"; exit 0; fi
timeout 1500 python3 bench.py --strong --families 40 --members 50 --steps 1 --warmup 1 --no-fasta-leg > $OUT/strong_2000.json 2> $OUT/strong_2000.err
tail -c 2500 $OUT/strong_2000.json; tail -3 $OUT/strong_2000.err

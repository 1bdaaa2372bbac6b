#!/bin/bash
# round 5: four times config 3 (4000 x 4000 genomes of 5 Mb, 1.6 x 10^9 index records -- three quarters of the 2^31 records one index
# holds, ~100 GB of HBM): every genome must still hit itself at exactly 100.0
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_scale
mkdir -p $OUT
avail=$(free -g | awk '/^Mem:/ {print $7}')
if [ "${avail:-0}" -lt 192 ]; then echo "less than 192 GB of host memory available: not run"; exit 0; fi
timeout 2400 python3 bench.py --strong --families 80 --members 50 --steps 1 --warmup 1 --no-fasta-leg > $OUT/strong_4000.json 2> $OUT/strong_4000.err
tail -c 1800 $OUT/strong_4000.json; tail -3 $OUT/strong_4000.err
rocm-smi --showmeminfo vram 2>/dev/null | tail -4

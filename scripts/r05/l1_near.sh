#!/bin/bash
# round 5: k_l1 without the coordinate gathers of hopeless hits (no marking pass: partner read + ballot shifts) -- parity, then
# A/B of time on the config-3-shaped harness (32-bit coordinates) and on the full 1000 x 1000 (64-bit), plus K1 tile lengths
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_l1near2
mkdir -p $OUT
timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
for v in 0 1; do
  FA_L1_NEAR=$v python3 scripts/time_config3.py 4 50 5000000 3 2>/dev/null | tail -1 > $OUT/time_4x50_$v.json; cat $OUT/time_4x50_$v.json
  FA_L1_NEAR=$v python3 scripts/time_config3.py 10 50 5000000 2 2>/dev/null | tail -1 > $OUT/time_10x50_$v.json; cat $OUT/time_10x50_$v.json
done
for v in 0 1; do
  FA_L1_NEAR=$v python3 scripts/time_config3.py 20 50 5000000 2 2>/dev/null | tail -1 > $OUT/time_20x50_$v.json; cat $OUT/time_20x50_$v.json
done
for t in 1024 0; do
  for ng in 1 40; do
    FA_K1_TILE=$t python3 scripts/bench_k1.py $ng 20 2>/dev/null | tail -1 | sed "s/^/FA_K1_TILE=$t /" | tee -a $OUT/k1_tiles.txt
  done
done
FA_K1_TILE=1024 python3 scripts/bench_k1.py 40 10 14 5000 2>/dev/null | tail -1 | sed "s/^/FA_K1_TILE=1024 /" | tee -a $OUT/k1_tiles.txt
python3 scripts/bench_k1.py 40 10 14 5000 2>/dev/null | tail -1 | sed "s/^/FA_K1_TILE=0 /" | tee -a $OUT/k1_tiles.txt
FA_K1_TILE=1024 python3 scripts/time_index.py > $OUT/time_index_1024.txt 2>/dev/null; tail -2 $OUT/time_index_1024.txt
python3 scripts/time_index.py > $OUT/time_index_auto.txt 2>/dev/null; tail -2 $OUT/time_index_auto.txt

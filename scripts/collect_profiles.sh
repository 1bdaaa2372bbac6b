#!/bin/bash
# Profiles behind the bench line, collected on the GPU box (run from the repo root through gpurun):
#   bash scripts/collect_profiles.sh r02
# writes gpurun_out/<tag>_prof/{stats,fetch,write,sq_a,sq_b}/ and the summaries gpurun_out/<tag>_*.{json,csv}; copy the
# summaries into profiles/ afterwards.  Counters are collected in their own passes (MI355X_MICROARCH.md, HBM / rocprofv3
# PMC slots: FETCH_SIZE and WRITE_SIZE do not fit one pass) and never together with a trace domain.
set -u
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_prof
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --clients 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH --steps 20 --warmup 3 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $BENCH --steps 3 --warmup 1 > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $BENCH --steps 3 --warmup 1 > /dev/null 2> "$OUT/write.err"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sq_a" -- $BENCH --steps 3 --warmup 1 > /dev/null 2> "$OUT/sq_a.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq_b" -- $BENCH --steps 3 --warmup 1 > /dev/null 2> "$OUT/sq_b.err"
python3 scripts/summarize_profiles.py "$OUT" "gpurun_out/${TAG}"

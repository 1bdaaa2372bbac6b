#!/bin/bash
# Profiles behind the bench line, collected on the GPU box (run from the repo root through gpurun):
#   bash scripts/collect_profiles.sh r03            the timed step: 1 query x 100 references per launch
#   bash scripts/collect_profiles.sh r03 batch16    `saturated.batch16`: 16 queries per launch
#   bash scripts/collect_profiles.sh r03 config3    `saturated.config3`: 1000 x 1000 (bench.py --strong, two steps summed)
#   bash scripts/collect_profiles.sh r05 config4    `saturated.config4`: 500 x 500 draft assemblies (bench.py --leg config4)
#   bash scripts/collect_profiles.sh r05 config5:k21f3000   one cell of `config5_cells` (bench.py --leg config5:k21f3000)
#   bash scripts/collect_profiles.sh r06 genome_like        the genome-like leg (bench.py --leg genome_like)
# writes gpurun_out/<tag>[_<mode>]_prof/{stats,fetch,write,sq_a,sq_b}/ and the summaries gpurun_out/<tag>[_<mode>]_*.{json,csv};
# copy the summaries into profiles/ afterwards.  Counters are collected in their own passes (MI355X_MICROARCH.md, HBM /
# rocprofv3 PMC slots: FETCH_SIZE and WRITE_SIZE do not fit one pass) and never together with a trace domain; the program
# stands directly behind `--` (no env / bash -c hop: the profiler's preloaded library has initialised the GPU by then).
set -u
TAG=${1:-r03}
MODE=${2:-default}
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --clients 0 --no-saturated"
case "$MODE" in
  default) NAME=$TAG;           ARGS="$COMMON";            STATS="--steps 20 --warmup 3"; PMC="--steps 3 --warmup 1"; export FA_PROFILE_SUM_STEPS=0 ;;
  batch16) NAME=${TAG}_batch16; ARGS="$COMMON --no-boundary --batch 16"; STATS="--steps 10 --warmup 2"; PMC="--steps 2 --warmup 1"; export FA_PROFILE_SUM_STEPS=0 ;;
  config3) NAME=${TAG}_config3; ARGS="--strong --no-fasta-leg"; STATS="--steps 2 --warmup 1";  PMC="--steps 1 --warmup 1"; export FA_PROFILE_SUM_STEPS=2 ;;
  # (round 5) one leg of the line alone: BASELINE config 4, or one (k, fragment length) cell of config 5 -- `config5:k21f3000`;
  # a leg runs one warm-up step and N timed ones, every launch of every step is summed
  config4) NAME=${TAG}_config4; ARGS="--leg config4";      STATS="--saturated-steps 2";   PMC="--saturated-steps 1"; export FA_PROFILE_SUM_STEPS=2 ;;
  genome_like) NAME=${TAG}_genome_like; ARGS="--leg genome_like"; STATS="--saturated-steps 2"; PMC="--saturated-steps 1"; export FA_PROFILE_SUM_STEPS=2 ;;
  config5:*) CELL=${MODE#config5:}; NAME=${TAG}_config5_$(echo $CELL | sed 's/k\([0-9]*\)f\([0-9]*\)/k\1_f\2/'); ARGS="--leg $MODE"; STATS=""; PMC=""; export FA_PROFILE_SUM_STEPS=3 ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
OUT=gpurun_out/${NAME}_prof
mkdir -p "$OUT"
export FA_PROFILE_COMMAND="python3 bench.py $ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS $STATS --detail "$OUT/bench_detail.json" > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py $ARGS $PMC --detail /dev/null > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py $ARGS $PMC --detail /dev/null > /dev/null 2> "$OUT/write.err"
if [ "$MODE" = "default" ] || [ "$MODE" = "batch16" ]; then
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sq_a" -- python3 bench.py $ARGS $PMC --detail /dev/null > /dev/null 2> "$OUT/sq_a.err"
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq_b" -- python3 bench.py $ARGS $PMC --detail /dev/null > /dev/null 2> "$OUT/sq_b.err"
fi
cp "$OUT/bench_under_rocprof.json" "gpurun_out/${NAME}_bench_under_rocprof.json"
python3 scripts/summarize_profiles.py "$OUT" "gpurun_out/${NAME}"

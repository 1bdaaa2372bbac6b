#!/bin/bash
# round 6: the bench line at the head, the scale runs beyond config 3, and the fuzz campaign on the final library
O=gpurun_out; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/r06_bench_detail.json > $O/r06_bench_default.json 2> $O/r06_bench_default.err; echo "bench rc=$?"; tail -c 2900 $O/r06_bench_default.json; echo
timeout 900 python3 bench.py --strong --families 40 --members 50 --steps 1 --warmup 1 --no-fasta-leg --detail $O/r06_scale_2000x2000.json > /dev/null 2> $O/s2000.err
timeout 900 python3 bench.py --strong --families 80 --members 50 --steps 1 --warmup 1 --no-fasta-leg --detail $O/r06_scale_4000x4000.json > /dev/null 2> $O/s4000.err
for n in 2000 4000; do python3 -c "
import json; d=json.load(open('$O/r06_scale_${n}x${n}.json')); print('${n}x${n}', round(d['value']), {k: round(v,1) for k,v in d['phases_ms'].items()}, d['config']['table_sha256'], d.get('off_fast_path_share_rank0'))"; done
{
  echo "# differential fuzzing against the CPU oracle on the final library of round 6 (scripts/fuzz_parity.py <cases> <seed> <seconds>); every L2 mapping and every hit must match"
  run() { echo "## $1: python scripts/fuzz_parity.py $2 $3 $4 $5"; env $1 timeout $(( $4 + 120 )) python scripts/fuzz_parity.py $2 $3 $4 $5 2>&1 | tail -2; }
  run "FA_NONE=1" 20000 66001 560
  run "FA_L1_PREFILTER=1" 10000 66002 300
  run "FA_L1_PREFILTER=1 FA_L1_THIN_SMALL=0 FA_L1_THIN_MID=0" 5000 66003 150
  run "FA_EV_RANK=0" 5000 66004 150
  run "FA_PASS_FRAGMENTS=40" 5000 66005 150
  run "FA_NONE=1" 8000 66006 200 default-cell
} > $O/r06_fuzz_campaign.txt 2>&1
cat $O/r06_fuzz_campaign.txt

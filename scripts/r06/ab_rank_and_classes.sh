#!/bin/bash
# round 6 A/B on one MI355X: the rank structure of k_l2_events (FA_EV_RANK=0: bucket table + four-entry probe; 1: occupancy words) and the
# size classes of k_l1 (FA_L1_THIN_SMALL=0: the small class always has its own launch; 2: small fragments ride in the 512-thread form)
O=${1:-gpurun_out/r06f}; mkdir -p $O
show() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], round(d.get("value", d.get("pairs_per_s", 0))), {k: round(v, 4) for k, v in d["phases_ms"].items()}, d.get("config", {}).get("table_sha256", d.get("table_sha256")))
PY
}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
for rk in 0 1 0 1; do
  FA_EV_RANK=$rk timeout 300 python bench.py --no-cpu-baseline --no-saturated --no-genome-like --clients 0 --no-boundary --steps 50 --detail $O/step_rk$rk.json > /dev/null 2>> $O/err.log
  show "step rank=$rk" $O/step_rk$rk.json
done
for rk in 0 1; do
  FA_EV_RANK=$rk timeout 600 python bench.py --strong --steps 2 --warmup 1 --no-fasta-leg --detail $O/c3_rk$rk.json > /dev/null 2>> $O/err.log
  show "config3 rank=$rk" $O/c3_rk$rk.json
done
for ts in 0 2; do
  FA_L1_THIN_SMALL=$ts FA_DEBUG_L1=1 timeout 600 python bench.py --strong --steps 2 --warmup 1 --no-fasta-leg --detail $O/c3_ts$ts.json > /dev/null 2> $O/c3_ts$ts.err
  show "config3 thin_small=$ts" $O/c3_ts$ts.json; grep "k_l1 classes" $O/c3_ts$ts.err | tail -1
  FA_L1_THIN_SMALL=$ts FA_DEBUG_L1=1 timeout 300 python bench.py --leg genome_like > $O/gl_ts$ts.json 2> $O/gl_ts$ts.err
  show "genome_like thin_small=$ts" $O/gl_ts$ts.json; grep "k_l1 classes" $O/gl_ts$ts.err | tail -1
  FA_L1_THIN_SMALL=$ts timeout 300 python bench.py --leg config5:k14f1000 > $O/c5_ts$ts.json 2>> $O/err.log
  python3 -c "
import json; d=json.loads(open('$O/c5_ts$ts.json').read().strip().splitlines()[-1])['cells'][0]; print('config5 (14,1000) thin_small=$ts', round(d['value']), d['phases_ms'])"
done

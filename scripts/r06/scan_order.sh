#!/bin/bash
# round 6: k_l2_scan takes its loci sorted by stream length (FA_L2_SCAN_ORDER=0 / 1) -- parity forced on, then the A/B
O=${1:-gpurun_out/r06k}; mkdir -p $O
show() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
d = d["cells"][0] if "cells" in d else d
print(sys.argv[1], round(d.get("value", 0)), {k: round(v, 3) for k, v in d["phases_ms"].items()}, d.get("config", {}).get("table_sha256", d.get("table_sha256")))
PY
}
FA_L2_SCAN_ORDER=1 timeout 400 python scripts/fuzz_parity.py 6000 69001 200 2>&1 | tail -1
FA_L2_SCAN_ORDER=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
for so in 0 1 0 1; do
  FA_L2_SCAN_ORDER=$so timeout 600 python bench.py --strong --steps 3 --warmup 1 --no-fasta-leg --detail $O/c3_so$so.json > /dev/null 2>> $O/err.log; show "config3 scan_order=$so" $O/c3_so$so.json
done
for so in 0 1; do
  FA_L2_SCAN_ORDER=$so timeout 300 python bench.py --leg genome_like > $O/gl_so$so.json 2>> $O/err.log; show "genome_like scan_order=$so" $O/gl_so$so.json
  FA_L2_SCAN_ORDER=$so timeout 300 python bench.py --leg config4 > $O/c4_so$so.json 2>> $O/err.log; show "config4 scan_order=$so" $O/c4_so$so.json
  FA_L2_SCAN_ORDER=$so timeout 300 python bench.py --leg config5:k16f1000 > $O/c5_so$so.json 2>> $O/err.log; show "config5 (16,1000) scan_order=$so" $O/c5_so$so.json
  FA_L2_SCAN_ORDER=$so timeout 300 python bench.py --no-cpu-baseline --no-saturated --no-genome-like --clients 0 --no-boundary --steps 50 --batch 16 --detail $O/b16_so$so.json > /dev/null 2>> $O/err.log; show "batch16 scan_order=$so" $O/b16_so$so.json
  FA_L2_SCAN_ORDER=$so timeout 300 python bench.py --no-cpu-baseline --no-saturated --no-genome-like --clients 0 --no-boundary --steps 50 --detail $O/step_so$so.json > /dev/null 2>> $O/err.log; show "step scan_order=$so" $O/step_so$so.json
done

#!/bin/bash
# round 6: the pre-filter of k_l1's block sort (FA_L1_PREFILTER=1 forces it on any index) -- parity first, then what it costs / buys
O=${1:-gpurun_out/r06g}; mkdir -p $O
show() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], round(d.get("value", 0)), {k: round(v, 3) for k, v in d["phases_ms"].items()}, d.get("config", {}).get("table_sha256"), "off-fast share", d.get("off_fast_path_share_rank0"))
PY
}
export FA_L1_PREFILTER=1
timeout 600 python scripts/fuzz_parity.py 6000 60601 240 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
unset FA_L1_PREFILTER
for pf in 0 1; do
  FA_L1_PREFILTER=$pf FA_DEBUG_L1=1 timeout 600 python bench.py --strong --steps 2 --warmup 1 --no-fasta-leg --detail $O/c3_pf$pf.json > /dev/null 2> $O/c3_pf$pf.err
  show "config3 prefilter=$pf" $O/c3_pf$pf.json; grep "k_l1 classes" $O/c3_pf$pf.err | tail -1
done
for pf in 0 1; do
  FA_L1_PREFILTER=$pf timeout 900 python3 bench.py --strong --families 40 --members 50 --steps 1 --warmup 1 --no-fasta-leg --detail $O/s2000_pf$pf.json > /dev/null 2> $O/s2000_pf$pf.err
  show "2000x2000 prefilter=$pf" $O/s2000_pf$pf.json
done

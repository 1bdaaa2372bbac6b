#!/bin/bash
# round 6: where does the pre-filter of k_l1's block sort start to pay?  (FA_L1_PREFILTER=0 / 1 on small and mid-size indices; then 4000 x 4000 on its default)
O=${1:-gpurun_out/r06h}; mkdir -p $O
show() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
d = d["cells"][0] if "cells" in d else d
print(sys.argv[1], round(d.get("value", 0)), {k: round(v, 4) for k, v in d["phases_ms"].items()}, d.get("config", {}).get("table_sha256", d.get("table_sha256")), "off-fast", d.get("off_fast_path_share_rank0", d.get("off_fast_path_share")))
PY
}
for pf in 0 1 0 1; do
  FA_L1_PREFILTER=$pf timeout 300 python bench.py --no-cpu-baseline --no-saturated --no-genome-like --clients 0 --no-boundary --steps 50 --detail $O/step_pf$pf.json > /dev/null 2>> $O/err.log
  show "step prefilter=$pf" $O/step_pf$pf.json
done
for pf in 0 1; do
  FA_L1_PREFILTER=$pf timeout 300 python bench.py --leg config5:k16f3000 > $O/c5_pf$pf.json 2>> $O/err.log; show "config5 (16,3000) prefilter=$pf" $O/c5_pf$pf.json
  FA_L1_PREFILTER=$pf timeout 300 python bench.py --leg config5:k14f1000 > $O/c5a_pf$pf.json 2>> $O/err.log; show "config5 (14,1000) prefilter=$pf" $O/c5a_pf$pf.json
  FA_L1_PREFILTER=$pf FA_DEBUG_L1=1 timeout 300 python bench.py --leg genome_like > $O/gl_pf$pf.json 2> $O/gl_pf$pf.err; show "genome_like prefilter=$pf" $O/gl_pf$pf.json; grep "k_l1 classes" $O/gl_pf$pf.err | tail -1
  FA_L1_PREFILTER=$pf timeout 300 python bench.py --leg config4 > $O/c4_pf$pf.json 2>> $O/err.log; show "config4 prefilter=$pf" $O/c4_pf$pf.json
done
FA_DEBUG_L1=1 timeout 600 python bench.py --strong --steps 2 --warmup 1 --no-fasta-leg --detail $O/c3.json > /dev/null 2> $O/c3.err; show "config3 (pf auto)" $O/c3.json; grep "k_l1 classes" $O/c3.err | tail -1
free -g | head -2
FA_TRACE=2 timeout 700 python3 bench.py --strong --families 80 --members 50 --steps 1 --warmup 1 --no-fasta-leg --detail $O/s4000.json > /dev/null 2> $O/s4000.err; show "4000x4000 (pf auto)" $O/s4000.json; tail -3 $O/s4000.err | cut -c1-200

#!/bin/bash
# round 5: the row-forming tail of k_cgi_rows (last workgroup) on passes of thousands of pairs: config-4 leg + parity suite
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_cgi_tail
mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
python3 bench.py --leg config4 > $OUT/config4.json 2> $OUT/config4.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_cgi_tail/config4.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","ms_per_step","phases_ms","table_sha256","index_build_s") if k in d})
PY
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o c4 -- python3 $GRAFT_REPO_ROOT/bench.py --leg config4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200

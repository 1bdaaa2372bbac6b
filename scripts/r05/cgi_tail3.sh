#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_cgi_tail
mkdir -p $OUT
for e in 16384 2048; do
FA_ROWS_EMIT_MAX=$e python3 bench.py --leg config4 > $OUT/config4b_$e.json 2> $OUT/config4b_$e.err
python3 - $e <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r05_cgi_tail/config4b_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], {k: d[k] for k in ("value","ms_per_step","phases_ms","table_sha256") if k in d})
PY
done
python3 scripts/time_boundary.py > $OUT/boundary.txt 2>&1; tail -12 $OUT/boundary.txt
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest2.txt 2>&1
tail -3 $OUT/pytest2.txt

#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_flush_overlap
mkdir -p $OUT
for i in 1 2; do
timeout 900 python3 scripts/time_index.py 1000 5000000 3 2>/dev/null | python3 -c "
import json,sys
for r in json.loads(sys.stdin.read()): print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items()})"
done
FA_TRACE=1 timeout 900 python3 scripts/time_index.py 1000 5000000 2 2>&1 >/dev/null | grep "fa trace" | tail -4

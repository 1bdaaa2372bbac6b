#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_ingest_trace
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k fasta > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
FA_TRACE=1 python3 scripts/time_ingest.py 1000 > $OUT/time_ingest_1000.json 2> $OUT/trace.txt; cat $OUT/time_ingest_1000.json; grep "fa trace" $OUT/trace.txt | head -20
python3 bench.py --strong --steps 2 --warmup 1 > $OUT/bench_strong.json 2> $OUT/bench_strong.err; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_ingest_trace/bench_strong.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["phases_ms"], d["config"]["index_build_s"], d["config"]["host_pack_s"])
print(json.dumps(d["fasta_to_table"], indent=0))
PY

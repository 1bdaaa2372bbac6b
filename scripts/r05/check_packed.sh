#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_packed
mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fasta or add_drafts or device_pool" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python3 bench.py --strong --steps 2 --warmup 1 > $OUT/bench_strong.json 2> $OUT/bench_strong.err; tail -3 $OUT/bench_strong.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_packed/bench_strong.json").read().strip().splitlines()[-1])
f=d["fasta_to_table"]
print(round(d["value"]), "index_build_s", round(d["config"]["index_build_s"],3), "host_pack_s", round(d["config"]["host_pack_s"],3))
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in f.items() if k not in ("workload","read_once")})
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in f["read_once"].items()})
PY

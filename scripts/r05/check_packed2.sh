#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_packed
mkdir -p $OUT
for i in 1 2 3; do
python3 bench.py --strong --steps 1 --warmup 1 > $OUT/bench_strong.json 2> $OUT/bench_strong.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_packed/bench_strong.json").read().strip().splitlines()[-1])
f=d["fasta_to_table"]
print("two reads:", {k: round(f[k], 3) for k in ("wall_s","refs_wall_s","index_s","stream_s","stream_ingest_s","stream_map_s","stream_wait_s","device_pass_s","overlap")})
print("read once:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in f["read_once"].items() if k != "table_sha256"})
PY
done

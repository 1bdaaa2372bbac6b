#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_strong_twice
mkdir -p $OUT
for i in 1 2; do
python3 bench.py --strong --steps 2 --warmup 1 > $OUT/bench_strong_$i.json 2> $OUT/bench_strong_$i.err
python3 - $i <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r05_strong_twice/bench_strong_{sys.argv[1]}.json").read().strip().splitlines()[-1])
f=d["fasta_to_table"]
print(round(d["value"]), round(d["ms_per_step"],1), "index_build_s", round(d["config"]["index_build_s"],3), "host_pack_s", round(d["config"]["host_pack_s"],3), "| fasta wall", round(f["wall_s"],3), "refs", round(f["refs_wall_s"],3), "index", round(f["index_s"],3), "stream", round(f["stream_s"],3), "overlap", round(f["overlap"],3))
PY
done
FA_TRACE=1 python3 scripts/time_index.py 1000 5000000 2 2>&1 | grep "fa trace" | tail -4

#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_add_drafts
mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_rccl.py -x -q -m gpu -k "add_drafts or sharded or rccl_at_world_size_one or strong_one_gpu or two_ranks" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python3 bench.py --strong --steps 2 --warmup 1 > $OUT/bench_strong.json 2> $OUT/bench_strong.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_add_drafts/bench_strong.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], "index_build_s", d["config"]["index_build_s"], "host_pack_s", d["config"]["host_pack_s"], d["config"]["table_sha256"])
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["fasta_to_table"].items() if k != "workload"})
PY

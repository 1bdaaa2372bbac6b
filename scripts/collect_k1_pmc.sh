#!/bin/bash
# SQ counters of the sketch kernel alone (scripts/bench_k1.py 20 2), two rocprofv3 --pmc passes, summarised into
# gpurun_out/<tag>_k1_pmc.json:   bash scripts/collect_k1_pmc.sh r02
set -u
TAG=${1:-r03}
OUT=gpurun_out/${TAG}_k1_prof
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/a" -- python3 scripts/bench_k1.py 20 2 > "$OUT/a.log" 2> "$OUT/a.err"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM --output-format csv -d "$OUT/b" -- python3 scripts/bench_k1.py 20 2 > "$OUT/b.log" 2> "$OUT/b.err"
python3 - "$OUT" "gpurun_out/${TAG}_k1_pmc.json" <<'PY'
import csv, glob, json, os, sys
src, dst = sys.argv[1], sys.argv[2]
tot, launches, rows = {}, {}, []
for sub in ("a", "b"):
    for path in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "k_sketch_fast<16, 24>" not in r["Kernel_Name"]:
                continue
            rows.append(r)
grid = max(int(r["Grid_Size"]) for r in rows)                 # the launches over the whole batch, not the one-genome index's
for r in rows:
    if int(r["Grid_Size"]) != grid:
        continue
    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    launches.setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
per = {k: v / max(1, len(launches[k])) for k, v in tot.items()}
line = open(os.path.join(src, "a.log")).read().strip().splitlines()[-1]
bases = float(line.split("bases=")[1].split()[0])
d = {"command": "two rocprofv3 --pmc passes -- python3 scripts/bench_k1.py 20 2 (scripts/collect_k1_pmc.sh)", "bench_line_under_the_profiler": line,
     "launches_averaged": {k: len(v) for k, v in launches.items()}, "counters_per_launch": per, "derived": {}}
if "SQ_INSTS_VALU" in per:
    d["derived"]["valu_lane_instructions_per_base"] = per["SQ_INSTS_VALU"] * 64.0 / bases
ms = float(line.split("ms=")[1].split()[0])
d["derived"]["ms_per_launch_under_the_profiler"] = ms
d["derived"]["note"] = ("VALU wave-instructions per base = valu_lane_instructions_per_base / 64; scripts/valu_model.py prices them with the issue-slot "
                        "costs measured by scripts/ubench/valu_rates.hip (profiles/r03_valu_rates.txt): 2.33 cycles for plain add / logic / right "
                        "shift, 4.2 for everything else, multiplies included")
json.dump(d, open(dst, "w"), indent=1)
print(json.dumps(d["derived"]), line)
PY

"""Turn the rocprofv3 CSVs written by scripts/collect_profiles.sh into the summaries kept under profiles/.

  python3 scripts/summarize_profiles.py gpurun_out/r02_prof gpurun_out/r02
writes <prefix>_bench_kernel_stats.csv (the --stats table), <prefix>_traffic.json (FETCH_SIZE / WRITE_SIZE in KB of the
LAST launch of every kernel = the query step) and <prefix>_map_kernels_pmc.json (SQ counters of the last launch)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

src, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "").replace("fa::", "")
    cut = name.rfind("(")
    return name[:cut] if cut > 0 else name


SUM_STEPS = int(os.environ.get("FA_PROFILE_SUM_STEPS", "0"))     # > 0: sum ALL launches of a kernel (a run of that many identical steps)


def last_launch(pattern):
    """{kernel: {counter: value}} of the last dispatch of every kernel in a counter_collection.csv -- or, with
    FA_PROFILE_SUM_STEPS, of all its dispatches together (config 3: a step is ~36 passes, every one launches the kernels)"""
    out = {}
    for path in glob.glob(os.path.join(src, pattern, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(path)))
        last = {}
        for r in rows:
            last[r["Kernel_Name"]] = max(last.get(r["Kernel_Name"], 0), int(r["Dispatch_Id"]))
        for r in rows:
            if SUM_STEPS or int(r["Dispatch_Id"]) == last[r["Kernel_Name"]]:
                out.setdefault(short(r["Kernel_Name"]), {})[r["Counter_Name"]] = out.get(short(r["Kernel_Name"]), {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out


def head():
    try:
        return subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return os.environ.get("FA_HEAD", "unknown")


for path in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(path, prefix + "_bench_kernel_stats.csv")
fetch, write = last_launch("fetch"), last_launch("write")
kernels = {k: {"fetch_size_kb": fetch.get(k, {}).get("FETCH_SIZE", 0.0), "write_size_kb": write.get(k, {}).get("WRITE_SIZE", 0.0)}
           for k in sorted(set(fetch) | set(write))}
json.dump({"head": head(),
           "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- " + os.environ.get("FA_PROFILE_COMMAND", "python3 bench.py ..."),
           "steps_summed": max(SUM_STEPS, 1),
           "note": ("KB summed over ALL launches of each kernel in a run of `steps_summed` identical steps (bench.py divides by it)" if SUM_STEPS else
                    "KB per launch, last launch of each kernel (the query step; index-build kernels launch once)") +
                   ".  On gfx950 FETCH_SIZE reports half the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM): bench.py doubles it; "
                   "narrower access patterns are uncalibrated.",
           "kernels": kernels}, open(prefix + "_traffic.json", "w"), indent=1)
sq = last_launch("sq_a")
for k, v in last_launch("sq_b").items():
    sq.setdefault(k, {}).update(v)
keep = {k: v for k, v in sq.items() if k.startswith(("k_l2", "k_l1", "k_sketch_tiles", "k_lookup", "k_query_sketch", "k_query_fused", "k_sketch_fast", "k_cgi"))}
for k, v in keep.items():
    if v.get("SQ_LDS_IDX_ACTIVE"):
        v["lds_bank_conflict_frac"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
    if v.get("SQ_WAVE_CYCLES"):
        v["valu_active_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / v["SQ_WAVE_CYCLES"]
json.dump({"head": head(), "note": "SQ counters of the last launch of each mapping kernel (two rocprofv3 --pmc passes, summed over all SEs/XCDs)",
           "kernels": keep}, open(prefix + "_map_kernels_pmc.json", "w"), indent=1)
print("wrote", prefix + "_{bench_kernel_stats.csv,traffic.json,map_kernels_pmc.json}")

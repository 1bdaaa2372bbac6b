"""BASELINE configs 4 and 5 on one MI355X, checked through oracle-free properties.

  python scripts/run_config45.py 4 [families members]      500 draft assemblies (50 log-normal contigs, ~5 Mb), all-vs-all
  python scripts/run_config45.py 5 [families members len]  200 genomes, all-vs-all for k in {14,16,21} x frag in {1000,3000,5000}

Properties: every genome hits itself at (nearly) exactly 100.0 with (nearly) all of its fragments, hits stay inside the
family, hit membership is symmetric.  (k=21, frag=1000) is the degenerate cell: no window fits a fragment, nothing maps.
The same runs are driver-run tests: tests/test_gpu_fullsize.py::test_config4_fullsize / ::test_config5_fullsize."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastani_amd import workloads

which = int(sys.argv[1]) if len(sys.argv) > 1 else 4
strip = lambda r: {k: v for k, v in r.items() if not k.startswith("_")}  # noqa: E731

if which == 4:
    families = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    members = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    length = int(sys.argv[4]) if len(sys.argv) > 4 else 5_000_000
    t0 = time.time()
    genomes, fam = workloads.config4(families, members, length)
    t_gen = time.time() - t0
    out = strip(workloads.all_vs_all(genomes, fam, {}))
    print(json.dumps({"config": f"4: {len(genomes)} draft assemblies ({families} families x {members}), 50 contigs each, {length / 1e6:g} Mb, add_draft path",
                      "generate_s": t_gen, **out}))
else:
    families = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    members = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    length = int(sys.argv[4]) if len(sys.argv) > 4 else 5_000_000
    t0 = time.time()
    genomes, fam = workloads.config5(families, members, length)
    t_gen = time.time() - t0
    cells = []
    only = os.environ.get("CELLS")          # e.g. CELLS=21:1000,16:3000
    for k, frag in workloads.CONFIG5_CELLS:
        if only and f"{k}:{frag}" not in only.split(","):
            continue
        r = strip(workloads.all_vs_all(genomes, fam, {"k": k, "fragment_length": frag}))
        if r["window_size"] >= frag:      # degenerate cell: nothing can map
            r["self_hits_exact"] = r["rows"] == 0
        cells.append({"k": k, "fragment_length": frag, **r})
        print(json.dumps(cells[-1]), file=sys.stderr, flush=True)
    print(json.dumps({"config": f"5: {len(genomes)} genomes ({families} families x {members}) of {length / 1e6:g} Mb, all-vs-all per (k, fragment_length) cell",
                      "generate_s": t_gen, "cells": cells}))

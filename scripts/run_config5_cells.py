"""BASELINE config 5 (200 genomes all-vs-all in every (k, fragment_length) cell) with the sketch kernel timed per cell:

  python scripts/run_config5_cells.py > profiles/r03_config5_cells.json

Per cell: window size, K1 alone over the 200 resident genomes (Gbases/s; which kernel served it), the mapping time of the
40 000 pairs with its device phases, and the oracle-free properties of scripts/run_config45.py."""
import sys, os, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastani_amd import workloads
from pyfastani_amd._lib import lib, check

t0 = time.time()
genomes, fam = workloads.config5()
t_gen = time.time() - t0
cells = []
for k, frag in workloads.CONFIG5_CELLS:
    r = workloads.all_vs_all(genomes, fam, {"k": k, "fragment_length": frag})
    mapper = r.pop("_mapper"); r.pop("_rows")
    w = r["window_size"]
    cell = {"k": k, "fragment_length": frag, **r}
    if w < frag:
        batch = mapper.upload_genomes(genomes[:40])
        ms, bases, mins = C.c_float(0), C.c_uint64(0), C.c_uint64(0)
        check(lib.fa_bench_sketch_kernel(mapper._h, batch._h, 5, C.byref(ms), C.byref(bases), C.byref(mins)))
        cell["k1"] = {"kernel": "k_sketch_fast" if 4 <= w <= 64 else "k_sketch_tiles", "tables": k in (14, 16, 21), "genomes": 40,
                      "ms_per_launch": ms.value, "gbases_per_s": bases.value / ms.value / 1e6, "minimizers": mins.value,
                      "hbm_frac_algorithmic": (bases.value * 0.25 + mins.value * 12.0) / ms.value / 1e6 / 8000.0}
        del batch
    else:
        cell["self_hits_exact"] = cell["rows"] == 0       # degenerate cell: no window fits a fragment, nothing can map
    cells.append(cell)
    print(json.dumps(cell), file=sys.stderr, flush=True)
    del mapper
print(json.dumps({"command": "python scripts/run_config5_cells.py", "config": "5: 200 genomes (10 families x 20) of 5 Mb, all-vs-all per (k, fragment_length) cell",
                  "generate_s": t_gen, "cells": cells}, indent=1))

"""Phase timings of a config-3-shaped step in miniature -- the A/B harness for k_l1 and the saturated regime:
   python scripts/time_config3.py [families=4] [members=50] [length=5000000] [steps=3]
Every query fragment finds `members` relatives, as in BASELINE config 3 (20 x 50); the cost per fragment is the full
configuration's at 1/25 of its size.  Prints per-stage ms (device stamps), pairs/s and a digest of the hit table (equal
digests = equal rows: compare builds / environment settings with it)."""
import sys, os, json, time, hashlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pyfastani_amd as pf
from pyfastani_amd import workloads, sharding
from pyfastani_amd._lib import lib

fam = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mem = int(sys.argv[2]) if len(sys.argv) > 2 else 50
length = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
genomes, _ = workloads.config3(fam, mem, length)
n = len(genomes)
sk = pf.Sketch()
for i, c in enumerate(genomes):
    sk.add_draft(i, c)
mapper = sk.index()
batch = mapper.upload_genomes(genomes)
table = sharding.ResidentHitTable(list(range(n)), n * n, 1)
tables = table.step(batch)
ph, t = np.zeros(24), 0.0
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tables = table.step(batch)
    torch.cuda.synchronize()
    t += time.perf_counter() - t0
    ms = (C.c_float * 24)(); lib.fa_mapper_last_timings(mapper._h, ms, 24)
    ph += np.array(list(ms))
ph /= steps
rows = sharding.ResidentHitTable.rows_of(tables)
order = np.lexsort((rows["ref_genome_id"], rows["query_id"]))
print(json.dumps({"config": f"{n} x {n} ({fam} x {mem}), {length / 1e6:g} Mb", "env": {k: v for k, v in os.environ.items() if k.startswith("FA_")},
                  "ms_per_step": t / steps * 1e3, "pairs_per_s": n * n * steps / t, "sketch_ms": ph[0], "lookup_l1_ms": ph[1], "l2_ms": ph[2], "cgi_ms": ph[3],
                  "total_ms": ph[4], "repeats": ph[9], "l1_block_sorted": ph[20], "l1_merged": ph[21], "rows": int(len(rows)), "sha": hashlib.sha256(np.ascontiguousarray(rows[order]).tobytes()).hexdigest()[:16]}))

"""End-to-end latency of the reference-shaped call Mapper.query_draft (host buffer in, Hit list out) on BASELINE config 2:
host packing + PCIe upload + the device pass + row download + Hit construction."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import synthetic as syn

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
query, refs, names = syn.config2(n_related=int(round(n_refs * 0.6)), n_unrelated=n_refs - int(round(n_refs * 0.6)))
sk = pf.Sketch()
for name, r in zip(names, refs):
    sk.add_genome(name, r)
mapper = sk.index()
qb = bytes(query)
for _ in range(3):
    hits = mapper.query_draft([qb])
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    hits = mapper.query_draft([qb])
    ts.append(time.perf_counter() - t0)
batch = mapper.upload_genomes([[qb]])
for _ in range(3):
    batch.query_rows(0, 1)
tr = []
for _ in range(20):
    t0 = time.perf_counter()
    rows = batch.query_rows(0, 1)
    tr.append(time.perf_counter() - t0)
print(json.dumps({"refs": n_refs, "hits": len(hits), "query_draft_ms_median": 1e3 * float(np.median(ts)), "query_draft_ms_min": 1e3 * min(ts),
                  "pairs_per_s_query_draft": n_refs / float(np.median(ts)),
                  "resident_query_rows_ms_median": 1e3 * float(np.median(tr)), "pairs_per_s_resident": n_refs / float(np.median(tr))}))

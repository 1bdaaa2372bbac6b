"""Phase timings of the resident bench step (config 2) -- a quick A/B harness for kernel experiments:
   python scripts/time_pass.py [steps]   (environment variables select the variant)"""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import workloads
from pyfastani_amd._lib import lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # query genomes per launch sequence (1 = the bench step, 16 = `saturated.batch16`)
anc, names, refs = workloads.config2_references(100, 5_000_000)
sk = pf.Sketch()
for n, c in zip(names, refs):
    sk.add_draft(n, c)
mapper = sk.index()
batch = mapper.upload_genomes(workloads.config2_query(anc, 0, 1) if nq == 1 else [workloads.config2_query(anc, 100 + i, 1)[0] for i in range(nq)])
for _ in range(3):
    rows = batch.query_rows(0, nq)
ph = np.zeros(24)
for _ in range(steps):
    rows = batch.query_rows(0, nq)
    ms = (C.c_float * 24)(); lib.fa_mapper_last_timings(mapper._h, ms, 24)
    ph += np.array(list(ms)[:24])
ph /= steps
import hashlib
digest = hashlib.sha256(np.ascontiguousarray(rows).tobytes()).hexdigest()[:16]
print(json.dumps({"queries": nq, "env": {k: v for k, v in os.environ.items() if k.startswith("FA_")}, "rows": int(len(rows)),
                  "sketch_ms": ph[0], "lookup_l1_ms": ph[1], "l2_ms": ph[2], "cgi_ms": ph[3], "total_ms": ph[4], "events": ph[7], "smax": ph[14], "loci": ph[6], "rows_sha": digest}))

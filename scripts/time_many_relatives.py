"""The "many strains of one species" regime: R near-identical references, every query fragment gathers ~200 x R seed
hits and R candidate loci.  Prints the per-query time and the phase split."""
import sys, os, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import synthetic as syn
from pyfastani_amd._lib import lib

R = int(sys.argv[1]) if len(sys.argv) > 1 else 300
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
g = syn.rng(77)
anc = syn.random_codes(g, length)
sk = pf.Sketch()
for i in range(R):
    sk.add_genome(i, syn.to_ascii(syn.mutate_codes(g, anc, 0.01)))
t0 = time.time(); mapper = sk.index(); t_index = time.time() - t0
q = syn.to_ascii(syn.mutate_codes(g, anc, 0.01))
batch = mapper.upload_genomes([[q]])
batch.query_rows(0, 1)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); rows = batch.query_rows(0, 1); ts.append(time.perf_counter() - t0)
ms = (C.c_float * 16)(); lib.fa_mapper_last_timings(mapper._h, ms, 16)
print(json.dumps({"references": R, "length": length, "threshold": mapper.occurences_threshold, "index_s": t_index, "rows": int(len(rows)),
                  "query_ms": 1e3 * min(ts), "pairs_per_s": R / min(ts),
                  "phases_ms": dict(zip(["sketch", "lookup_l1", "l2", "cgi", "total"], [round(float(x), 3) for x in list(ms)[:5]])),
                  "loci": float(ms[6]), "events": float(ms[7])}))

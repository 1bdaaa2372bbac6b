"""Device-memory stability: repeated queries on one mapper, and repeated construction / destruction of sketches,
mappers and resident batches, must not grow the device allocation."""
import sys, os, gc, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyfastani_amd as pf
from pyfastani_amd import synthetic as syn

def used():
    torch.cuda.synchronize()
    pf.device_trim()                      # (freed blocks are kept for reuse: what the pool holds is not a leak)
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20

g = syn.rng(5)
anc = syn.random_codes(g, 1_000_000)
refs = [syn.to_ascii(syn.mutate_codes(g, anc, d)) for d in (0.0, 0.03, 0.08)] + [syn.to_ascii(syn.random_codes(g, 1_000_000))]
queries = [syn.to_ascii(syn.mutate_codes(g, anc, d)) for d in (0.02, 0.1)]
out = {}
sk = pf.Sketch()
for i, r in enumerate(refs): sk.add_genome(i, r)
m = sk.index()
for q in queries: m.query_genome(q)
base = used()
for _ in range(300):
    for q in queries: m.query_genome(q)
out["after_600_queries_mb"] = used() - base
b = m.upload_genomes([[q] for q in queries]); b.query(); del b; gc.collect()
base2 = used()
for _ in range(50):
    b = m.upload_genomes([[q] for q in queries]); b.query(); del b
gc.collect()
out["after_50_batches_mb"] = used() - base2
del m, sk; gc.collect()
def cycle():
    sk = pf.Sketch()
    for i, r in enumerate(refs): sk.add_genome(i, r)
    m = sk.index(); m.query_genome(queries[0]); del m, sk
for _ in range(4): cycle()              # the runtime's own pools settle during the first life cycles
gc.collect()
base3 = used()
for _ in range(20): cycle()
gc.collect()
out["after_20_mappers_mb"] = used() - base3
print(json.dumps(out))

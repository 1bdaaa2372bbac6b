"""Round 5 probe: whole passes of an all-vs-all in flight on two workspaces (two host threads, alternate chunks) against
one after another.  python3 scripts/r05/two_streams.py [families] [members] [threads...]"""
import sys, os, time, json, threading, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import workloads

fams = int(sys.argv[1]) if len(sys.argv) > 1 else 8
members = int(sys.argv[2]) if len(sys.argv) > 2 else 50
modes = [int(x) for x in sys.argv[3:]] or [1, 2, 3]
genomes, fam = workloads.families(2000, fams, members, 5_000_000)
n = len(genomes)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sk = pf.Sketch()
    sk.add_drafts(list(range(n)), genomes)
    mapper = sk.index()
    batch = mapper.upload_genomes(genomes)
chunk = 29
starts = list(range(0, n, chunk))
out = {"genomes": n, "chunk": chunk}
ref = None
for threads in modes:
    def run(which, res):
        res[which] = [batch.query_rows(s, min(chunk, n - s)) for s in starts[which::threads]]
    for rep in range(3):
        res = {}
        ts = [threading.Thread(target=run, args=(w, res)) for w in range(threads)]
        t0 = time.perf_counter()
        [t.start() for t in ts]; [t.join() for t in ts]
        dt = time.perf_counter() - t0
    rows = [None] * len(starts)
    for w in range(threads):
        for j, r in enumerate(res[w]):
            rows[w + j * threads] = r
    rows = np.concatenate(rows)
    if ref is None:
        ref = rows
    out[f"threads_{threads}"] = {"s": dt, "pairs_per_s": n * n / dt, "same_rows": bool(np.array_equal(rows, ref))}
print(json.dumps(out))

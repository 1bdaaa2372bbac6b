"""Throughput of K host threads querying ONE mapper concurrently (each call takes its own workspace and stream)."""
import sys, os, time, json, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyfastani_amd as pf
from pyfastani_amd import synthetic as syn

query, refs, names = syn.config2()
sk = pf.Sketch()
for n_, r in zip(names, refs):
    sk.add_genome(n_, r)
mapper = sk.index()
batch = mapper.upload_genomes([[bytes(query)]])
N = 200
out = []
for K in (1, 2, 3, 4):
    rows = [torch.zeros((100, 5), dtype=torch.int32, device="cuda") for _ in range(K)]
    def run(i, n=N):
        for _ in range(n):
            batch.query_rows_device(0, 1, rows[i].data_ptr(), rows[i].shape[0])
    for i in range(K): run(i, 5)
    ths = [threading.Thread(target=run, args=(i,)) for i in range(K)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    out.append({"threads": K, "ms_per_query": dt / (N * K) * 1e3, "pairs_per_s": 100 * N * K / dt})
print(json.dumps(out))

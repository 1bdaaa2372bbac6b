"""Time sketching + index construction of N synthetic 5 Mb genomes (FA_TRACE=1 prints the stage split on stderr)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyfastani_amd as pf
from pyfastani_amd import synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
length = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = syn.rng(7)
genomes = [syn.to_ascii(syn.random_codes(g, length)) for _ in range(n)]
out = []
for rep in range(reps):
    sk = pf.Sketch()
    t0 = time.time()
    for i, s in enumerate(genomes):
        sk.add_genome(i, s)
    t1 = time.time()
    nmin = len(sk.minimizers)
    t2 = time.time()
    m = sk.index()
    t3 = time.time()
    out.append({"rep": rep, "pack_s": t1 - t0, "sketch_s": t2 - t1, "index_s": t3 - t2, "minimizers": nmin,
                "gbases_per_s_sketch_index": n * length / (t3 - t1) / 1e9})
    del m, sk
print(json.dumps(out))

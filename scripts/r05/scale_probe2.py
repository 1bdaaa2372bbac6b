"""Round 5: an index of 1.6 x 10^9 records built from 400 generated genomes added ten times each (generation is the slow part of
the real thing) -- which stage of build_index does not come back?  FA_TRACE=2 prints every stage as it ends."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
T0 = time.time()
def say(*a):
    print(f"[{time.time() - T0:8.2f}s]", *a, flush=True)
import pyfastani_amd as pf
from pyfastani_amd import workloads
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 10
genomes, fam = workloads.families(2000, 8, 50, 5_000_000)
say("generated", len(genomes))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sk = pf.Sketch()
    for c in range(copies):
        sk.add_drafts([f"c{c}_{i}" for i in range(len(genomes))], genomes)
    say("packed on the host")
    mapper = sk.index()
    say("indexed:", len(mapper.minimizers), "records,", len(mapper.lookup_index), "distinct, threshold", mapper.occurences_threshold)
import numpy as np
batch = mapper.upload_genomes(genomes[:58])
say("uploaded 58 query genomes")
for c in range(2):
    rows = batch.query_rows(29 * c, 29)
    own = rows[rows["query_id"] == (rows["ref_genome_id"] % len(genomes))]
    say("chunk", c, "rows", len(rows), "rows against the copies of the query itself", len(own), "all exactly 100:", bool(np.all(own["identity"] == 100.0)))
say("done")

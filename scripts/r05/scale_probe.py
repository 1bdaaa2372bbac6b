"""Round 5: where does a 4000-genome index (1.6 x 10^9 records) spend its time?  Progress lines with wall-clock stamps, flushed, so
that a run cut off by `timeout` still says how far it got.  python3 scripts/r05/scale_probe.py <families> <members> [chunks]"""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
T0 = time.time()
def say(*a):
    print(f"[{time.time() - T0:8.2f}s]", *a, flush=True)
import pyfastani_amd as pf
from pyfastani_amd import workloads
fams, members = int(sys.argv[1]), int(sys.argv[2])
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 3
say("generating", fams * members, "genomes")
genomes, fam = workloads.families(2000, fams, members, 5_000_000)
n = len(genomes)
say("generated")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    sk = pf.Sketch()
    sk.add_drafts(list(range(n)), genomes)
    say("packed on the host")
    mapper = sk.index()
    say("indexed:", len(mapper.minimizers), "records,", len(mapper.lookup_index), "distinct, threshold", mapper.occurences_threshold)
    first = genomes[: 29 * chunks]
    batch = mapper.upload_genomes(first)
    say("uploaded", len(first), "query genomes")
    for c in range(chunks):
        rows = batch.query_rows(29 * c, 29)
        own = rows[rows["query_id"] == rows["ref_genome_id"]]
        say("chunk", c, "rows", len(rows), "self rows", len(own), "all exactly 100:", bool(np.all(own["identity"] == 100.0)))
say("done")

import sys, os, ctypes as C
sys.path.insert(0, ".")
import pyfastani_amd as pf
from pyfastani_amd import workloads
from pyfastani_amd._lib import lib, check
def probe(tag):
    out = []
    for kb in (16, 22, 24):
        n = C.c_int(0); check(lib.fa_debug_probe_occupancy(kb * 1024, C.byref(n))); out.append((kb, n.value / 256))
    print(tag, out, flush=True)
probe("fresh process")
anc, names, refs = workloads.config2_references(20, 5_000_000)
sk = pf.Sketch()
for n, c in zip(names, refs): sk.add_draft(n, c)
probe("after packing")
mapper = sk.index()
probe("after index")
batch = mapper.upload_genomes(workloads.config2_query(anc, 0, 1))
probe("after upload")
rows = batch.query_rows(0, 1)
probe("after one pass")

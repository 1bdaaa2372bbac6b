import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyfastani_amd as pf
from pyfastani_amd import workloads, synthetic as syn
anc, names, refs = workloads.config2_references(100, 5_000_000)
sk = pf.Sketch()
for n, c in zip(names, refs): sk.add_draft(n, c)
mapper = sk.index()
q = workloads.config2_query(anc, 0)[0]
codes = np.frombuffer(bytes(q[0]), dtype=np.uint8)
comp = np.zeros(256, np.uint8); comp[ord('A')] = ord('T'); comp[ord('T')] = ord('A'); comp[ord('C')] = ord('G'); comp[ord('G')] = ord('C')
rc = comp[codes[::-1]].tobytes()
b = mapper.upload_genomes([q, [rc]])
rows = b.query_rows(0, 2)
a = rows[rows["query_id"] == 0]; r = rows[rows["query_id"] == 1]
print(len(a), len(r))
rb = {int(x["ref_genome_id"]): x for x in r}
out = []
for x in a:
    o = rb.get(int(x["ref_genome_id"]))
    out.append((int(x["count_seq"]), None if o is None else int(o["count_seq"]), float(x["identity"]), None if o is None else float(o["identity"])))
out.sort(key=lambda t: t[0])
for t in out: print(t, "" if t[1] is None else (t[0]-t[1], round(t[2]-t[3], 4)))
only_r = set(int(x["ref_genome_id"]) for x in r) - set(int(x["ref_genome_id"]) for x in a)
print("only in rc:", [(int(rb[i]["count_seq"]), float(rb[i]["identity"])) for i in only_r])

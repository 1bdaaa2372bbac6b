import sys, warnings
sys.path.insert(0, '/root/repo')
import pyfastani_amd as pf
from pyfastani_amd import synthetic as syn
from oracle.oracle import OracleSketch
g = syn.rng(96)
anc = syn.random_codes(g, 120_000)
sk, osk = pf.Sketch(), OracleSketch()
for i, d in enumerate((0.01, 0.05, 0.1)):
    r = syn.to_ascii(syn.mutate_codes(g, anc, d)); sk.add_genome(i, r); osk.add_genome(i, r)
m = sk.index(); osk.index()
q = syn.to_ascii(syn.mutate_codes(g, anc, 0.03))
got = [(h.name, h.identity, h.matches, h.fragments) for h in m.query_genome(q)]
assert got == osk.query_draft([q]) and len(got) == 3, got
print("OK")

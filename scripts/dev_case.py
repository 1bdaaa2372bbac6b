import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, warnings
import pyfastani_amd as pf
from pyfastani_amd import _lib, synthetic as syn
from pyfastani_amd._lib import lib, check
from oracle.oracle import OracleSketch
g = syn.rng(55)
ref = syn.random_codes(g, 60_000)
refs = [[syn.to_ascii(ref)], [syn.to_ascii(syn.mutate_codes(g, ref, 0.05))]]
q = bytearray(b"N" * 12_000)
for f, (a, n) in enumerate([(20_000, 150), (31_000, 90), (40_500, 60), (50_000, 40)]):
    q[f * 3000 + 1400: f * 3000 + 1400 + n] = bytes(syn.to_ascii(ref[a: a + n]))
params = {"minimum_fraction": 0.0, "percentage_identity": 70.0}
sk, osk = pf.Sketch(**params), OracleSketch(**params)
print("window", sk.window_size, osk.window_size)
for i, r in enumerate(refs):
    sk.add_draft(f"r{i}", r); osk.add_draft(f"r{i}", r)
m = sk.index(); osk.index()
hits = m.query_draft([bytes(q)])
oh, det = osk.query_draft([bytes(q)], details=True)
print("gpu hits", hits); print("oracle", oh)
om = det["mappings"]; print("oracle mappings", list(zip(om["qseq"], om["rseq"], om["rstart"], om["sketch"], om["shared"])))
cap = 1 << 16
arr = [np.empty(cap, np.int32) for _ in range(4)]; n = C.c_int64(0)
check(lib.fa_mapper_debug_l1(m._h, *[a.ctypes.data for a in arr], cap, C.byref(n)))
print("gpu loci", n.value, [tuple(a[i] for a in arr) for i in range(min(n.value, 12))])
for f in range(4):
    sz = C.c_int32(0); buf = np.empty(4096, np.uint32)
    check(lib.fa_mapper_debug_query_sketch(m._h, f, buf.ctypes.data, 4096, C.byref(sz)))
    ss, mh, loci = osk.l1_fragment(bytes(q[f*3000:(f+1)*3000]))
    print("frag", f, "gpu s", sz.value, "oracle s", ss, "minhits", mh, "oracle loci", loci[:6])
buf = (_lib.Mapping * cap)(); n = C.c_int64(0)
check(lib.fa_mapper_debug_mappings(m._h, buf, cap, C.byref(n)))
print("gpu mappings", [(buf[i].query_seq_id, buf[i].ref_seq_id, buf[i].ref_start_pos, buf[i].sketch_size, buf[i].conserved) for i in range(n.value)])

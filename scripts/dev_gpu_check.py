"""Developer script: stage-by-stage comparison of the HIP path against the CPU oracle on a GPU box."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import _lib, synthetic as syn
from pyfastani_amd._lib import lib, check
from oracle.oracle import OracleSketch

def params_of(sk):
    return sk._param

def gpu_sketch_sequence(sk, seq):
    b = bytes(seq) if not isinstance(seq, (bytes, str)) else (seq.encode() if isinstance(seq, str) else seq)
    cap = max(len(b), 1)
    h = np.empty(cap, np.uint32); w = np.empty(cap, np.int32); n = C.c_int64(0)
    check(lib.fa_debug_sketch_sequence(C.byref(sk._param), b, len(b), 1, h.ctypes.data, w.ctypes.data, cap, C.byref(n)))
    return h[:n.value], w[:n.value]

def cmp_stream(name, a, b):
    (ha, wa), (hb, wb) = a, b
    ok = len(ha) == len(hb) and np.array_equal(ha, hb) and np.array_equal(wa, wb)
    print(f"[{'OK' if ok else 'FAIL'}] {name}: gpu {len(ha)} vs oracle {len(hb)}")
    if not ok:
        n = min(len(ha), len(hb))
        bad = np.nonzero((ha[:n] != hb[:n]) | (wa[:n] != wb[:n]))[0]
        if len(bad):
            i = bad[0]; print("   first diff at", i, "gpu", ha[max(0,i-2):i+3], wa[max(0,i-2):i+3], "oracle", hb[max(0,i-2):i+3], wb[max(0,i-2):i+3])
    return ok

g = syn.rng(7)
allok = True
# ---- K1 stream parity ----
for (k, frag) in [(16, 3000), (14, 1000), (21, 3000), (5, 3000), (33, 3000), (12, 500), (16, 5000)]:
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sk = pf.Sketch(k=k, fragment_length=frag)
    osk = OracleSketch(k=k, fragment_length=frag)
    if osk.window_size < 0:
        print('skip degenerate', k, frag); continue
    assert sk.window_size == osk.window_size, (sk.window_size, osk.window_size)
    cases = {
        "random10k": syn.to_ascii(syn.random_codes(g, 10000)),
        "random3000": syn.to_ascii(syn.random_codes(g, 3000)),
        "ATGC*1000": b"ATGC" * 1000,
        "polyA": b"A" * 5000,
        "ATrepeat": b"AT" * 3000,
        "withN": bytes(syn.to_ascii(syn.random_codes(g, 3000))) + b"N" * 100 + bytes(syn.to_ascii(syn.random_codes(g, 4000))) + b"nnRYKM" + bytes(syn.to_ascii(syn.random_codes(g, 500))),
        "lower": bytes(syn.to_ascii(syn.random_codes(g, 5000))).lower(),
        "short": b"ACGTACGTACGTACGTACGTACGTACGTAC",
        "w+k-1": bytes(syn.to_ascii(syn.random_codes(g, sk.window_size + k - 1))),
        "w+k": bytes(syn.to_ascii(syn.random_codes(g, sk.window_size + k))),
    }
    for name, seq in cases.items():
        allok &= cmp_stream(f"k={k} w={sk.window_size} {name}", gpu_sketch_sequence(sk, seq), osk.sketch_sequence(seq))

# ---- reference sketch of several contigs + index ----
anc, members, ds = syn.family(11, 4, 200000)
sk = pf.Sketch(); osk = OracleSketch()
for i, m in enumerate(members):
    contigs = syn.split_contigs(g, m, 3) + [b"ACGT"]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sk.add_draft(f"g{i}", contigs)
    osk.add_draft(f"g{i}", contigs)
h, s, w = sk.minimizers._arrays()
oh, os_, ow = osk.minimizers()
ok = np.array_equal(h, oh) and np.array_equal(s, os_) and np.array_equal(w, ow)
print(f"[{'OK' if ok else 'FAIL'}] reference minimizers {len(h)} vs {len(oh)}"); allok &= ok
mapper = sk.index(); osk.index()
ok = len(mapper.lookup_index) == osk.index_size and mapper.occurences_threshold == osk.freq_threshold
print(f"[{'OK' if ok else 'FAIL'}] index size {len(mapper.lookup_index)} vs {osk.index_size}; thr {mapper.occurences_threshold} vs {osk.freq_threshold}"); allok &= ok

# ---- query ----
q = syn.to_ascii(syn.mutate_codes(g, anc, 0.04))
t = time.time(); hits = mapper.query_genome(q); tg = time.time() - t
ohits, det = osk.query_draft([q], details=True)
print("gpu   ", [(h.name, h.identity, h.matches, h.fragments) for h in hits], f"{tg*1e3:.1f} ms")
print("oracle", ohits)
ok = [(h.name, h.identity, h.matches, h.fragments) for h in hits] == ohits
print(f"[{'OK' if ok else 'FAIL'}] hits"); allok &= ok
# mappings
cap = 1 << 20
buf = (_lib.Mapping * cap)(); n = C.c_int64(0)
check(lib.fa_mapper_debug_mappings(mapper._h, buf, cap, C.byref(n)))
gm = sorted((buf[i].query_seq_id, buf[i].ref_seq_id, buf[i].ref_start_pos, buf[i].sketch_size, buf[i].conserved) for i in range(n.value))
om = det["mappings"]
omm = sorted(zip(om["qseq"].tolist(), om["rseq"].tolist(), om["rstart"].tolist(), om["sketch"].tolist(), om["shared"].tolist()))
ok = gm == omm
print(f"[{'OK' if ok else 'FAIL'}] L2 mappings gpu {len(gm)} oracle {len(omm)}"); allok &= ok
if not ok:
    sg, so = set(gm), set(omm)
    print("  only gpu", sorted(sg - so)[:5]); print("  only oracle", sorted(so - sg)[:5])
ms = (C.c_float * 8)(); lib.fa_mapper_last_timings(mapper._h, ms, 8); print("timings ms", list(ms)[:5])

# ---- protein golden ----
def fasta(p):
    recs = []; cur = None
    for l in open(p):
        l = l.strip()
        if l.startswith(">"): cur = []; recs.append(cur)
        elif l: cur.append(l)
    return ["".join(r) for r in recs]
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
b1 = fasta(f"{G}/BGC0001425.faa"); b3 = fasta(f"{G}/BGC0001428.faa")
sk = pf.Sketch(protein=True, fragment_length=100)
sk.add_draft("BGC0001425", b1); sk.add_draft("BGC0001427", b1)
print("protein minimizers", len(sk.minimizers))
mp = sk.index()
hits = mp.query_draft(b3)
print(hits)
ok = [(h.name, h.matches, h.fragments) for h in hits] == [("BGC0001425", 130, 176), ("BGC0001427", 130, 176)]
print(f"[{'OK' if ok else 'FAIL'}] protein golden"); allok &= ok
print("ALL OK" if allok else "SOME FAILED")

"""Differential fuzzing of the oversized-fragment path (k_l1_big and its fallback to the HBM sort of k_l1) against the
CPU oracle: many near-identical references in random layouts -- contigs per strain, tandem copies per contig, strains
that miss a part of the genome, N runs -- and random parameters.  Every L2 mapping and every hit must match.
Usage: python scripts/fuzz_chunked_l1.py [cases] [seed]"""
import sys, os, ctypes as C, warnings, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import _lib, synthetic as syn
from pyfastani_amd._lib import lib, check
from oracle.oracle import OracleSketch


def mappings(mapper):
    cap = 1 << 22
    buf = (_lib.Mapping * cap)(); n = C.c_int64(0)
    check(lib.fa_mapper_debug_mappings(mapper._h, buf, cap, C.byref(n)))
    assert n.value <= cap
    return sorted((buf[i].query_seq_id, buf[i].ref_seq_id, buf[i].ref_start_pos, buf[i].sketch_size, buf[i].conserved) for i in range(n.value))


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = syn.rng(seed)
bad = 0
t0 = time.time()
for case in range(cases):
    k, frag = [(16, 3000), (16, 3000), (14, 1000), (16, 5000), (21, 3000), (12, 1500)][int(g.integers(0, 6))]
    params = dict(k=k, fragment_length=frag, minimum_fraction=float(g.choice([0.0, 0.2])))
    length = int(g.integers(3 * frag, 8 * frag))
    base = syn.random_codes(g, length)
    strains = int(g.integers(60, 420))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sk, osk = pf.Sketch(**params), OracleSketch(**params)
        style = int(g.integers(0, 4))
        for i in range(strains):
            d = float(g.choice([0.0, 0.005, 0.01, 0.03]))
            copies = 1
            if style == 1 and g.random() < 0.2: copies = int(g.integers(2, 9))            # a few heavy contigs
            if style == 2 and i == strains // 2: copies = int(g.integers(40, 140))        # one uncuttable contig
            codes = np.concatenate([syn.mutate_codes(g, base, d) if d else base for _ in range(copies)])
            if g.random() < 0.2:                                                            # strain misses a part of the genome
                a = int(g.integers(0, len(codes) - frag)); codes = np.concatenate([codes[:a], codes[a + int(g.integers(1, frag)):]])
            contigs = syn.split_contigs(g, syn.to_ascii(codes), int(g.integers(1, 5)) if style != 3 else 1)
            contigs = [bytes(c) for c in contigs]
            if g.random() < 0.1:
                b = bytearray(contigs[0]); p = int(g.integers(0, max(1, len(b) - 100))); b[p:p + 60] = b"N" * len(b[p:p + 60]); contigs[0] = bytes(b)
            sk.add_draft(i, contigs); osk.add_draft(i, contigs)
        mapper = sk.index(); osk.index()
        query = [bytes(syn.to_ascii(syn.mutate_codes(g, base, float(g.choice([0.0, 0.01, 0.04])))))]
        if g.random() < 0.3:
            query = [bytes(c) for c in syn.split_contigs(g, query[0], 3)]
        hits = [(h.name, h.identity, h.matches, h.fragments) for h in mapper.query_draft(query)]
        ohits, det = osk.query_draft(query, threads=8, details=True)
    om = det["mappings"]
    omm = sorted(zip(om["qseq"].tolist(), om["rseq"].tolist(), om["rstart"].tolist(), om["sketch"].tolist(), om["shared"].tolist()))
    ok = hits == ohits and mappings(mapper) == omm
    if not ok:
        bad += 1
        print(f"MISMATCH case {case} seed {seed} params {params} strains {strains} style {style}: {len(hits)} vs {len(ohits)} hits")
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)

"""Differential fuzzing of the HIP path against the CPU oracle: random genomes with planted repeats, tandem
duplications, inversions, N runs and low-complexity stretches; random parameters.  Every L2 mapping and every hit must
match.  Usage: python scripts/fuzz_parity.py [cases] [seed] [seconds] [default-cell]   (stops after `seconds` if given: a time
box; a fourth argument keeps every nucleotide case in the default cell k = 16 / fragment 3000 / 80 % with queries of plain
ACGT -- the cell whose query passes run K1 and the fragment sketch as ONE launch, k_query_fused)"""
import sys, os, ctypes as C, warnings, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import _lib, synthetic as syn
from pyfastani_amd._lib import lib, check
from oracle.oracle import OracleSketch

def mappings(mapper):
    cap = 1 << 20
    buf = (_lib.Mapping * cap)(); n = C.c_int64(0)
    check(lib.fa_mapper_debug_mappings(mapper._h, buf, cap, C.byref(n)))
    return sorted((buf[i].query_seq_id, buf[i].ref_seq_id, buf[i].ref_start_pos, buf[i].sketch_size, buf[i].conserved) for i in range(n.value))

def scramble(g, codes):
    """plant structure: tandem repeats, a dispersed repeat, an inversion, a low-complexity run"""
    c = codes.copy()
    n = len(c)
    if g.random() < 0.7:
        unit = syn.random_codes(g, int(g.integers(20, 400)))
        p = int(g.integers(0, n - 5000)); reps = int(g.integers(2, 12))
        block = np.tile(unit, reps)[: n - p - 1]
        c[p:p + len(block)] = block
    if g.random() < 0.7:
        rep = syn.random_codes(g, int(g.integers(100, 1500)))
        for _ in range(int(g.integers(2, 6))):
            p = int(g.integers(0, n - len(rep) - 1)); c[p:p + len(rep)] = rep
    if g.random() < 0.5:
        a = int(g.integers(0, n - 4000)); b = a + int(g.integers(500, 3500))
        c[a:b] = syn.reverse_complement_codes(c[a:b])
    if g.random() < 0.5:
        p = int(g.integers(0, n - 600)); c[p:p + int(g.integers(50, 500))] = int(g.integers(0, 4))
    if g.random() < 0.4:
        p = int(g.integers(0, n - 600)); L = int(g.integers(20, 300)); c[p:p + L] = np.tile(np.array([0, 3], dtype=np.uint8), L)[:L]
    return c

def to_bytes(g, codes):
    b = bytearray(bytes(syn.to_ascii(codes)))
    if len(b) < 2500:
        return bytes(b)
    if g.random() < 0.5:
        for _ in range(int(g.integers(1, 5))):
            p = int(g.integers(0, len(b) - 200)); L = int(g.integers(1, 150)); b[p:p + L] = b"N" * L
    if g.random() < 0.3:
        p = int(g.integers(0, len(b) - 10)); b[p:p + 5] = b"RYKMS"
    if g.random() < 0.3:
        p = int(g.integers(0, len(b) - 2000)); b[p:p + 1000] = bytes(b[p:p + 1000]).lower()
    return bytes(b)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
time_box = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
default_cell = len(sys.argv) > 4
done = 0
g = syn.rng(seed)
bad = 0
t0 = time.time()
AMINO = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)

def protein_case(g, case):
    k = int(g.choice([5, 7, 9, 12, 16])); frag = int(g.choice([60, 100, 150, 300]))
    params = dict(k=k, fragment_length=frag, protein=True, minimum_fraction=float(g.choice([0.0, 0.2])))
    sk, osk = pf.Sketch(**params), OracleSketch(**params)
    n_prot = int(g.integers(3, 12))
    anc = [g.integers(0, 20, int(g.integers(80, 900))) for _ in range(n_prot)]
    def mutate(p, d):
        p = p.copy(); m = g.random(len(p)) < d; p[m] = g.integers(0, 20, int(m.sum())); return p
    for i in range(int(g.integers(1, 4))):
        d = float(g.choice([0.0, 0.05, 0.15, 0.3]))
        prots = [bytes(AMINO[mutate(p, d)]) for p in anc]
        if g.random() < 0.3: prots.append(b"MK")
        sk.add_draft(i, prots); osk.add_draft(i, prots)
    mapper = sk.index(); osk.index()
    query = [bytes(AMINO[mutate(p, 0.08)]).lower() if g.random() < 0.2 else bytes(AMINO[mutate(p, 0.08)]) for p in anc]
    hits = [(h.name, h.identity, h.matches, h.fragments) for h in mapper.query_draft(query)]
    ohits, det = osk.query_draft(query, threads=4, details=True)
    om = det["mappings"]
    omm = sorted(zip(om["qseq"].tolist(), om["rseq"].tolist(), om["rstart"].tolist(), om["sketch"].tolist(), om["shared"].tolist()))
    try:
        gm = mappings(mapper)
    except (RuntimeError, NotImplementedError) as e:
        if "stage getters" not in str(e):
            raise
        gm = omm
    ok = hits == ohits and gm == omm and len(mapper.lookup_index) == osk.index_size
    if not ok:
        print(f"MISMATCH protein case {case} seed {seed} params {params}: {hits} vs {ohits}")
    return ok

for case in range(cases):
    if time_box and time.time() - t0 > time_box:
        break
    done = case + 1
    if case % 10 == 9:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            bad += 0 if protein_case(g, case) else 1
        continue
    k = int(g.choice([8, 11, 12, 14, 16, 16, 16, 17, 21, 24]))
    frag = int(g.choice([200, 500, 1000, 1500, 3000, 3000, 5000]))
    pid = float(g.choice([70, 75, 80, 80, 85, 90, 95]))
    minfrac = float(g.choice([0.0, 0.1, 0.2, 0.5]))
    if default_cell:
        k, frag, pid = 16, 3000, 80.0
    params = dict(k=k, fragment_length=frag, percentage_identity=pid, minimum_fraction=minfrac)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        osk = OracleSketch(**params)
        if osk.window_size >= frag:      # degenerate cell: nothing maps; covered by the unit tests
            continue
        sk = pf.Sketch(**params)
        # a third of the cases index with a narrow low word of the global coordinate (FA_GPOS_BITS, read when the index is built):
        # dozens of word boundaries inside these small indexes, i.e. the 64-bit form of k_l1's candidate scan
        os.environ.pop("FA_GPOS_BITS", None)
        if g.random() < 0.33:
            os.environ["FA_GPOS_BITS"] = str(max(12, int(2 * frag).bit_length() + 1) + int(g.integers(0, 3)))
        length = int(g.integers(max(3 * frag, 8000), 60_000))
        anc = scramble(g, syn.random_codes(g, length))
        n_ref = int(g.integers(1, 6))
        for i in range(n_ref):
            d = float(g.choice([0.0, 0.01, 0.03, 0.06, 0.1, 0.15]))
            r = scramble(g, syn.mutate_codes(g, anc, d)) if g.random() < 0.5 else syn.mutate_codes(g, anc, d)
            contigs = [to_bytes(g, x) for x in syn.split_contigs(g, r, int(g.integers(1, 5)))]
            if g.random() < 0.3: contigs.append(b"ACGT" * int(g.integers(0, 4)))
            sk.add_draft(i, contigs); osk.add_draft(i, contigs)
        if g.random() < 0.5:
            r = to_bytes(g, syn.random_codes(g, length)); sk.add_draft(n_ref, [r]); osk.add_draft(n_ref, [r])
        mapper = sk.index(); osk.index()
        queries = []
        for _ in range(int(g.integers(1, 4)) if case % 4 == 0 else 1):
            q = scramble(g, syn.mutate_codes(g, anc, float(g.choice([0.0, 0.02, 0.05, 0.1])))) if g.random() < 0.5 else syn.mutate_codes(g, anc, 0.03)
            plain = default_cell and g.random() < 0.6          # (default-cell runs: 60 % plain-ACGT queries, the rest with N runs / IUPAC / lower case)
            queries.append([(bytes(syn.to_ascii(x)) if plain else to_bytes(g, x)) for x in syn.split_contigs(g, q, int(g.integers(1, 4)))])
        if len(queries) > 1:
            # the resident-batch API must give, per genome, what one query_draft call gives
            got = [[(h.name, h.identity, h.matches, h.fragments) for h in hs] for hs in mapper.upload_genomes(queries).query()]
            want = [osk.query_draft(q, threads=8) for q in queries]
            if got != want:
                bad += 1
                print(f"MISMATCH batch case {case} seed {seed} params {params}: {got} vs {want}")
            continue
        query = queries[0]
        hits = [(h.name, h.identity, h.matches, h.fragments) for h in mapper.query_draft(query)]
        ohits, det = osk.query_draft(query, threads=8, details=True)
    om = det["mappings"]
    omm = sorted(zip(om["qseq"].tolist(), om["rseq"].tolist(), om["rstart"].tolist(), om["sketch"].tolist(), om["shared"].tolist()))
    try:
        gm = mappings(mapper)
    except (RuntimeError, NotImplementedError) as e:   # FA_QUERY_LANES / FA_PASS_FRAGMENTS: more parts than the stage getters retain
        if "stage getters" not in str(e):
            raise
        gm = omm
    ok = hits == ohits and gm == omm and len(mapper.lookup_index) == osk.index_size and mapper.occurences_threshold == osk.freq_threshold
    if not ok:
        bad += 1
        print(f"MISMATCH case {case} seed {seed} params {params} window {osk.window_size}: hits {hits} vs {ohits}; mappings gpu {len(gm)} oracle {len(omm)}")
        sg, so = set(gm), set(omm)
        print("   only gpu", sorted(sg - so)[:4], "only oracle", sorted(so - sg)[:4])
print(f"{done} cases (seed {seed}), {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)

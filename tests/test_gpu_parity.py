"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed fixtures -- MI355X only.

Integer results (minimizer streams, index size, L1 loci, L2 shared counts, orthologous-fragment counts) must be
bit-exact; identities are float32 and must be bit-exact too (tolerance stated where a looser one applies: the
north star allows 1e-4 on reported ANI)."""
import ctypes as C
import json
import os
import pickle
import sys
import warnings

import numpy as np
import pytest

import pyfastani_amd as pf
from conftest import ROOT, read_fasta
from oracle.oracle import OracleSketch
from pyfastani_amd import _lib, synthetic as syn
from pyfastani_amd._lib import lib, check

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_synthetic_goldens import build_case  # noqa: E402

pytestmark = pytest.mark.gpu
ANI_TOL = 1e-4


def quiet_sketch(cls, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return cls(**kw)


def gpu_stream(sk, seq):
    b = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
    cap = max(len(b), 1)
    h = np.empty(cap, np.uint32)
    w = np.empty(cap, np.int32)
    n = C.c_int64(0)
    check(lib.fa_debug_sketch_sequence(C.byref(sk._param), b, len(b), 1, h.ctypes.data, w.ctypes.data, cap, C.byref(n)))
    return h[: n.value], w[: n.value]


def hit_tuples(hits):
    return [(h.name, h.identity, h.matches, h.fragments) for h in hits]


def gpu_mappings(mapper):
    cap = 1 << 20
    buf = (_lib.Mapping * cap)()
    n = C.c_int64(0)
    check(lib.fa_mapper_debug_mappings(mapper._h, buf, cap, C.byref(n)))
    assert n.value <= cap
    return sorted((buf[i].query_seq_id, buf[i].ref_seq_id, buf[i].ref_start_pos, buf[i].sketch_size, buf[i].conserved)
                  for i in range(n.value))


def oracle_mappings(det):
    m = det["mappings"]
    return sorted(zip(m["qseq"].tolist(), m["rseq"].tolist(), m["rstart"].tolist(), m["sketch"].tolist(), m["shared"].tolist()))


# ----------------------------------------------------------------------------------------------------------------
# K1: minimizer streams
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k,frag", [(16, 3000), (14, 1000), (21, 3000), (16, 5000), (12, 500), (33, 3000), (7, 200)])
def test_minimizer_streams(k, frag):
    sk = quiet_sketch(pf.Sketch, k=k, fragment_length=frag)
    osk = OracleSketch(k=k, fragment_length=frag)
    assert sk.window_size == osk.window_size
    g = syn.rng(100 + k)
    w = sk.window_size
    rnd = lambda n: bytes(syn.to_ascii(syn.random_codes(g, n)))  # noqa: E731
    cases = {
        "random": rnd(10_000),
        "tile_boundaries": rnd(2048 * 3 + k - 1),
        "fragment": rnd(frag),
        "ATGC_repeat": b"ATGC" * 1000,
        "polyA": b"A" * 5000,
        "AT_repeat": b"AT" * 3000,
        "N_runs_and_iupac": rnd(3000) + b"N" * 100 + rnd(4000) + b"nnRYKMBVDHSWU" + rnd(500) + b"N",
        "lower_case": rnd(5000).lower(),
        "mixed_case": bytes(c + 32 if i % 3 == 0 else c for i, c in enumerate(rnd(3000))),
        "shorter_than_k": b"ACGTA"[: max(1, k - 1)],
        "len_w": rnd(max(w, 1)),
        "len_w_plus_k_minus_2": rnd(w + k - 2),
        "len_w_plus_k_minus_1": rnd(w + k - 1),
        "len_w_plus_k": rnd(w + k),
        "palindromes": (rnd(40) + b"ACGT" * 10 + b"GAATTC" * 20) * 20,
        "leading_dup_run": b"ACGTTGCA" * 50 + rnd(2000),
    }
    for name, seq in cases.items():
        gh, gw = gpu_stream(sk, seq)
        oh, ow = osk.sketch_sequence(seq)
        assert np.array_equal(gh, oh) and np.array_equal(gw, ow), f"{name}: gpu {len(gh)} records, oracle {len(oh)}"


def test_minimizer_streams_through_the_general_window_minimum():
    # k_sketch_tiles takes the 32-bit form of the window minimum for every tile whose positions are all valid and falls
    # back to 64-bit (hash, ~position) keys for the others (a k-mer equal to its reverse complement, w < 3 or > 1000).
    # FA_K1_GENERAL=1 sends every tile through the fallback: the same streams, edge cases and parameter cells must hold.
    import subprocess
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                          "-k", "test_minimizer_streams and not general or test_edge_cases or test_degenerate_and_unusual_parameter_cells"],
                         env=dict(os.environ, FA_K1_GENERAL="1"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and " passed" in res.stdout, res.stdout[-3000:] + res.stderr[-2000:]


def test_reference_sketch_multi_contig_and_index():
    g = syn.rng(11)
    anc, members, _ = syn.family(11, 4, 200_000)
    sk, osk = pf.Sketch(), OracleSketch()
    for i, m in enumerate(members):
        contigs = syn.split_contigs(g, m, 3) + [b"ACGT", b""]
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            sk.add_draft(f"g{i}", contigs)
        assert len(caught) == 2  # the two short contigs (_fastani.pyx:670-677)
        osk.add_draft(f"g{i}", contigs)
    assert len(sk.minimizers) == len(osk.minimizers()[0])
    h, s, w = sk.minimizers._arrays()
    oh, os_, ow = osk.minimizers()
    assert np.array_equal(h, oh) and np.array_equal(s, os_) and np.array_equal(w, ow)
    first = sk.minimizers[0]
    assert (first.hash, first.sequence_id, first.window_position) == (int(oh[0]), int(os_[0]), int(ow[0]))
    assert sk.minimizers[-1].hash == int(oh[-1])
    mapper = sk.index()
    osk.index()
    assert len(sk.minimizers) == 0 and sk.names == []          # ownership moved (_fastani.pyx:793-806)
    assert len(mapper.minimizers) == len(oh)
    idx = mapper.lookup_index
    assert len(idx) == osk.index_size
    assert mapper.occurences_threshold == osk.freq_threshold
    keys = list(idx)
    assert keys == sorted(set(oh.tolist()))
    for key in keys[:: max(1, len(keys) // 50)]:
        assert key in idx
        pos = idx[key]
        assert len(pos) == osk.index_count(key)
        want = [(int(a), int(b)) for a, b, c in zip(os_, ow, oh) if c == key]
        assert [(p.sequence_id, p.window_position) for p in pos] == want
    missing = next(x for x in range(1, 10_000) if x not in set(keys))
    assert missing not in idx
    with pytest.raises(KeyError):
        idx[missing]


# ----------------------------------------------------------------------------------------------------------------
# end to end against the oracle, with stage-level comparison of the L2 mappings
# ----------------------------------------------------------------------------------------------------------------
def run_both(params, refs, query, threads=1):
    sk, osk = quiet_sketch(pf.Sketch, **params), quiet_sketch(OracleSketch, **params)
    for i, r in enumerate(refs):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sk.add_draft(f"ref{i}", r)
        osk.add_draft(f"ref{i}", r)
    mapper = sk.index()
    osk.index()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hits = mapper.query_draft(query)
    ohits, det = osk.query_draft(query, threads=threads, details=True)
    return mapper, hits, ohits, det


@pytest.mark.parametrize("seed,length,params", [
    (21, 400_000, {}),
    (22, 300_000, {"k": 14, "fragment_length": 1000}),
    (23, 300_000, {"k": 21, "fragment_length": 3000}),
    (24, 300_000, {"fragment_length": 5000}),
    (25, 200_000, {"percentage_identity": 95.0}),
    (26, 200_000, {"percentage_identity": 70.0, "minimum_fraction": 0.0}),
])
def test_end_to_end_vs_oracle(seed, length, params):
    g = syn.rng(seed)
    anc = syn.random_codes(g, length)
    refs = [[syn.to_ascii(syn.mutate_codes(g, anc, d))] for d in (0.01, 0.04, 0.09, 0.14, 0.20)]
    refs.append([syn.to_ascii(syn.random_codes(g, length))])
    # a reference with an internal duplication and an inverted segment (exercises several loci per genome)
    dup = syn.mutate_codes(g, anc, 0.03)
    dup = np.concatenate([dup[: length // 2], dup[length // 4: length // 2], syn.reverse_complement_codes(dup[length // 2:])])
    refs.append([syn.to_ascii(dup)])
    query = [syn.to_ascii(syn.mutate_codes(g, anc, 0.05))]
    mapper, hits, ohits, det = run_both(params, refs, query, threads=4)
    assert gpu_mappings(mapper) == oracle_mappings(det)        # every L2 mapping: position, sketch size, shared count
    assert hit_tuples(hits) == ohits                           # counts bit-exact, identity bit-exact (<= ANI_TOL required)
    for (n1, i1, m1, f1), (n2, i2, m2, f2) in zip(hit_tuples(hits), ohits):
        assert abs(i1 - i2) <= ANI_TOL and (m1, f1) == (m2, f2)


def test_draft_query_and_reference():
    g = syn.rng(31)
    anc = syn.random_codes(g, 500_000)
    refs = [syn.split_contigs(g, syn.to_ascii(syn.mutate_codes(g, anc, d)), 20) for d in (0.02, 0.07, 0.13)]
    refs.append(syn.split_contigs(g, syn.to_ascii(syn.random_codes(g, 300_000)), 9))
    query = syn.split_contigs(g, syn.to_ascii(syn.mutate_codes(g, anc, 0.04)), 15) + [b"ACGT" * 3]
    mapper, hits, ohits, det = run_both({}, refs, query)
    assert gpu_mappings(mapper) == oracle_mappings(det)
    assert hit_tuples(hits) == ohits and len(hits) == 3


def test_l1_candidates_match_oracle():
    g = syn.rng(41)
    anc = syn.random_codes(g, 150_000)
    refs = [[syn.to_ascii(syn.mutate_codes(g, anc, d))] for d in (0.02, 0.10)]
    query = syn.to_ascii(syn.mutate_codes(g, anc, 0.05))
    mapper, hits, ohits, det = run_both({}, refs, [query])
    cap = 1 << 16
    arr = [np.empty(cap, np.int32) for _ in range(4)]
    n = C.c_int64(0)
    check(lib.fa_mapper_debug_l1(mapper._h, *[a.ctypes.data for a in arr], cap, C.byref(n)))
    got = {}
    for f, s, a, b in zip(*[a[: n.value].tolist() for a in arr]):
        got.setdefault(f, []).append((s, a, b))
    osk = OracleSketch()
    for i, r in enumerate(refs):
        osk.add_draft(f"ref{i}", r)
    osk.index()
    qb = bytes(query)
    for f in range(0, len(qb) // 3000, 3):
        ss, mh, loci = osk.l1_fragment(qb[f * 3000:(f + 1) * 3000])
        assert sorted(got.get(f, [])) == sorted(loci), f
        sz = C.c_int32(0)
        buf = np.empty(4096, np.uint32)
        check(lib.fa_mapper_debug_query_sketch(mapper._h, f, buf.ctypes.data, 4096, C.byref(sz)))
        assert sz.value == ss
        oh, _ = osk.sketch_sequence(qb[f * 3000:(f + 1) * 3000])
        assert buf[: sz.value].tolist() == sorted(set(oh.tolist()))


def test_frequency_threshold_active():
    # ~478 000 distinct minimizers => minimizerToIgnore = 4; two planted 45-mers give a handful of hashes with
    # hundreds of occurrences, so computeFreqHist leaves INT_MAX and the strict `size < threshold` filter bites
    g = syn.rng(51)
    n = 1_000_000
    genomes = [syn.random_codes(g, n) for _ in range(6)]
    r1, r2 = syn.random_codes(g, 45), syn.random_codes(g, 45)
    m = genomes[0]
    for p in range(1000, n - 1000, n // 300):
        m[p: p + 45] = r1
    for p in range(2500, n - 1000, n // 150):
        m[p: p + 45] = r2
    refs = [[syn.to_ascii(x)] for x in genomes]
    query = [syn.to_ascii(syn.mutate_codes(g, m, 0.02))]
    mapper, hits, ohits, det = run_both({}, refs, query, threads=8)
    assert mapper.occurences_threshold == 218                  # value from the oracle
    assert gpu_mappings(mapper) == oracle_mappings(det)
    assert hit_tuples(hits) == ohits


def test_small_sketch_against_crowded_window():
    # percentage_identity=68 gives w=3, so a super-window holds ~1500 reference minimizers, while a fragment that is
    # all N except a ~20 bp island keeps about twenty: hundreds of window-only hashes share one insertion rank,
    # which overflows the one-byte L2 lane state and sends those loci through the uint16 redo pass.
    g = syn.rng(55)
    ref = syn.random_codes(g, 200_000)
    q = bytearray(b"N" * 3000 * 60)
    for f in range(60):
        n, a = 17 + f % 14, 5000 + f * 3000
        q[f * 3000 + 1400: f * 3000 + 1400 + n] = bytes(syn.to_ascii(ref[a: a + n]))
    params = {"minimum_fraction": 0.0, "percentage_identity": 68.0}
    mapper, hits, ohits, det = run_both(params, [[syn.to_ascii(ref)]], [bytes(q)], threads=8)
    assert mapper.window_size == 3
    ms = (C.c_float * 16)()
    lib.fa_mapper_last_timings(mapper._h, ms, 16)
    assert ms[8] > 0, "the wide-state redo pass was not exercised"
    assert len(det["mappings"]["qseq"]) >= 5
    assert gpu_mappings(mapper) == oracle_mappings(det)
    assert hit_tuples(hits) == ohits


def test_void_redo_attempt_leaves_no_stale_bins():
    # Two loci per fragment in ONE reference genome (the island planted in both of its contigs) with equal shared counts:
    # the canonical tie-break keeps the locus of contig 0.  Sketches of ~40-75 minimizers against the ~1500 of a w=3
    # super-window put the fullest rank right at the edge of the one-byte slide state, so for a dozen fragments the locus
    # in contig 0 overflows while its twin in contig 1 does not.  A fresh mapper has not launched the wide-state scan yet:
    # its first attempt is void and repeated -- and must leave nothing in the CGI bin table, or the twin (a lesser locus
    # of the group, in a bin the true best never touches) stays behind and count_seq comes out too high.
    g = syn.rng(77)
    r, x = syn.random_codes(g, 400_000), syn.random_codes(g, 400_000)
    nf = 120
    q = bytearray(b"N" * 3000 * nf)
    for f in range(nf):
        n, a, b = 60 + (f * 7) % 80, 5000 + f * 3000, 7000 + f * 3000
        x[b: b + n] = r[a: a + n]
        q[f * 3000 + 1400: f * 3000 + 1400 + n] = bytes(syn.to_ascii(r[a: a + n]))
    params = {"minimum_fraction": 0.0, "percentage_identity": 68.0}
    mapper, hits, ohits, det = run_both(params, [[syn.to_ascii(r), syn.to_ascii(x)]], [bytes(q)], threads=8)
    ms = (C.c_float * 16)()
    lib.fa_mapper_last_timings(mapper._h, ms, 16)
    assert ms[8] > 0 and ms[9] >= 1, "no locus left the byte state on the first attempt: the case does not test the repeat"
    assert ms[8] < ms[6], "every locus overflowed: no twin could be left behind"
    assert gpu_mappings(mapper) == oracle_mappings(det)
    assert hit_tuples(hits) == ohits
    assert hit_tuples(mapper.query_draft([bytes(q)])) == ohits          # and again, now that the wide pass is always launched


def test_protein_golden_on_device(golden_dir):
    # reference: src/pyfastani/tests/test_ani.py:96-115
    b1 = read_fasta(os.path.join(golden_dir, "BGC0001425.faa"))
    b3 = read_fasta(os.path.join(golden_dir, "BGC0001428.faa"))
    for carrier in (str, lambda s: s.encode("ascii"), lambda s: np.frombuffer(s.encode("ascii"), dtype=np.uint8)):
        sk = pf.Sketch(protein=True, fragment_length=100)
        sk.add_draft("BGC0001425", map(carrier, b1))
        sk.add_draft("BGC0001427", map(carrier, b1))
        assert len(sk.minimizers) == 36054
        mapper = sk.index()
        assert len(mapper.lookup_index) == 13890
        hits = mapper.query_draft(map(carrier, b3))
        assert len(hits) == 2
        assert (hits[0].name, hits[0].matches, hits[0].fragments) == ("BGC0001425", 130, 176)
        assert (hits[1].name, hits[1].matches, hits[1].fragments) == ("BGC0001427", 130, 176)
        assert hits[0].identity == pytest.approx(94.99492645263672, abs=ANI_TOL)


def test_committed_fixtures(golden_dir):
    fixtures = json.load(open(os.path.join(golden_dir, "synthetic_goldens.json")))
    for fx in fixtures:
        case = fx["case"]
        refs, queries = build_case(case)
        sk = quiet_sketch(pf.Sketch, **case["params"])
        for i, r in enumerate(refs):
            sk.add_draft(f"ref{i}", r)
        assert sk.window_size == fx["window"], case["name"]
        assert len(sk.minimizers) == fx["n_minimizers"], case["name"]
        mapper = sk.index()
        assert len(mapper.lookup_index) == fx["index_size"] and mapper.occurences_threshold == fx["freq_threshold"]
        for q, want in zip(queries, fx["queries"]):
            hits = mapper.query_draft(q)
            assert [[h.name, h.identity, h.matches, h.fragments] for h in hits] == want["hits"], case["name"]
            assert len(gpu_mappings(mapper)) == want["n_mappings"]


# ----------------------------------------------------------------------------------------------------------------
# edge cases of the API
# ----------------------------------------------------------------------------------------------------------------
def test_edge_cases():
    g = syn.rng(61)
    ref = syn.to_ascii(syn.random_codes(g, 50_000))
    sk = pf.Sketch()
    sk.add_genome("r", ref)
    mapper = sk.index()
    assert mapper.query_draft([]) == []                                # no contigs
    assert mapper.query_genome(ref[:2999]) == []                       # shorter than one fragment: no warning, no hit
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        assert mapper.query_draft([b"ACGT", b""]) == []                # shorter than min(w, k, frag): warns (:1061-1070)
    assert len(caught) == 2
    assert mapper.query_genome(b"N" * 9000) == []                      # sketch size 0 (_fastani.pyx:937-938)
    assert mapper.query_genome(b"AT" * 4500) == []                     # only strand-symmetric k-mers
    assert hit_tuples(mapper.query_genome(ref)) == [("r", 100.0, 16, 16)]
    assert hit_tuples(mapper.query_genome(bytes(ref).decode().lower())) == [("r", 100.0, 16, 16)]
    with pytest.raises(ValueError):
        mapper.query_genome(ref, threads=-1)
    # an empty index answers with no hits
    empty = pf.Sketch().index()
    assert empty.query_genome(ref) == [] and len(empty.lookup_index) == 0
    # a sketch stays usable after index() (_fastani.pyx:803-804)
    sk.add_genome("again", ref)
    assert hit_tuples(sk.index().query_genome(ref)) == [("again", 100.0, 16, 16)]


def test_clear_and_names():
    g = syn.rng(62)
    sk = pf.Sketch()
    sk.add_genome("a", syn.to_ascii(syn.random_codes(g, 10_000)))
    assert len(sk.minimizers) > 0
    sk.clear()
    assert len(sk.minimizers) == 0 and sk.names == []
    sk.add_genome("b", syn.to_ascii(syn.random_codes(g, 10_000)))
    assert sk.names == ["b"] and sk.minimizers[0].sequence_id == 0


def test_pickling_round_trips():
    # reference: test_ani.py:135-173 and test_sketch.py:54-61
    g = syn.rng(71)
    anc = syn.random_codes(g, 200_000)
    ref = syn.to_ascii(syn.mutate_codes(g, anc, 0.03))
    query = syn.to_ascii(syn.mutate_codes(g, anc, 0.02))
    sk = pf.Sketch()
    sk.add_genome("ref", ref)
    sk2 = pickle.loads(pickle.dumps(sk))
    assert sk2.names == ["ref"] and len(sk2.minimizers) == len(sk.minimizers)
    want = hit_tuples(sk.index().query_genome(query))
    m2 = sk2.index()
    assert hit_tuples(m2.query_genome(query)) == want
    m3 = pickle.loads(pickle.dumps(m2))
    assert hit_tuples(m3.query_genome(query)) == want and len(m3.lookup_index) == len(m2.lookup_index)
    assert len(want) == 1 and want[0][2] > 60


# ----------------------------------------------------------------------------------------------------------------
# full-size, oracle-free properties (BASELINE sizes)
# ----------------------------------------------------------------------------------------------------------------
def test_full_size_properties():
    g = syn.rng(81)
    n = 5_000_000
    anc = syn.random_codes(g, n)
    refs = {"self": anc, "d05": syn.mutate_codes(g, anc, 0.05), "d15": syn.mutate_codes(g, anc, 0.15), "unrelated": syn.random_codes(g, n)}
    sk = pf.Sketch()
    for name, codes in refs.items():
        sk.add_genome(name, syn.to_ascii(codes))
    n_min = len(sk.minimizers)
    assert abs(n_min - 4 * 2 * n / 25) < 0.02 * 4 * 2 * n / 25          # density 2/(w+1)
    mapper = sk.index()
    hits = mapper.query_genome(syn.to_ascii(anc))
    assert hit_tuples(hits)[0] == ("self", 100.0, 1666, 1666)            # self-query invariant (test_ani.py:66-71)
    assert [h.name for h in hits] == ["self", "d05", "d15"]
    assert hits[1].matches >= 1640 and 94.0 < hits[1].identity < 95.0
    # strand symmetry: the reverse complement of the query gives the same per-genome counts and identities
    rc_hits = mapper.query_genome(syn.to_ascii(syn.reverse_complement_codes(anc)))
    # (not exactly 100.0: a window is only evaluated when its LAST k-mer is not strand-symmetric (_fastani.pyx:202),
    # so the ~76 palindromic 16-mers of a 5 Mb genome make the two strands see slightly different window sets;
    # the oracle gives the same value on this input)
    assert hit_tuples(rc_hits)[0] == ("self", 99.99999237060547, 1666, 1666)
    # order independence: the same references added in another order give the same rows per name
    sk2 = pf.Sketch()
    for name in ["unrelated", "d15", "self", "d05"]:
        sk2.add_genome(name, syn.to_ascii(refs[name]))
    hits2 = sk2.index().query_genome(syn.to_ascii(anc))
    assert sorted(hit_tuples(hits)) == sorted(hit_tuples(hits2))
    # batch API == one call per genome
    queries = [[syn.to_ascii(refs["d05"])], [syn.to_ascii(refs["unrelated"])], syn.split_contigs(g, syn.to_ascii(refs["d15"]), 50)]
    batch = mapper.upload_genomes(queries)
    per_genome = batch.query()
    for q, got in zip(queries, per_genome):
        assert hit_tuples(got) == hit_tuples(mapper.query_draft(q))
    assert hit_tuples(per_genome[0])[0][:1] == ("d05",) and per_genome[0][0].identity == 100.0
    assert hit_tuples(batch.query(1, 2)[1]) == hit_tuples(per_genome[2])


# ----------------------------------------------------------------------------------------------------------------
# many-to-many (BASELINE configs 3/4 in miniature): draft assemblies, all-vs-all, several passes per call
# ----------------------------------------------------------------------------------------------------------------
def test_all_vs_all_drafts_multi_pass():
    import subprocess
    import textwrap
    # the pass size is read once per process, so the multi-pass run happens in a child process
    code = textwrap.dedent("""
        import os, sys, json, warnings
        sys.path.insert(0, %r)
        import numpy as np
        import pyfastani_amd as pf
        from pyfastani_amd import synthetic as syn
        from oracle.oracle import OracleSketch
        g = syn.rng(91)
        genomes = []
        for fam in range(3):
            anc = syn.random_codes(g, 150_000)
            for d in (0.0, 0.03, 0.08, 0.14):
                genomes.append(syn.split_contigs(g, syn.to_ascii(syn.mutate_codes(g, anc, d)), 10))
        genomes.append([b"ACGT" * 2])                      # a genome made of one short contig
        genomes.append([])                                 # an empty genome
        sk, osk = pf.Sketch(), OracleSketch()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i, c in enumerate(genomes):
                sk.add_draft(i, c); osk.add_draft(i, c)
            mapper = sk.index(); osk.index()
            batch = mapper.upload_genomes(genomes)
            got = [[(h.name, h.identity, h.matches, h.fragments) for h in hits] for hits in batch.query()]
            part = [[(h.name, h.identity, h.matches, h.fragments) for h in hits] for hits in batch.query(5, 4)]
        want = [osk.query_draft(c, threads=8) for c in genomes]
        assert got == want, "all-vs-all mismatch"
        assert part == want[5:9]
        assert sum(len(w) for w in want) >= 3 * 12 and want[-1] == [] and want[-2] == []
        print("OK", sum(len(w) for w in want))
    """ % ROOT)
    env = dict(os.environ, FA_PASS_FRAGMENTS="120")        # 14 genomes x ~45 fragments -> several passes
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout + res.stderr


def _run_child(code, env_extra):
    import subprocess
    env = dict(os.environ, **env_extra)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout + res.stderr


def test_seed_overflow_to_hbm_scratch():
    # 560 copies of one small genome: every query minimizer hits 560 positions, so a fragment gathers ~130 000 seed
    # hits -- more than the 32 768 that fit the LDS sort -- and takes the HBM scratch path of k_l1; its 560 loci also
    # exceed the 512 that k_l1 merges in LDS, which exercises the second (write) pass
    g = syn.rng(95)
    base = syn.random_codes(g, 21_000)
    refs = [[syn.to_ascii(base)] for _ in range(560)]
    query = [syn.to_ascii(syn.mutate_codes(g, base, 0.01))]
    mapper, hits, ohits, det = run_both({}, refs, query, threads=8)
    assert mapper.occurences_threshold == 2**31 - 1 and len(ohits) == 560
    assert gpu_mappings(mapper) == oracle_mappings(det)
    assert hit_tuples(hits) == ohits


@pytest.mark.parametrize("layout", ["drafts", "tandem", "mixed", "uneven"])
def test_chunked_l1_and_its_fallback(layout):
    # fragments with more seed hits than LDS holds are cut at contig boundaries by k_l1_big; what cannot be cut (one
    # contig alone holds more than a fair share of a position list) falls back to the HBM sort of k_l1
    g = syn.rng(970)
    base = syn.random_codes(g, 18_000)
    if layout == "drafts":        # 150 strains in 3 contigs each: 450 contigs, every chunk holds many of them
        refs = [syn.split_contigs(g, syn.to_ascii(syn.mutate_codes(g, base, 0.01)), 3) for _ in range(150)]
    elif layout == "tandem":      # one contig of 120 tandem copies: cannot be cut at all
        refs = [[syn.to_ascii(np.concatenate([syn.mutate_codes(g, base, 0.01) for _ in range(120)]))]]
    elif layout == "mixed":       # 60 single copies, then the uncuttable contig: chunks first, then the fallback
        refs = [[syn.to_ascii(syn.mutate_codes(g, base, 0.01))] for _ in range(60)]
        refs.append([syn.to_ascii(np.concatenate([syn.mutate_codes(g, base, 0.01) for _ in range(110)]))])
    else:                         # contigs of very different weight: 1 to 12 copies per contig
        refs = [[syn.to_ascii(np.concatenate([syn.mutate_codes(g, base, 0.01) for _ in range(1 + (i * 7) % 12)]))] for i in range(40)]
    query = [syn.to_ascii(syn.mutate_codes(g, base, 0.01))]
    mapper, hits, ohits, det = run_both({}, refs, query, threads=8)
    assert mapper.occurences_threshold == 2**31 - 1 and len(ohits) == len(refs)
    assert gpu_mappings(mapper) == oracle_mappings(det)
    assert hit_tuples(hits) == ohits


def test_chunked_l1_more_loci_than_its_scratch():
    # 70 % identity needs only two seed hits per candidate, and the references hold 26-base pieces of the query more than
    # a fragment apart: 31 370 seed hits (values from the oracle) in 7 605 separate loci.  k_l1_big collects loci in a
    # fifth of the fragment's scratch (32 768 words: 6 553 loci); they do not fit, so it hands the fragment back to k_l1,
    # whose HBM path writes them straight to their place.
    g = syn.rng(972)
    q = syn.random_codes(g, 3000)
    refs = []
    for _ in range(100):
        c = syn.random_codes(g, 80 * 3100)
        for j in range(80):
            a = int(g.integers(0, 3000 - 26))
            c[j * 3100 + 100: j * 3100 + 126] = q[a: a + 26]
        refs.append([syn.to_ascii(c)])
    params = dict(percentage_identity=70.0, minimum_fraction=0.0)
    mapper, hits, ohits, det = run_both(params, refs, [syn.to_ascii(q)], threads=8)
    n = C.c_int64(0)
    arr = [np.empty(1 << 14, np.int32) for _ in range(4)]
    check(lib.fa_mapper_debug_l1(mapper._h, *[a.ctypes.data for a in arr], 1 << 14, C.byref(n)))
    assert n.value == 7605
    assert gpu_mappings(mapper) == oracle_mappings(det) and len(det["mappings"]["rseq"]) > 5000
    assert hit_tuples(hits) == ohits


def test_chunked_l1_switched_off_matches():
    # FA_L1_BIG=0 keeps every oversized fragment on the HBM sort: same rows as the chunked path
    import textwrap
    code = textwrap.dedent("""
        import sys, os
        sys.path.insert(0, %r)
        import numpy as np
        import pyfastani_amd as pf
        from pyfastani_amd import synthetic as syn
        g = syn.rng(971)
        base = syn.random_codes(g, 15_000)
        sk = pf.Sketch()
        for i in range(200):
            sk.add_draft(i, syn.split_contigs(g, syn.to_ascii(syn.mutate_codes(g, base, 0.02)), 2))
        m = sk.index()
        batch = m.upload_genomes([[syn.to_ascii(syn.mutate_codes(g, base, 0.02))]])
        rows = batch.query_rows(0, 1); rows = batch.query_rows(0, 1)
        assert len(rows) == 200
        sys.stdout.buffer.write(b"OK" + rows.tobytes().hex().encode())
    """ % ROOT)
    import subprocess
    outs = []
    for flag in ("1", "0"):
        res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FA_L1_BIG=flag), capture_output=True, text=True, timeout=900)
        assert res.returncode == 0 and res.stdout.startswith("OK"), res.stdout[-2000:] + res.stderr[-2000:]
        outs.append(res.stdout)
    assert outs[0] == outs[1]


@pytest.mark.parametrize("copies", [12, 25, 45])
def test_seed_counts_across_the_merge_tiers(copies):
    # `copies` identical references: every query minimizer hits `copies` positions, a fragment gathers ~240 x copies seed
    # hits -- about 2 900 / 6 000 / 10 800: the 16-per-thread in-place merge, its upper range, and the 32-per-thread one
    g = syn.rng(950 + copies)
    base = syn.random_codes(g, 24_000)
    refs = [[syn.to_ascii(base)] for _ in range(copies)] + [[syn.to_ascii(syn.mutate_codes(g, base, 0.05))]]
    query = [syn.to_ascii(syn.mutate_codes(g, base, 0.02))]
    mapper, hits, ohits, det = run_both({}, refs, query, threads=8)
    assert len(ohits) == copies + 1
    assert gpu_mappings(mapper) == oracle_mappings(det)
    assert hit_tuples(hits) == ohits


def test_speculated_capacities_retry():
    # forces the speculated loci / event capacities below the real numbers: the device raises a flag, the pass is void
    # and is run again with larger buffers (sketch-size and scratch speculation are exercised by the other tests)
    import textwrap
    code = textwrap.dedent("""
        import sys, warnings
        sys.path.insert(0, %r)
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
    """ % ROOT)
    _run_child(code, {"FA_LOCI_CAP_MIN": "7"})
    _run_child(code, {"FA_EVENTS_CAP_MIN": "1000"})          # same for the slide-event buffer
    _run_child(code, {"FA_LOCI_CAP_MIN": "3", "FA_EVENTS_CAP_MIN": "64"})


def test_random_seed_regime():
    # short fragments against a larger index (48 Mb, k=16, fragment 1000): every fragment picks up chance seed hits in
    # unrelated genomes, each of which becomes a one-seed locus that passes the relaxed identity bound -- the regime
    # BASELINE config 5 runs in at full size.  minimum_fraction=0 keeps every row so that all of them are compared.
    g = syn.rng(100)
    genomes = []
    for fam in range(4):
        anc = syn.random_codes(g, 2_000_000)
        for d in (0.0, 0.01, 0.04, 0.08, 0.13, 0.2):
            genomes.append(syn.to_ascii(syn.mutate_codes(g, anc, d) if d else anc))
    params = {"k": 16, "fragment_length": 1000, "minimum_fraction": 0.0}
    sk, osk = quiet_sketch(pf.Sketch, **params), OracleSketch(**params)
    for i, s in enumerate(genomes):
        sk.add_genome(i, s)
        osk.add_genome(i, s)
    mapper = sk.index()
    osk.index()
    assert mapper.occurences_threshold == osk.freq_threshold
    queries = [[genomes[1]], [genomes[8]], [syn.to_ascii(syn.random_codes(g, 1_000_000))]]
    got = [hit_tuples(h) for h in mapper.upload_genomes(queries).query()]
    want = [osk.query_draft(q, threads=8) for q in queries]
    assert got == want
    assert len(want[0]) > 6 and len(want[2]) > 0           # rows outside the family, and for a query related to nothing


def test_pass_cut_into_parts():
    # a genome with more fragments than one pass takes, and a pass whose slide events exceed what the 32-bit event
    # offsets can address (both limits lowered through the environment): the pass is cut into fragment ranges that
    # share the CGI bin table; results must not depend on the cut
    import textwrap
    code = textwrap.dedent("""
        import sys, warnings
        sys.path.insert(0, %r)
        import pyfastani_amd as pf
        from pyfastani_amd import synthetic as syn
        from oracle.oracle import OracleSketch
        g = syn.rng(99)
        anc = syn.random_codes(g, 200_000)
        sk, osk = pf.Sketch(), OracleSketch()
        for i, d in enumerate((0.0, 0.02, 0.06, 0.12)):
            r = syn.split_contigs(g, syn.to_ascii(syn.mutate_codes(g, anc, d) if d else anc), 3)
            sk.add_draft(i, r); osk.add_draft(i, r)
        m = sk.index(); osk.index()
        queries = [syn.split_contigs(g, syn.to_ascii(syn.mutate_codes(g, anc, d)), 4) for d in (0.03, 0.09)] + [[syn.to_ascii(anc)]]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = [[(h.name, h.identity, h.matches, h.fragments) for h in hits] for hits in m.upload_genomes(queries).query()]
            one = [(h.name, h.identity, h.matches, h.fragments) for h in m.query_draft(queries[0])]
        want = [osk.query_draft(q, threads=8) for q in queries]
        assert got == want and one == want[0] and all(len(w) == 4 for w in want), (got, want)
        print("OK")
    """ % ROOT)
    _run_child(code, {"FA_PASS_FRAGMENTS": "16"})                                   # ~65 fragments per genome: 5 parts each
    _run_child(code, {"FA_EVENTS_CAP_MAX": "60000", "FA_EVENTS_CAP_MIN": "1000"})   # ~4000 events per fragment
    _run_child(code, {"FA_EVENTS_CAP_MAX": "9000", "FA_PASS_FRAGMENTS": "50"})      # parts of one or two fragments


def test_long_locus_and_reference_exceptions():
    # a tandem array in the reference merges dozens of overlapping candidates into one locus whose event stream is far
    # longer than the 2048 events staged in LDS; N runs and IUPAC codes in the reference take the byte path of K1
    g = syn.rng(97)
    unit = syn.random_codes(g, 1000)
    flank = syn.random_codes(g, 40_000)
    ref = np.concatenate([flank[:20_000]] + [syn.mutate_codes(g, unit, 0.01) for _ in range(40)] + [flank[20_000:]])
    ref_ascii = bytearray(bytes(syn.to_ascii(ref)))
    ref_ascii[5_000:5_060] = b"N" * 60
    ref_ascii[30_500:30_503] = b"RYK"
    ref_ascii[2048 + 23] = ord("n")                          # right at a tile boundary
    other = bytes(syn.to_ascii(syn.mutate_codes(g, ref, 0.06)))
    query = np.concatenate([flank[18_000:20_000]] + [unit] * 4 + [flank[20_000:22_000]] + [unit] * 3)
    mapper, hits, ohits, det = run_both({}, [[bytes(ref_ascii)], [other]], [syn.to_ascii(query)])
    ms = (C.c_float * 16)()
    lib.fa_mapper_last_timings(mapper._h, ms, 16)
    assert ms[7] / max(ms[6], 1) > 2048, "expected loci with more than 2048 slide events"
    assert gpu_mappings(mapper) == oracle_mappings(det) and len(ohits) == 2
    assert hit_tuples(hits) == ohits


def test_degenerate_and_unusual_parameter_cells():
    g = syn.rng(98)
    anc = syn.random_codes(g, 60_000)
    refs = [[syn.to_ascii(syn.mutate_codes(g, anc, d))] for d in (0.01, 0.06)]
    query = [syn.to_ascii(syn.mutate_codes(g, anc, 0.03))]
    # (k=21, fragment_length=1000): recommendedWindowSize returns the fragment length itself, no fragment holds a full
    # window, nothing maps -- must be handled without error on both sides
    mapper, hits, ohits, det = run_both({"k": 21, "fragment_length": 1000}, refs, query)
    assert mapper.window_size == 1000 and hits == [] and ohits == []
    # short fragments and a small k
    for params in ({"k": 11, "fragment_length": 500}, {"k": 16, "fragment_length": 333}, {"k": 24, "fragment_length": 2000}):
        mapper, hits, ohits, det = run_both(params, refs, query)
        assert gpu_mappings(mapper) == oracle_mappings(det), params
        assert hit_tuples(hits) == ohits, params
    with pytest.raises(NotImplementedError):
        quiet_sketch(pf.Sketch, fragment_length=20).add_genome("r", refs[0][0]).index().query_genome(query[0])


@pytest.mark.parametrize("k", [14, 16, 21])
@pytest.mark.parametrize("frag", [1000, 3000, 5000])
def test_config5_cells_all_vs_all(k, frag):
    """BASELINE config 5 scaled down: every (k, fragment_length) cell, all-vs-all through a resident batch, against the oracle."""
    g = syn.rng(4000 + k * 10 + frag // 1000)
    genomes = []
    for fam in range(2):
        anc = syn.random_codes(g, 120_000)
        for d in (0.0, 0.02, 0.07, 0.15):
            genomes.append([syn.to_ascii(syn.mutate_codes(g, anc, d) if d else anc)])
    params = {"k": k, "fragment_length": frag}
    sk, osk = quiet_sketch(pf.Sketch, **params), OracleSketch(**params)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i, c in enumerate(genomes):
            sk.add_draft(i, c)
            osk.add_draft(i, c)
        mapper = sk.index()
        osk.index()
        assert mapper.window_size == osk.window_size
        got = [hit_tuples(h) for h in mapper.upload_genomes(genomes).query()]
    want = [osk.query_draft(c, threads=8) for c in genomes]
    assert got == want
    if mapper.window_size < frag:
        # (not exactly 100.0: a fragment that ends at the contig end can miss one minimizer, in the oracle too)
        assert all(any(n == i and ident >= 99.99 for n, ident, m, f in w) for i, w in enumerate(want))
    else:
        assert all(w == [] for w in want)       # (21, 1000): no window fits a fragment


def test_protein_small_k_and_wide_strings(golden_dir):
    b1 = read_fasta(os.path.join(golden_dir, "BGC0001425.faa"))
    b2 = read_fasta(os.path.join(golden_dir, "BGC0001427.faa"))
    b3 = read_fasta(os.path.join(golden_dir, "BGC0001428.faa"))
    for k, frag in ((5, 60), (9, 150)):
        sk, osk = pf.Sketch(k=k, fragment_length=frag, protein=True), OracleSketch(k=k, fragment_length=frag, protein=True)
        for name, prots in (("a", b1), ("b", b2)):
            sk.add_draft(name, prots)
            osk.add_draft(name, prots)
        m = sk.index()
        osk.index()
        assert len(m.lookup_index) == osk.index_size
        assert hit_tuples(m.query_draft(b3)) == osk.query_draft(b3)
    # UCS2 / UCS4 str carriers read the same characters (_fastani.pyx:144-148)
    g = syn.rng(99)
    ref = bytes(syn.to_ascii(syn.random_codes(g, 30_000))).decode()
    sk = pf.Sketch()
    sk.add_genome("r", ref)
    m = sk.index()
    want = hit_tuples(m.query_genome(ref))
    assert want == [("r", 100.0, 10, 10)]
    wide = ref[:12_000] + "\u0394" + ref[12_001:]          # one non-Latin-1 character forces the UCS2 representation
    wide4 = ref[:12_000] + "\U0001F9EC" + ref[12_001:]     # UCS4
    for q in (wide, wide4):
        got = m.query_genome(q)
        assert len(got) == 1 and got[0].matches >= 9


@pytest.mark.parametrize("fragment_length", [513, 514, 515, 516, 520])
def test_sketch_sizes_around_the_16_bit_event_limit(fragment_length):
    # protein mode, k = 5, w = 1: a fragment keeps fragment_length - 4 minimizers (5-mers of random residues hardly
    # repeat), i.e. sketches of 509 .. 516 entries: the 16-bit slide event holds slots up to 511 (sketches up to 510),
    # beyond that the pass must switch to 32-bit events
    g = syn.rng(1200 + fragment_length)
    amino = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    prots = [bytes(amino[g.integers(0, 20, 9_000)]) for _ in range(3)]
    params = dict(k=5, fragment_length=fragment_length, protein=True, minimum_fraction=0.0)
    query = []
    for p in prots:
        a = np.frombuffer(p, dtype=np.uint8).copy()
        m = g.random(len(a)) < 0.02
        a[m] = amino[g.integers(0, 20, int(m.sum()))]
        query.append(bytes(a))
    mapper, hits, ohits, det = run_both(params, [prots, [prots[1], prots[0]]], query, threads=8)
    sizes = set(det["mappings"]["sketch"].tolist())
    assert len(det["mappings"]["rseq"]) > 20 and fragment_length - 8 <= min(sizes) and max(sizes) <= fragment_length - 4
    assert gpu_mappings(mapper) == oracle_mappings(det)
    assert hit_tuples(hits) == ohits and ohits


def test_huge_sketch_with_seed_overflow():
    # protein mode (w = 1) with 12 000-residue fragments: a fragment keeps ~12 000 minimizers, far beyond what the LDS
    # tables of the chunked L1 kernel are laid out for, and four copies of the reference give it ~48 000 seed hits, more
    # than the LDS merge holds: the pass must take the HBM sort of k_l1 (k_l1_big stands aside) and still agree
    g = syn.rng(98)
    amino = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    prots = [bytes(amino[g.integers(0, 20, 13_000)]) for _ in range(3)]
    params = dict(k=7, fragment_length=12_000, protein=True, minimum_fraction=0.0)
    refs = [prots for _ in range(4)]
    query = []
    for p in prots:
        a = np.frombuffer(p, dtype=np.uint8).copy()
        m = g.random(len(a)) < 0.03
        a[m] = amino[g.integers(0, 20, int(m.sum()))]
        query.append(bytes(a))
    mapper, hits, ohits, det = run_both(params, refs, query, threads=8)
    assert len(ohits) == 4 and ohits[0][2] == 3
    assert gpu_mappings(mapper) == oracle_mappings(det)
    assert hit_tuples(hits) == ohits


def test_sharding_all_vs_all_single_rank():
    from pyfastani_amd import sharding
    g = syn.rng(100)
    anc = syn.random_codes(g, 90_000)
    genomes = [[syn.to_ascii(syn.mutate_codes(g, anc, d))] for d in (0.0, 0.02, 0.05)] + [[syn.to_ascii(syn.random_codes(g, 90_000))]]
    sk = pf.Sketch()
    for i, c in enumerate(genomes):
        sk.add_draft(i, c)
    mapper = sk.index()
    rows = sharding.all_vs_all(mapper, genomes, rank=0, world_size=1)
    pairs = {(int(r["query_id"]), int(r["ref_genome_id"])) for r in rows if r["count_seq"] >= 6}
    assert pairs == {(a, b) for a in range(3) for b in range(3)} | {(3, 3)}
    # the strided shard of a 2-rank job sees exactly its own queries
    rows0 = sharding.all_vs_all(mapper, genomes, rank=0, world_size=1)
    assert rows0.tobytes() == rows.tobytes()
    owned = sharding.shard_indices(len(genomes), 1, 2)
    batch = mapper.upload_genomes([genomes[i] for i in owned])
    part = sharding.remap_query_ids(batch.query_rows(), owned)
    assert sorted(map(tuple, part.tolist())) == sorted(t for t in map(tuple, rows.tolist()) if t[0] in owned)


def _sharded_index_genomes():
    g = syn.rng(123)
    genomes = []
    for fam in range(2):
        anc = syn.random_codes(g, 90_000)
        for d in (0.0, 0.03, 0.1):
            genomes.append(syn.split_contigs(g, syn.to_ascii(syn.mutate_codes(g, anc, d) if d else anc), 4))
    genomes.append([b"ACGT" * 3])                          # a genome whose only contig is too short: no records
    genomes.insert(2, [syn.to_ascii(syn.random_codes(g, 50_000)), b"ACGTACGT"])
    return genomes


def test_sharded_index_build_single_rank_and_device_merge():
    """build_index_sharded at world size 1 goes through the HBM-to-HBM record export / import and must give the index a
    single Sketch builds; merge_record_shards is run on device tensors against the same records dealt to 3 ranks."""
    import torch
    from pyfastani_amd import sharding
    genomes = _sharded_index_genomes()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sk = pf.Sketch()
        for i, c in enumerate(genomes):
            sk.add_draft(f"g{i}", c)
        want_min = tuple(a.copy() for a in sk._read_minimizers())
        direct = sk.index()
        sharded = sharding.build_index_sharded(genomes, names=[f"g{i}" for i in range(len(genomes))], rank=0, world_size=1, device="cuda")
        assert sharded.occurences_threshold == direct.occurences_threshold and len(sharded.lookup_index) == len(direct.lookup_index)
        got_min = sharded._read_minimizers()
        assert all(np.array_equal(a, b) for a, b in zip(got_min, want_min))
        for q in (genomes[1], genomes[4]):
            assert hit_tuples(sharded.query_draft(q)) == hit_tuples(direct.query_draft(q))
        # the same records dealt round-robin to 3 "ranks", merged on the device
        world, n = 3, len(genomes)
        shards, rec_off, ctg = [], [], []
        for r in range(world):
            local = pf.Sketch()
            for i in sharding.shard_indices(n, r, world):
                local.add_draft(i, genomes[i])
            rec, (lengths, sbf, counter) = local._export_records("cuda")
            sbf64 = np.asarray(sbf, np.int64)
            c = np.diff(np.concatenate([[0], sbf64]))
            first = torch.as_tensor((sbf64 - c).astype(np.int32), device="cuda")
            off = torch.searchsorted(rec[1].contiguous(), first).to(torch.int64).cpu()
            rec_off.append(torch.cat([off, torch.tensor([rec.shape[1]])]))
            ctg.append(torch.as_tensor(c))
            shards.append(rec)
        n_max = max(int(s.shape[1]) for s in shards)
        gathered = torch.zeros((world, 3, n_max), dtype=torch.int32, device="cuda")
        for r, s in enumerate(shards):
            gathered[r, :, : s.shape[1]] = s
        merged, sbf = sharding.merge_record_shards(gathered, rec_off, ctg)
        m = merged.cpu().numpy()
        assert np.array_equal(m[0].view(np.uint32), want_min[0]) and np.array_equal(m[1], want_min[1]) and np.array_equal(m[2], want_min[2])


def test_sharded_index_build_two_ranks_one_gpu(tmp_path):
    """Two gloo ranks share GPU 0: each sketches half of the references, the shards are exchanged (through host tensors,
    gloo has no device all-gather) and both ranks must end up with the index of a single Sketch."""
    import socket
    import subprocess
    import textwrap
    code = textwrap.dedent("""
        import os, sys, warnings
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np, torch.distributed as dist
        import pyfastani_amd as pf
        from pyfastani_amd import sharding
        from test_gpu_parity import _sharded_index_genomes
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        genomes = _sharded_index_genomes()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sk = pf.Sketch()
            for i, c in enumerate(genomes):
                sk.add_draft(i, c)
            want = tuple(a.copy() for a in sk._read_minimizers())
            direct = sk.index()
            m = sharding.build_index_sharded(genomes, rank=rank, world_size=world, device="cpu")
            assert all(np.array_equal(a, b) for a, b in zip(m._read_minimizers(), want))
            assert m.occurences_threshold == direct.occurences_threshold
            tup = lambda hits: [(h.name, h.identity, h.matches, h.fragments) for h in hits]
            assert tup(m.query_draft(genomes[1])) == tup(direct.query_draft(genomes[1]))
        dist.barrier(); dist.destroy_process_group()
        open(os.path.join(%r, f"idx{rank}.ok"), "w").write("OK")
    """ % (ROOT, os.path.join(ROOT, "tests"), str(tmp_path)))
    script = tmp_path / "worker.py"
    script.write_text(code)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr
    assert (tmp_path / "idx0.ok").exists() and (tmp_path / "idx1.ok").exists()


def _ref_sharded_case():
    """Six 1 Mb references; two planted 45-mers sit in genomes 0 AND 1 -- with two shards (0,2,4 / 1,3,5) their hashes
    occur ~150 times per shard and ~300 times over the whole index, so whether they are ignored depends on the sums."""
    g = syn.rng(53)
    n = 1_000_000
    genomes = [syn.random_codes(g, n) for _ in range(6)]
    r1, r2 = syn.random_codes(g, 45), syn.random_codes(g, 45)
    for m in genomes[:2]:
        for p in range(1000, n - 1000, n // 150):
            m[p: p + 45] = r1
        for p in range(2500, n - 1000, n // 75):
            m[p: p + 45] = r2
    refs = [[syn.to_ascii(x)] for x in genomes]
    queries = [[syn.to_ascii(syn.mutate_codes(g, genomes[0], 0.02))], [syn.to_ascii(syn.mutate_codes(g, genomes[3], 0.04))],
               [syn.to_ascii(syn.mutate_codes(g, genomes[1], 0.01))]]
    return refs, queries


@pytest.mark.parametrize("world", [2, 8])
def test_reference_sharded_index_matches_single_index(world):
    """SURVEY.md 8e, alternative partitioning: every shard indexes a part of the references, the frequency threshold is
    taken over the position lists of all shards (`sharding.merged_frequency`) and installed in every shard; the rows of
    the shards together must be the rows of the single index.  Both shards live in this one process here."""
    import torch
    from pyfastani_amd import sharding
    refs, queries = _ref_sharded_case()            # six references: with eight shards two of them are empty
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sk = pf.Sketch()
        for i, r in enumerate(refs):
            sk.add_draft(i, r)
        single = sk.index()
        want = np.concatenate([single.upload_genomes(queries).query_rows(0, len(queries))])
        shards = []
        for rank in range(world):
            owned = sharding.shard_indices(len(refs), rank, world)
            loc = pf.Sketch()
            for i in owned:
                loc.add_draft(i, refs[i])
            shards.append((loc.index(), owned))
    exported = [m._export_lookup("cuda") for m, _ in shards]
    local_thr = [m.occurences_threshold for m, _ in shards]
    k = torch.cat([a.to(torch.int64) & 0xFFFFFFFF for a, _ in exported])
    c = torch.cat([b.to(torch.int64) for _, b in exported])
    thr, drop = sharding.merged_frequency(k, c)
    assert thr == single.occurences_threshold and thr < 2**31 - 1 and drop.numel() > 0
    assert any(t != thr for t in local_thr)            # the shards on their own would filter differently
    # a single shard run through global_frequency (world 1) finds its own threshold again
    assert sharding.global_frequency(*exported[0])[0] == local_thr[0]
    got = []
    for m, owned in shards:
        m._set_global_frequency(thr, drop)
        assert m.occurences_threshold == thr
        got.append(sharding.query_ref_sharded(m, owned, queries))
    got = np.concatenate(got)
    got = got[np.lexsort((got["ref_genome_id"], got["query_id"]))]
    want = want[np.lexsort((want["ref_genome_id"], want["query_id"]))]
    assert len(want) >= 3 and got.tobytes() == want.tobytes()


def test_reference_sharded_two_ranks_one_gpu(tmp_path):
    """The same through `build_ref_sharded_mapper` / `query_ref_sharded` on two gloo ranks that share GPU 0."""
    import socket
    import subprocess
    import textwrap
    code = textwrap.dedent("""
        import os, sys, warnings
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np, torch.distributed as dist
        import pyfastani_amd as pf
        from pyfastani_amd import sharding
        from test_gpu_parity import _ref_sharded_case
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        refs, queries = _ref_sharded_case()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sk = pf.Sketch()
            for i, r in enumerate(refs):
                sk.add_draft(i, r)
            single = sk.index()
            want = single.upload_genomes(queries).query_rows(0, len(queries))
            m, owned = sharding.build_ref_sharded_mapper(refs, rank=rank, world_size=world, device="cpu")
            got = sharding.query_ref_sharded(m, owned, queries, world_size=world, device="cpu")
        want = want[np.lexsort((want["ref_genome_id"], want["query_id"]))]
        assert m.occurences_threshold == single.occurences_threshold
        assert got.tobytes() == want.tobytes(), (rank, len(got), len(want))
        dist.barrier(); dist.destroy_process_group()
        open(os.path.join(%r, f"refshard{rank}.ok"), "w").write("OK")
    """ % (ROOT, os.path.join(ROOT, "tests"), str(tmp_path)))
    script = tmp_path / "worker.py"
    script.write_text(code)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr
    assert (tmp_path / "refshard0.ok").exists() and (tmp_path / "refshard1.ok").exists()


def test_fasta_ingest_matches_python_path(tmp_path):
    """Sketch.add_fasta / Mapper.upload_fasta (native parse + pack) against the same records fed through add_draft /
    query_draft: lower case, N runs, wrapped lines, a short contig, CRLF-free files."""
    from pyfastani_amd._fasta import Parser
    g = syn.rng(321)
    anc = syn.random_codes(g, 160_000)

    def write_fasta(path, contigs, width):
        with open(path, "wb") as f:
            for i, c in enumerate(contigs):
                f.write(b">contig_%d some text\n" % i)
                b = bytes(c)
                for j in range(0, len(b), width):
                    f.write(b[j:j + width] + b"\n")

    files = []
    for gi, d in enumerate((0.0, 0.04, 0.11)):
        seq = bytearray(bytes(syn.to_ascii(syn.mutate_codes(g, anc, d) if d else anc)))
        seq[1000:1100] = b"N" * 100
        seq[50_000:51_000] = bytes(seq[50_000:51_000]).lower()
        contigs = syn.split_contigs(g, np.frombuffer(bytes(seq), np.uint8), 5) + [b"ACGTAC"]
        path = str(tmp_path / f"g{gi}.fna")
        write_fasta(path, contigs, 60 + 10 * gi)
        files.append(path)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        native, plain = pf.Sketch(), pf.Sketch()
        for i, path in enumerate(files):
            native.add_fasta(i, path)
            plain.add_draft(i, [r.seq for r in Parser(path)])
        a, b = native._read_minimizers(), plain._read_minimizers()
        assert all(np.array_equal(x, y) for x, y in zip(a, b)) and len(a[0]) > 10_000
        m1, m2 = native.index(), plain.index()
        got = [hit_tuples(h) for h in m1.upload_fasta(files).query()]
        want = [hit_tuples(m2.query_draft([r.seq for r in Parser(path)])) for path in files]
    assert got == want and all(len(w) == 3 for w in want)
    # ... and against the ORACLE fed by a plain-Python reader of the same files (no native code on that side): the records
    # the native ingest produces, the sketch built from them and the hits of every file used as a query
    osk = OracleSketch()
    for i, path in enumerate(files):
        osk.add_draft(i, read_fasta(path))
    for x, y in zip(a, osk.minimizers()):
        assert np.array_equal(x, y)
    osk.index()
    for path, hits in zip(files, got):
        assert hits == osk.query_draft(read_fasta(path), threads=2)
    assert [r.seq for r in Parser(files[0])] == [c.upper().encode() for c in read_fasta(files[0])]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert hit_tuples(m1.query_fasta(files[1])) == want[1]
    with pytest.warns(UserWarning):
        pf.Sketch().add_fasta("x", files[0])                 # the 6-base contig is reported like add_draft does
    with pytest.raises(OSError):
        pf.Sketch().add_fasta("x", str(tmp_path / "missing.fna"))
    # round 5: every file read + packed by its own host task in ONE sweep (fa_sketch_add_fasta_many), and the streamed form of
    # the query side (chunks of files into two recycled batches while the previous chunk maps) -- the same records, sketch and hits
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        many = pf.Sketch().add_fasta_many(range(len(files)), files)
        assert all(np.array_equal(x, y) for x, y in zip(many._read_minimizers(), a))
        assert many.names == [0, 1, 2]
        stats = {}
        streamed_refs = pf.Sketch().add_fasta_stream(range(len(files)), files, chunk=1, stats=stats)   # the device sketches chunk c while c + 1 is read
        assert all(np.array_equal(x, y) for x, y in zip(streamed_refs._read_minimizers(), a)) and stats["chunks"] == 3 and stats["sketch_s"] > 0
        m3 = many.index()
        for chunk in (1, 2, 5, None):                       # (None: chunks sized by the files' bytes)
            streamed = {}
            for first, hits in m3.query_fasta_stream(files * 2, chunk=chunk):
                for i, h in enumerate(hits):
                    streamed[first + i] = hit_tuples(h)
            assert [streamed[i] for i in range(2 * len(files))] == want * 2
        rows = [r for _, r in m3.query_fasta_stream(files, chunk=2, rows=True)]
        assert sum(len(r) for r in rows) == 9
        # ... and with every file read ONCE: a PackedGenomes as the references and as the query stream
        packed = pf.PackedGenomes(files)
        once = pf.Sketch().add_packed(range(len(files)), packed)
        assert all(np.array_equal(x, y) for x, y in zip(once._read_minimizers(), a))
        kept = pf.PackedGenomes([])
        piped = pf.Sketch().add_fasta_stream(range(len(files)), files, chunk=2, keep=kept)      # read once, kept, sketched behind the reader
        assert len(kept) == len(files) and all(np.array_equal(x, y) for x, y in zip(piped._read_minimizers(), a))
        m4 = once.index()
        for chunk in (1, 2, None):
            got4 = {}
            for first, hits in m4.query_fasta_stream(packed if chunk != 2 else kept, chunk=chunk):
                for i, h in enumerate(hits):
                    got4[first + i] = hit_tuples(h)
            assert [got4[i] for i in range(len(files))] == want
    with pytest.warns(UserWarning):
        pf.Sketch().add_fasta_many(["x"], files[:1])
    with pytest.raises(OSError):
        pf.Sketch().add_fasta_many(["x", "y"], [files[0], str(tmp_path / "missing.fna")])
    with pytest.raises(OSError):
        list(m3.query_fasta_stream([files[0], str(tmp_path / "missing.fna")], chunk=1))
    # a protein file through the packed reader (bytes kept, upper-cased)
    prot = str(tmp_path / "p.faa")
    with open(prot, "wb") as f:
        f.write(b">p1\nMKVlaa\nGG\n>p2\n" + b"ACDEFGHIKLMNPQRSTVWY" * 40 + b"\n")
    ps, pp = pf.Sketch(protein=True, fragment_length=100), pf.Sketch(protein=True, fragment_length=100)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ps.add_fasta_many(["p"], [prot])
        pp.add_draft("p", [r.seq for r in Parser(prot)])
    assert all(np.array_equal(x, y) for x, y in zip(ps._read_minimizers(), pp._read_minimizers()))


def test_fastani_style_outputs_from_device_rows(tmp_path):
    """the identity matrix / hit list of an all-vs-all run built from the rows the DEVICE returns, against
    the same outputs built from the ORACLE's rows (per-pair CGI results of every query)."""
    from pyfastani_amd import outputs
    from pyfastani_amd._batch import ROW_DTYPE
    g = syn.rng(654)
    genomes, names = [], []
    for fam in range(2):
        anc = syn.random_codes(g, 150_000)
        for d in (0.0, 0.03, 0.08, 0.14):
            genomes.append([syn.to_ascii(syn.mutate_codes(g, anc, d) if d else anc)])
            names.append(f"f{fam}_d{d}")
    sk, osk = pf.Sketch(), OracleSketch()
    for n, c in zip(names, genomes):
        sk.add_draft(n, c)
        osk.add_draft(n, c)
    mapper = sk.index()
    osk.index()
    rows = mapper.upload_genomes(genomes).query_rows()
    want = []
    for q, c in enumerate(genomes):
        _, det = osk.query_draft(c, threads=2, details=True)
        r = det["rows"]
        want += [(q, int(gi), int(cnt), int(det["total_fragments"]), float(ident)) for gi, ident, cnt in zip(r["genome"], r["identity"], r["count"])]
    want = np.array(want, dtype=ROW_DTYPE)
    assert rows.tobytes() == want.tobytes()
    lengths = [sum((len(x) // 3000) * 3000 for x in c) for c in genomes]
    qlen = [sum(len(x) for x in c) for c in genomes]
    kept, okept = outputs.filter_rows(rows, qlen, lengths, 3000, 0.2), outputs.filter_rows(want, qlen, lengths, 3000, 0.2)
    m, om = outputs.identity_matrix(kept, 8, 8, symmetric=True), outputs.identity_matrix(okept, 8, 8, symmetric=True)
    assert np.array_equal(np.isnan(m), np.isnan(om)) and np.array_equal(m[~np.isnan(m)], om[~np.isnan(om)])
    # (a self mapping is 100.0 up to the end-of-contig effect the oracle shows too: the slide stops when the last record of
    # the contig is admitted, so the fragment that ends exactly at the contig end can miss one minimizer)
    assert np.all(np.diag(m) >= 99.999) and np.isnan(m[0, 4]) and m[0, 1] > 95.0     # families do not mix
    outputs.write_matrix(str(tmp_path / "gpu.matrix"), names, m)
    outputs.write_matrix(str(tmp_path / "cpu.matrix"), names, om)
    outputs.write_hits(str(tmp_path / "gpu.tsv"), names, names, kept)
    outputs.write_hits(str(tmp_path / "cpu.tsv"), names, names, okept)
    assert (tmp_path / "gpu.matrix").read_text() == (tmp_path / "cpu.matrix").read_text()
    assert (tmp_path / "gpu.tsv").read_text() == (tmp_path / "cpu.tsv").read_text()
    # the hit list a user gets per query is the filtered table, best identity first
    per_query = [hit_tuples(h) for h in mapper.upload_genomes(genomes).query()]
    for q in range(8):
        mine = kept[kept["query_id"] == q]
        assert sorted((names[r["ref_genome_id"]], float(r["identity"])) for r in mine) == sorted((n, i) for n, i, _, _ in per_query[q])


def test_concurrent_queries_on_one_mapper():
    """Mapper.query_draft is re-entrant (_fastani.pyx:1158-1161): calls from several host threads run on separate
    workspaces / streams of the same mapper and must return what serial calls return."""
    import threading
    g = syn.rng(777)
    anc = syn.random_codes(g, 400_000)
    refs = [syn.to_ascii(syn.mutate_codes(g, anc, d)) for d in (0.0, 0.02, 0.06, 0.11)] + [syn.to_ascii(syn.random_codes(g, 200_000))]
    sk = pf.Sketch()
    for i, r in enumerate(refs):
        sk.add_genome(i, r)
    mapper = sk.index()
    queries = [syn.to_ascii(syn.mutate_codes(g, anc, d)) for d in (0.01, 0.03, 0.05, 0.08, 0.12, 0.16)]
    want = [hit_tuples(mapper.query_genome(q)) for q in queries]
    got = [[None] * 4 for _ in queries]
    errors = []

    def worker(qi):
        try:
            for rep in range(4):
                got[qi][rep] = hit_tuples(mapper.query_genome(queries[qi]))
        except Exception as e:            # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(queries))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for qi in range(len(queries)):
        assert all(r == want[qi] for r in got[qi]), qi
    # a resident batch queried from two threads at once
    batch = mapper.upload_genomes([[q] for q in queries])
    out = [None, None]

    def half(i):
        out[i] = [hit_tuples(h) for h in batch.query(3 * i, 3)]

    ts = [threading.Thread(target=half, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert out[0] + out[1] == want


@pytest.mark.parametrize("lanes,part", [(3, 40), (2, 0)])
def test_parts_pipelined_over_lanes_match(lanes, part):
    # FA_QUERY_LANES > 1 runs the parts of a pass on sub-workspaces with their own streams (run_query_pass); with tiny
    # buffers the parts are also declared void and repeated while other parts are in flight: same rows, same mappings
    import subprocess
    import textwrap
    code = textwrap.dedent("""
        import sys, os, json
        sys.path.insert(0, %r)
        sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np
        import pyfastani_amd as pf
        from pyfastani_amd import synthetic as syn
        import test_gpu_parity as T
        g = syn.rng(3131)
        anc = syn.random_codes(g, 900_000)
        refs = [[syn.to_ascii(syn.mutate_codes(g, anc, d))] for d in (0.01, 0.05, 0.1, 0.15)]
        q = [syn.to_ascii(syn.mutate_codes(g, anc, 0.03))]
        mapper, hits, ohits, det = T.run_both({}, refs, q, threads=8)
        if not os.environ.get("FA_PASS_FRAGMENTS"):      # (the stage getters keep the last part of every lane only)
            assert T.gpu_mappings(mapper) == T.oracle_mappings(det), "mappings"
        assert T.hit_tuples(hits) == ohits and len(ohits) == 4, "hits"
        ms = (T.C.c_float * 16)(); T.lib.fa_mapper_last_timings(mapper._h, ms, 16)
        print(json.dumps({"retries": ms[9], "loci": ms[6]}))
    """) % (ROOT, ROOT)
    env = dict(os.environ, FA_QUERY_LANES=str(lanes), FA_LANE_MIN_FRAGMENTS="64")
    if part:
        env.update(FA_PASS_FRAGMENTS=str(part), FA_LOCI_CAP_MIN="64", FA_EVENTS_CAP_MIN="4096")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["loci"] > 500 and (out["retries"] > 0 if part else True), out


@pytest.mark.parametrize("part", [0, 700])
def test_workgroup_order_of_multi_genome_passes_changes_nothing(part):
    # Passes that hold several genomes run k_l2_events in offset-major, XCD-aware workgroup order (build_frag_order): a
    # permutation of the work.  Uneven drafts (different fragment counts, short and empty genomes), with and without the pass
    # cut into parts in the middle of a genome: the rows must equal the query-major run (FA_FRAG_ORDER=0) byte for byte
    # AND the oracle's hits per genome.
    import subprocess
    import textwrap
    code = textwrap.dedent("""
        import sys, os, json, hashlib, warnings
        sys.path.insert(0, %r)
        import numpy as np
        import pyfastani_amd as pf
        from pyfastani_amd import synthetic as syn
        from oracle.oracle import OracleSketch
        g = syn.rng(777)
        genomes = []
        for fam in range(3):
            anc = syn.random_codes(g, 260_000 + 90_000 * fam)
            for d in (0.0, 0.02, 0.06, 0.12):
                genomes.append(syn.split_contigs(g, syn.to_ascii(syn.mutate_codes(g, anc, d)), 4 + fam))
        genomes.insert(5, [b"ACGT" * 3]); genomes.insert(9, [])
        sk, osk = pf.Sketch(), OracleSketch()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i, c in enumerate(genomes):
                sk.add_draft(i, c); osk.add_draft(i, c)
            mapper = sk.index(); osk.index()
            batch = mapper.upload_genomes(genomes)
            rows = batch.query_rows(0, len(genomes))
            hits = [[(h.name, h.identity, h.matches, h.fragments) for h in hs] for hs in batch.query()]
            want = [osk.query_draft(c, threads=8) for c in genomes]
        assert hits == want, "hits differ from the oracle"
        print(json.dumps({"rows": len(rows), "sha": hashlib.sha256(rows.tobytes()).hexdigest()}))
    """) % (ROOT,)
    outs = []
    for order in ("1", "0"):
        env = dict(os.environ, FA_FRAG_ORDER=order)
        if part:
            env.update(FA_PASS_FRAGMENTS=str(part))
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        outs.append(json.loads(res.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1] and outs[0]["rows"] > 40, outs


def test_stage_times_by_stamps_agree_with_hip_events():
    # the stage times of a pass come from device stamps of the 100 MHz counter (fa_mapper_last_timings [0..4]); on request
    # the L2 stage is bracketed by two HIP events on the library's stream as well ([16]): the two clocks must agree
    g = syn.rng(4242)
    anc = syn.random_codes(g, 1_500_000)
    sk = pf.Sketch()
    for i, d in enumerate((0.01, 0.04, 0.08, 0.12)):
        sk.add_genome(i, syn.to_ascii(syn.mutate_codes(g, anc, d)))
    mapper = sk.index()
    q = syn.to_ascii(syn.mutate_codes(g, anc, 0.03))
    mapper.query_genome(q)
    check(lib.fa_mapper_set_stage_events(mapper._h, 1))
    try:
        l2, ev, total, parts = [], [], [], []
        for _ in range(5):
            assert len(mapper.query_genome(q)) == 4
            ms = (C.c_float * 24)()
            check(lib.fa_mapper_last_timings(mapper._h, ms, 24))
            l2.append(ms[2]); ev.append(ms[16]); total.append(ms[4]); parts.append(sum(ms[0:4]))
    finally:
        check(lib.fa_mapper_set_stage_events(mapper._h, 0))
    assert min(l2) > 0.01 and min(ev) > 0.01, (l2, ev)
    assert abs(np.median(l2) - np.median(ev)) <= 0.15 * np.median(l2) + 0.01, (l2, ev)       # ms
    assert abs(np.median(total) - np.median(parts)) <= 0.02 * np.median(total) + 0.005, (total, parts)
    ms = (C.c_float * 24)()
    mapper.query_genome(q)
    check(lib.fa_mapper_last_timings(mapper._h, ms, 24))
    assert ms[16] == 0.0                                                                       # off again


def test_device_memory_is_stable():
    """Repeated queries, resident batches and mapper life cycles must not grow the device allocation (scripts/check_leaks.py)."""
    import subprocess
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_leaks.py")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["after_600_queries_mb"] <= 1.0 and out["after_50_batches_mb"] <= 1.0 and out["after_20_mappers_mb"] <= 16.0, out


@pytest.mark.parametrize("params,env", [
    ({}, {"FA_NO_PACKED_GEO": "1"}),                       # default cell, 16-bit events, record geometry from the plain arrays
    ({"fragment_length": 9000}, {}),                       # cmw + 1 >= 2^13: no packed geometry; sketches beyond 510: 32-bit events
    ({"fragment_length": 20_000, "k": 14}, {}),
])
def test_unpacked_record_geometry(params, env, monkeypatch):
    # k_l2_events<T, false>: the form every index takes whose fragment length leaves no room for the 13-bit distances of
    # rec_hg (fa_engine.hip, packed_geo), and FA_NO_PACKED_GEO forces on any index
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    g = syn.rng(133)
    anc = syn.random_codes(g, 400_000)
    refs = [[syn.to_ascii(syn.mutate_codes(g, anc, d))] for d in (0.01, 0.05, 0.12)]
    inv = syn.mutate_codes(g, anc, 0.03)
    refs.append([syn.to_ascii(np.concatenate([inv[:150_000], syn.reverse_complement_codes(inv[150_000:300_000]), inv[300_000:]]))])
    refs.append([syn.to_ascii(syn.random_codes(g, 200_000))])
    query = syn.split_contigs(g, syn.to_ascii(syn.mutate_codes(g, anc, 0.04)), 3)
    mapper, hits, ohits, det = run_both(params, refs, query, threads=4)
    assert len(ohits) >= 3
    assert gpu_mappings(mapper) == oracle_mappings(det)
    assert hit_tuples(hits) == ohits


@pytest.mark.parametrize("bits,length,contigs", [(14, 300_000, 1), (13, 120_000, 7), (16, 400_000, 3)])
def test_global_coordinate_across_word_boundaries(bits, length, contigs, monkeypatch):
    # k_l1 tests "same contig and wb - wa < fragment_length" on a padded global coordinate whose low word is gathered per hit
    # and whose high word is the number of word boundaries before the record (exact at any index size; BASELINE config 3 spans
    # 5 x 10^9 padded bases: one boundary).  FA_GPOS_BITS shrinks the low word to 2^13 - 2^16 bases, so that these small indexes
    # cross dozens of boundaries, candidates and loci straddle them, and the 64-bit form of the candidate scan runs: every
    # mapping and hit must still be the oracle's.
    monkeypatch.setenv("FA_GPOS_BITS", str(bits))
    g = syn.rng(900 + bits)
    anc = syn.random_codes(g, length)
    split = (lambda seq: syn.split_contigs(g, seq, contigs)) if contigs > 1 else (lambda seq: [seq])
    refs = [split(syn.to_ascii(syn.mutate_codes(g, anc, d))) for d in (0.0, 0.02, 0.06, 0.11)]
    refs.append(split(syn.to_ascii(syn.random_codes(g, length))))
    dup = syn.mutate_codes(g, anc, 0.03)
    dup = np.concatenate([dup[: length // 2], dup[length // 4: length // 2], syn.reverse_complement_codes(dup[length // 2:])])
    refs.append(split(syn.to_ascii(dup)))
    query = split(syn.to_ascii(syn.mutate_codes(g, anc, 0.04)))
    mapper, hits, ohits, det = run_both({}, refs, query, threads=4)
    assert gpu_mappings(mapper) == oracle_mappings(det) and len(oracle_mappings(det)) > 100
    assert hit_tuples(hits) == ohits and len(ohits) >= 4
    # the same index with the full 32-bit word (no boundary at this size) gives the same rows
    monkeypatch.delenv("FA_GPOS_BITS")
    mapper32, hits32, _, _ = run_both({}, refs, query, threads=4)
    assert hit_tuples(hits32) == ohits and gpu_mappings(mapper32) == gpu_mappings(mapper)


def test_fused_query_sketch_overflow_falls_back(monkeypatch):
    # k_query_fused (K1 + per-fragment sketch in one launch) holds a fragment's records in LDS; one with more than its
    # capacity voids the pass, which then runs through k_sketch_fast + k_query_sketch.  FA_QF_CAP lowers the capacity so
    # that ordinary fragments (about 240 records) take that road.
    g = syn.rng(141)
    anc = syn.random_codes(g, 300_000)
    refs = [[syn.to_ascii(syn.mutate_codes(g, anc, d))] for d in (0.02, 0.08)]
    query = [syn.to_ascii(syn.mutate_codes(g, anc, 0.04))]
    mapper, hits, ohits, det = run_both({}, refs, query)              # fused (the default where it applies)
    assert gpu_mappings(mapper) == oracle_mappings(det) and hit_tuples(hits) == ohits and len(ohits) == 2
    ms = (C.c_float * 24)()

    def again():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            h = mapper.query_draft(query)
        lib.fa_mapper_last_timings(mapper._h, ms, 24)
        return hit_tuples(h), int(ms[9]), (int(ms[17]), int(ms[18]))   # hits, void attempts, parts accepted (fused, two kernels)

    assert again() == (ohits, 0, (1, 0))                              # (the first call had sized the workspace)
    monkeypatch.setenv("FA_QF_CAP", "100")
    h, repeats, how = again()
    assert repeats >= 1 and how == (0, 1), "the fused launch should have been voided and its range repeated through the two kernels"
    assert h == ohits and gpu_mappings(mapper) == oracle_mappings(det)
    # ONE overflow is no verdict on the mapper: the next query -- an ordinary one -- runs k_query_fused again
    monkeypatch.delenv("FA_QF_CAP")
    assert again() == (ohits, 0, (1, 0))
    # overflows in a row back off: the second one makes the mapper skip the fused form for one pass, ...
    monkeypatch.setenv("FA_QF_CAP", "100")
    assert again() == (ohits, 1, (0, 1))
    assert again() == (ohits, 1, (0, 1))
    assert again() == (ohits, 0, (0, 1))                              # (served: straight through the two kernels, nothing void)
    assert again() == (ohits, 1, (0, 1))                              # tried again, overflowed again: three passes to skip now
    for _ in range(3):
        assert again() == (ohits, 0, (0, 1))
    # ... and a fused pass that is accepted clears the record
    monkeypatch.delenv("FA_QF_CAP")
    assert again() == (ohits, 0, (1, 0))
    monkeypatch.setenv("FA_QF_CAP", "100")
    assert again() == (ohits, 1, (0, 1))
    monkeypatch.delenv("FA_QF_CAP")
    assert again() == (ohits, 0, (1, 0))


def test_fused_sketch_stage_with_bytes_outside_acgt():
    # A query with N runs, IUPAC codes and lower case still takes the one-launch sketch stage: the tiles that touch such bytes
    # are sketched from the byte image by k_sketch_tiles<0, true> first, k_query_fused hashes the plain tiles and takes the
    # staged records of the others over.  Every mapping and hit as the oracle's; slot [17] of the timings says which form ran.
    g = syn.rng(515)
    anc = syn.random_codes(g, 200_000)
    refs = [[syn.to_ascii(syn.mutate_codes(g, anc, d))] for d in (0.01, 0.05, 0.10)]
    q = bytearray(syn.to_ascii(syn.mutate_codes(g, anc, 0.03)))
    q[1_000:1_040] = b"N" * 40                                   # inside the first fragment
    q[2_995:3_010] = b"N" * 15                                   # across a fragment boundary
    q[50_000:53_500] = b"N" * 3_500                              # a whole fragment and more
    q[90_000:90_010] = b"RYKMSWBDHV"                             # IUPAC
    q[120_000:121_000] = bytes(q[120_000:121_000]).lower()        # lower case (packed as plain bases: no exception)
    q[199_990:200_000] = b"N" * 10                               # the tail
    query = [bytes(q)]
    mapper, hits, ohits, det = run_both({}, refs, query, threads=4)
    assert gpu_mappings(mapper) == oracle_mappings(det) and hit_tuples(hits) == ohits and len(ohits) == 3
    ms = (C.c_float * 24)()
    lib.fa_mapper_last_timings(mapper._h, ms, 24)
    assert (int(ms[17]), int(ms[18])) == (1, 0), "the query pass should have run k_query_fused"
    # the resident-batch road, two genomes of which one is clean
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        batch = mapper.upload_genomes([query, [syn.to_ascii(syn.mutate_codes(g, anc, 0.02))]])
        got = [hit_tuples(h) for h in batch.query()]
    assert got[0] == ohits and len(got[1]) == 3


def test_batches_of_tiny_genomes_keep_the_identity_workgroup_order():
    # The offset-major workgroup order of k_l2_events deals groups of equal fragment offset to the eight XCDs.  A batch of
    # plasmid-sized genomes (one to three fragments each) has fewer groups than XCDs: the order would put every real workgroup
    # on one or two XCDs behind a grid that is mostly padding, so such passes keep the identity order (slot [19] of the
    # timings counts the parts that ran ordered); a batch of larger genomes (16 fragments each) is ordered.  Rows equal the oracle's either way.
    g = syn.rng(4242)
    anc = [syn.random_codes(g, 12_500) for _ in range(6)]
    refs = [[syn.to_ascii(syn.mutate_codes(g, a, 0.03))] for a in anc]
    tiny = [[syn.to_ascii(syn.mutate_codes(g, anc[i % 6], 0.05)[: 3_100 + 3_000 * (i % 3)])] for i in range(90)]
    big = [[syn.to_ascii(syn.mutate_codes(g, anc[i % 6], 0.05))] + [syn.to_ascii(syn.mutate_codes(g, anc[(i + 1) % 6], 0.02))] * 3 for i in range(8)]
    sk, osk = quiet_sketch(pf.Sketch), OracleSketch()
    for i, r in enumerate(refs):
        sk.add_draft(i, r); osk.add_draft(i, r)
    mapper = sk.index(); osk.index()
    ms = (C.c_float * 24)()
    for genomes, ordered in ((tiny, 0), (big, 1)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            batch = mapper.upload_genomes(genomes)
            assert int(batch.total_fragments.sum()) >= 64
            hits = [hit_tuples(hs) for hs in batch.query()]
            lib.fa_mapper_last_timings(mapper._h, ms, 24)
            want = [osk.query_draft(c, threads=8) for c in genomes]
        assert hits == want
        assert int(ms[19]) == ordered, (ordered, list(ms)[17:20])


def test_device_pool_keeps_and_returns_memory():
    """Device memory a handle gives back is kept by the library for the next request of its size class (the runtime scrubs memory
    that went through hipFree: a second index build of a process was 12 x slower than the first) and `device_trim` returns it."""
    import gc
    g = syn.rng(77)
    seq = syn.to_ascii(syn.random_codes(g, 300_000))
    pf.device_trim()
    for _ in range(2):
        sk = pf.Sketch()
        sk.add_genome("a", seq)
        m = sk.index()
        assert hit_tuples(m.query_genome(seq))[0][1] == 100.0
        del m, sk
        gc.collect()
    held = pf.device_trim()
    assert held > 0 and pf.device_trim() == 0


def test_add_drafts_gives_the_records_of_add_draft():
    """One run of the packer over the contigs of many genomes (Sketch.add_drafts) against a loop of add_draft: the same minimizer
    records, genome lengths and hits -- drafts with short contigs, an empty genome, lower case and N runs."""
    g = syn.rng(4242)
    anc = syn.random_codes(g, 90_000)
    genomes = []
    for d in (0.0, 0.03, 0.09):
        seq = bytearray(bytes(syn.to_ascii(syn.mutate_codes(g, anc, d) if d else anc)))
        seq[5_000:5_200] = b"N" * 200
        seq[20_000:20_500] = bytes(seq[20_000:20_500]).lower()
        genomes.append(syn.split_contigs(g, np.frombuffer(bytes(seq), np.uint8), 4) + [b"ACG"])
    genomes.insert(1, [])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        many, loop = pf.Sketch().add_drafts(range(len(genomes)), genomes), pf.Sketch()
        for i, contigs in enumerate(genomes):
            loop.add_draft(i, contigs)
        assert all(np.array_equal(x, y) for x, y in zip(many._read_minimizers(), loop._read_minimizers()))
        m1, m2 = many.index(), loop.index()
        q = [bytes(syn.to_ascii(syn.mutate_codes(g, anc, 0.02)))]
        assert hit_tuples(m1.query_draft(q)) == hit_tuples(m2.query_draft(q)) and len(m1.query_draft(q)) == 3


# ----------------------------------------------------------------------------------------------------------------
# the slide geometry of the index build (rec_prev / rec_fwd / rec_bwd / flags) against its definitions
# ----------------------------------------------------------------------------------------------------------------
def _links_by_definition(h, s, w, cmw):
    """DESIGN.md section 3 in numpy, contig by contig: what k_window_links and k_link_duplicates must produce."""
    n = len(h)
    fwd, bwd = np.zeros(n, np.int32), np.zeros(n, np.int32)
    prev, flags = np.full(n, -1, np.int32), np.zeros(n, np.uint8)
    bounds = np.flatnonzero(np.diff(s)) + 1
    for lo, hi in zip(np.r_[0, bounds], np.r_[bounds, n]):
        wp = w[lo:hi].astype(np.int64)
        fwd[lo:hi] = lo + np.searchsorted(wp, wp + cmw, "left")
        bwd[lo:hi] = lo + np.searchsorted(wp, wp - cmw + 1, "right") - 1
        same = np.isin(wp[1:] + cmw - 1, wp)                     # the position that drops r (wpos[r + 1]) also admits a record
        flags[lo:hi - 1][same] |= 4
        order = np.argsort(h[lo:hi], kind="stable")              # same hash, consecutive in record order, same contig
        hs = h[lo:hi][order]
        dup = np.flatnonzero(hs[1:] == hs[:-1])
        cur, prv = lo + order[dup + 1], lo + order[dup]
        prev[cur] = prv
        linked = (w[prv + 1].astype(np.int64) - 1) >= (w[cur].astype(np.int64) - cmw)
        flags[cur[linked]] |= 1
        np.bitwise_or.at(flags, prv[linked], 2)
    return prev, fwd, bwd, flags


def _check_links(params, seed):
    g = syn.rng(seed)
    sk = pf.Sketch(**params)
    frag, k = sk.fragment_length, sk.k
    for i in range(5):
        base = syn.random_codes(g, 120_000)
        base[40_000:52_000] = base[10_000:22_000]                 # a repeat inside the contig: earlier records with the same hash
        base[90_000:90_400] = 0                                   # a low-complexity run: same hash in consecutive windows
        seq = syn.to_ascii(base)
        sk.add_draft(f"g{i}", [seq[:70_000], seq[70_000:110_000], seq[110_000:], seq[:frag + 50]])
    mapper = sk.index()
    h, s, w = mapper.minimizers._arrays()
    n = len(h)
    prev, fwd, bwd = (np.empty(n, np.int32) for _ in range(3))
    flags = np.empty(n, np.uint8)
    got = C.c_int64(0)
    check(lib.fa_mapper_debug_links(mapper._h, prev.ctypes.data, fwd.ctypes.data, bwd.ctypes.data, flags.ctypes.data, n, C.byref(got)))
    assert got.value == n
    cmw = max(1, frag - (mapper.window_size - 1) - (k - 1))
    wprev, wfwd, wbwd, wflags = _links_by_definition(h, s, w, cmw)
    assert (wprev >= 0).sum() > 100 and (wflags & 1).any() and (wflags & 2).any() and (wflags & 4).any()
    assert np.array_equal(fwd, wfwd) and np.array_equal(bwd, wbwd)
    assert np.array_equal(prev, wprev)
    assert np.array_equal(flags, wflags)
    return cmw


@pytest.mark.parametrize("bits", ["2", "6", "10"])
def test_index_links_match_their_definitions(bits, monkeypatch):
    """Blocks of 4 and 64 records (inside one contig or straddling a boundary) and the default of 1 024 (every block of this
    small index straddles: the gather path); default parameters (cmw = 2 962: halo inside LDS), a short fragment (cmw of a few
    hundred) and a long one with w = 1 (cmw > the largest LDS halo: the searches leave the tile)."""
    monkeypatch.setenv("FA_LINK_BLOCK_BITS", bits)
    assert _check_links({}, 31) == 2962
    assert _check_links({"fragment_length": 500, "k": 16}, 32) < 500
    assert _check_links({"fragment_length": 7000, "percentage_identity": 60.0, "k": 12}, 33) > 5120


@pytest.mark.parametrize("over_cap", [None, "1"])
def test_frequency_threshold_with_lists_beyond_the_histogram(over_cap, monkeypatch):
    """computeFreqHist on the device takes the largest list lengths from a histogram of the lengths below 4 096 plus the longer
    ones verbatim.  With room for no
    long list (FA_FREQ_OVER_CAP=1) the build sorts all lengths instead: same threshold, same index as the oracle's.
    (Two 45-mers planted 5 300 and 4 600 times: list lengths 4 569, 4 569, 3 689, ... and 2 lists to ignore -> threshold 4 569.)"""
    if over_cap:
        monkeypatch.setenv("FA_FREQ_OVER_CAP", over_cap)
    g = syn.rng(77)
    n = 3_200_000
    m = syn.random_codes(g, n)
    r1, r2 = syn.random_codes(g, 45), syn.random_codes(g, 45)
    for p in range(1000, n - 1000, 600):
        m[p: p + 45] = r1
    for p in range(1300, n - 1000, 700):
        m[p: p + 45] = r2
    refs = [[syn.to_ascii(m)]]
    sk, osk = pf.Sketch(), OracleSketch()
    for i, r in enumerate(refs):
        sk.add_draft(f"r{i}", r)
        osk.add_draft(f"r{i}", r)
    mapper = sk.index()
    osk.index()
    assert len(mapper.lookup_index) == osk.index_size
    assert mapper.occurences_threshold == osk.freq_threshold
    assert osk.freq_threshold > 4096                                # the threshold itself comes from a list beyond the histogram


def test_rows_of_a_pass_by_every_road():
    """The rows of a pass are formed by the last workgroup of k_cgi_rows (below FA_ROWS_EMIT_MAX pairs, default 512) or by
    kernels of their own, and reach the host written by the publishing workgroup (up to 256 rows) or by one DMA copy: a call of
    12 queries x 40 references (480 pairs, 360 rows) with both formations, and one query x 40 (30 rows), must give the same rows."""
    import subprocess
    import textwrap
    code = textwrap.dedent("""
        import sys, os, json, hashlib
        sys.path.insert(0, %r)
        import numpy as np
        import pyfastani_amd as pf
        from pyfastani_amd import synthetic as syn
        g = syn.rng(4242)
        anc = syn.random_codes(g, 150_000)
        refs = [[syn.to_ascii(syn.mutate_codes(g, anc, 0.01 + 0.004 * i))] for i in range(30)] + [[syn.to_ascii(syn.random_codes(g, 150_000))] for _ in range(10)]
        sk = pf.Sketch()
        sk.add_drafts([f"r{i}" for i in range(40)], refs)
        mapper = sk.index()
        batch = mapper.upload_genomes([[syn.to_ascii(syn.mutate_codes(g, anc, 0.02))] for _ in range(12)])
        many = batch.query_rows(0, 12)
        one = batch.query_rows(3, 1)
        dig = lambda r: hashlib.sha256(np.ascontiguousarray(r).tobytes()).hexdigest()[:16]
        print(json.dumps({"many": dig(many), "n_many": int(len(many)), "one": dig(one), "n_one": int(len(one))}))
    """) % (ROOT,)
    outs = []
    for emit in ("16384", "1", None):
        env = dict(os.environ)
        if emit:
            env["FA_ROWS_EMIT_MAX"] = emit
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert res.returncode == 0, res.stdout + res.stderr
        outs.append(json.loads(res.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1] == outs[2], outs
    assert outs[0]["n_many"] == 12 * 30 and outs[0]["n_one"] == 30, outs


# ----------------------------------------------------------------------------------------------------------------
# genome-LIKE inputs: repeats (7 x 5 kb, 30 x 1.3 kb on both strands), low-complexity tracts, indels, a 100 kb inversion
# ----------------------------------------------------------------------------------------------------------------
def test_genome_like_all_vs_all_matches_oracle_mapping_for_mapping():
    """12 genome-like genomes of 1 Mb (3 families x 4: `workloads.genome_like`), every genome mapped against the index of all of
    them: every L2 mapping, every CGI row and every hit equals the oracle's.  The inputs the reference's own benchmark uses are
    real assemblies (benches/mapping/bench.py:25-29); these carry what i.i.d. sequence lacks -- several loci per fragment and
    genome, exact ties between repeat copies, fragments broken by indels, the reverse strand over 100 kb."""
    from pyfastani_amd import workloads
    genomes, fam = workloads.genome_like(6000, 3, 4, 1_000_000)
    n = len(genomes)
    sk, osk = pf.Sketch(), OracleSketch()
    sk.add_drafts(list(range(n)), genomes)
    osk.add_drafts(list(range(n)), genomes)
    mapper = sk.index()
    osk.index()
    assert len(mapper.minimizers) == len(osk.minimizers()[0]) and mapper.occurences_threshold == osk.freq_threshold
    n_maps = multi = 0
    for q, contigs in enumerate(genomes):
        hits = mapper.query_draft([bytes(c) for c in contigs])
        got = gpu_mappings(mapper)
        ohits, det = osk.query_draft(contigs, threads=os.cpu_count() or 1, details=True)
        want = oracle_mappings(det)
        assert got == want, f"query {q}: {len(got)} mappings, oracle {len(want)}"
        assert hit_tuples(hits) == ohits, f"query {q}"
        n_maps += len(got)
        per_frag_genome = {}
        for qs, rs, *_ in got:
            per_frag_genome[(qs, rs)] = per_frag_genome.get((qs, rs), 0) + 1
        multi += sum(1 for v in per_frag_genome.values() if v > 1)
        # a genome hits itself at 100.0 (a fragment that ends inside a low-complexity tract may share one minimizer less with the
        # whole-contig sketch: the oracle shows the same 99.9999x); the exact repeat copies cost it a few matches (fragments inside
        # copies 2..n tie and land in the first copy's bin -- the reference's Shigella golden shows the same, 1600/1608:
        # test_ani.py:86-91)
        me = [h for h in hits if h.name == q][0]
        assert me.identity >= 99.999 and me.fragments - 12 <= me.matches <= me.fragments
        assert all(fam[h.name] == fam[q] for h in hits)
    assert n_maps > 8000 and multi > 50              # (fragments with several loci on one contig: what i.i.d. genomes never produce)


def test_handles_released_from_a_thread_that_never_entered_the_library():
    """A Sketch / Mapper / GenomeBatch may be garbage-collected on ANY thread (a torch worker, a finaliser): the device pool files
    the blocks under -- and synchronises -- the device that owns them, not the releasing thread's current one (fa_common.h:
    DevPool::free).  Release everything from a fresh thread, then build and query again: same hits, and the pool handed the
    blocks out again instead of growing."""
    import threading
    from pyfastani_amd._lib import lib as L
    g = syn.rng(4242)
    anc = syn.random_codes(g, 200_000)
    refs = [bytes(syn.to_ascii(syn.mutate_codes(g, anc, d))) for d in (0.02, 0.06)]
    query = bytes(syn.to_ascii(syn.mutate_codes(g, anc, 0.04)))

    def build():
        sk = pf.Sketch()
        for i, r in enumerate(refs):
            sk.add_genome(i, r)
        mp = sk.index()
        return sk, mp, mp.upload_genomes([[query]]), hit_tuples(mp.query_genome(query))
    box = list(build())
    want = box.pop()
    pf.device_trim()

    def release():
        box.clear()                                   # the last references die here, on this thread
        import gc
        gc.collect()
    t = threading.Thread(target=release)
    t.start()
    t.join()
    held = C.c_uint64(0)
    check(L.fa_device_trim(C.byref(held)))            # what the pool held = the blocks the thread gave back
    assert held.value > 0
    sk, mp, batch, got = build()
    assert got == want and len(got) == 2
    import torch
    if torch.cuda.device_count() >= 2:                # the owning device is NOT the releasing thread's current one
        pf.set_device(1)
        box2 = list(build())
        assert box2.pop() == want
        t = threading.Thread(target=lambda: (box2.clear(), __import__("gc").collect()))
        t.start()
        t.join()
        assert hit_tuples(build()[1].query_genome(query)) == want
        pf.set_device(0)


@pytest.mark.parametrize("env", [{"FA_L1_PREFILTER": "1"}, {"FA_L1_PREFILTER": "1", "FA_L1_THIN_SMALL": "0", "FA_L1_THIN_MID": "0"},
                                 {"FA_L1_THIN_SMALL": "2"}, {"FA_EV_RANK": "0"}, {"FA_L2_SCAN_ORDER": "1"}, {"FA_L2_SCAN_ORDER": "0"}],
                         ids=["prefilter", "prefilter+every-class", "small-in-middle-form", "probe-ranks", "scan-sorted", "scan-identity"])
def test_round6_kernel_forms_forced(env):
    """The forms of round 6 that the defaults only pick on large or unusual indices, forced onto the tests whose inputs reach them:
    the pre-filter of k_l1's block sort on every index (chance hits, planted repeats, the frequency threshold, seed counts across
    the merge tiers, the random-seed regime); every size class of k_l1 with a launch of its own / the small class folded into the
    512-thread form (the genome-like genomes hold fragments of all three classes); k_l2_events with the rank structure of rounds
    2-5; k_l2_scan over loci sorted by stream length on passes of any size, and never.  Same oracle, same bit-exact bar."""
    import subprocess
    pick = ("test_genome_like or test_frequency_threshold_active or test_seed_counts_across_the_merge_tiers or test_random_seed_regime "
            "or test_l1_candidates or test_end_to_end_vs_oracle or test_seed_overflow_to_hbm_scratch or test_small_sketch_against_crowded_window "
            "or test_config5_cells_all_vs_all or test_committed_fixtures")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k", pick],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0 and " passed" in res.stdout, res.stdout[-3000:] + res.stderr[-2000:]

"""The CPU oracle against every golden the reference's own tests hold for this path."""
import json
import os

import numpy as np
import pytest

from conftest import read_fasta
from oracle.oracle import OracleSketch, lib as olib, murmur_hash
from pyfastani_amd import synthetic as syn


def test_murmur_vectors(golden_dir):
    # independent implementation (Appleby's MurmurHash3 as shipped in scikit-learn), see make_murmur_vectors.py
    vectors = json.load(open(os.path.join(golden_dir, "murmur3_vectors.json")))
    assert len(vectors) > 200
    for v in vectors:
        assert murmur_hash(v["kmer"]) == v["hash"], v


def test_window_size_default():
    # reference: src/pyfastani/tests/test_ani.py:60,80
    assert OracleSketch().window_size == 24


def test_window_size_table():
    # secondary, probe-derived values of SURVEY.md 8c (regression anchors)
    want = {(14, 1000): 12, (14, 3000): 37, (14, 5000): 50, (16, 1000): 13, (16, 3000): 24, (16, 5000): 40,
            (21, 1000): 1000, (21, 3000): 15, (21, 5000): 25}
    for (k, frag), w in want.items():
        assert OracleSketch(k=k, fragment_length=frag).window_size == w


def test_protein_golden(golden_dir):
    # reference: src/pyfastani/tests/test_ani.py:96-115 (note: BGC0001425's contigs are added twice there)
    b1 = read_fasta(os.path.join(golden_dir, "BGC0001425.faa"))
    b3 = read_fasta(os.path.join(golden_dir, "BGC0001428.faa"))
    sk = OracleSketch(protein=True, fragment_length=100)
    sk.add_draft("BGC0001425", b1)
    sk.add_draft("BGC0001427", b1)
    assert len(sk.minimizers()[0]) == 36054          # SURVEY.md 8c secondary value
    sk.index()
    assert sk.index_size == 13890 and sk.freq_threshold == 2**31 - 1
    hits, det = sk.query_draft(b3, details=True)
    assert [(h[0], h[2], h[3]) for h in hits] == [("BGC0001425", 130, 176), ("BGC0001427", 130, 176)]
    assert hits[0][1] == pytest.approx(94.99492645263672, abs=1e-6)
    assert len(det["mappings"]["qseq"]) == 496


def test_protein_intended_second_reference(golden_dir):
    b1 = read_fasta(os.path.join(golden_dir, "BGC0001425.faa"))
    b2 = read_fasta(os.path.join(golden_dir, "BGC0001427.faa"))
    b3 = read_fasta(os.path.join(golden_dir, "BGC0001428.faa"))
    sk = OracleSketch(protein=True, fragment_length=100)
    sk.add_draft("BGC0001425", b1)
    sk.add_draft("BGC0001427", b2)
    sk.index()
    hits = sk.query_draft(b3)
    assert [(h[0], h[2], h[3]) for h in hits] == [("BGC0001427", 132, 176), ("BGC0001425", 130, 176)]


@pytest.mark.parametrize("seed", [1, 2])
def test_self_query_invariant(seed):
    # implied by test_ani.py:66-71,86-91: a genome queried against itself scores exactly 100.0, all fragments matched
    g = syn.rng(seed)
    codes = syn.random_codes(g, 150_000)
    sk = OracleSketch()
    sk.add_genome("self", syn.to_ascii(codes))
    sk.index()
    assert sk.query_draft([syn.to_ascii(codes)]) == [("self", 100.0, 50, 50)]
    assert sk.query_draft([syn.to_ascii(syn.reverse_complement_codes(codes))]) == [("self", 100.0, 50, 50)]


def test_divergence_calibration():
    g = syn.rng(5)
    codes = syn.random_codes(g, 150_000)
    sk = OracleSketch()
    sk.add_genome("anc", syn.to_ascii(codes))
    sk.index()
    last = 101.0
    for d in (0.02, 0.05, 0.10, 0.15):
        hits = sk.query_draft([syn.to_ascii(syn.mutate_codes(g, codes, d))])
        assert len(hits) == 1 and hits[0][1] < last and abs(hits[0][1] - 100 * (1 - d)) < 4.0
        last = hits[0][1]
    assert sk.query_draft([syn.to_ascii(syn.random_codes(g, 150_000))]) == []


def test_minimum_hits_default_sketch():
    # s=240 -> strict 5, relaxed 2
    assert olib().fo_min_hits(240, 16, 80.0) == 5
    assert olib().fo_min_hits_relaxed(240, 16, 80.0) == 2


def test_winnowing_quirks():
    sk = OracleSketch()
    # period-4 repeat: every change of front carries the hash of the first record (wpos 0) -> suppressed until a
    # different hash shows up (_fastani.pyx:216,220)
    h, w = sk.sketch_sequence(b"A" * 5000)
    assert len(h) == 1 and w[0] == 0
    # strand-symmetric k-mers are skipped entirely (_fastani.pyx:202)
    h, w = sk.sketch_sequence(b"AT" * 3000)
    assert len(h) == 0
    # contigs with w <= len < w + k - 1 add nothing silently (S2.7)
    assert len(sk.sketch_sequence(b"ACGTTGCAAC" * 3)[0]) == 0


def test_canonical_strand():
    g = syn.rng(9)
    codes = syn.random_codes(g, 20_000)
    sk = OracleSketch()
    hf, _ = sk.sketch_sequence(syn.to_ascii(codes))
    hr, _ = sk.sketch_sequence(syn.to_ascii(syn.reverse_complement_codes(codes)))
    # same multiset of minimizer hashes up to boundary effects
    inter = len(set(hf.tolist()) & set(hr.tolist()))
    assert inter > 0.95 * len(set(hf.tolist()))


def test_parallel_add_matches_sequential():
    """OracleSketch.add_drafts (threaded, used by the full-size GPU tests) gives the records, counters and lengths of
    one add_draft call per genome -- including short contigs, empty contigs and the boundary between contigs."""
    g = syn.rng(77)
    genomes = []
    for i in range(5):
        seq = syn.to_ascii(syn.random_codes(g, 60_000 + 1000 * i))
        genomes.append(syn.split_contigs(g, seq, 4) + ([b"ACGT", b""] if i % 2 else []))
    genomes.append([b"ACGTTGCA" * 40])                     # leading duplicate run at the start of a contig
    names = [f"g{i}" for i in range(len(genomes))]
    a, b = OracleSketch(), OracleSketch()
    for n, c in zip(names, genomes):
        a.add_draft(n, c)
    b.add_drafts(names, genomes, threads=3)
    for x, y in zip(a.minimizers(), b.minimizers()):
        assert np.array_equal(x, y)
    assert a.names == b.names
    L = olib()
    assert L.fo_num_genomes(a._h) == L.fo_num_genomes(b._h) == len(genomes)
    assert [L.fo_genome_length(a._h, i) for i in range(len(genomes))] == [L.fo_genome_length(b._h, i) for i in range(len(genomes))]
    a.index(); b.index()
    assert a.index_size == b.index_size and a.freq_threshold == b.freq_threshold

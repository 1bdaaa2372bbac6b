"""The seven nucleotide known-answer constants of the reference (src/pyfastani/tests/test_ani.py:47-51,62-71,82-91).

Their inputs are FastANI's example genomes, which are dangling symlinks in the reference checkout.  Drop
``Escherichia_coli_str_K12_MG1655.fna`` and ``Shigella_flexneri_2a_01.fna`` into ``tests/golden/`` and these tests pin
nucleotide-mode parity of the oracle (CPU) and of the HIP path (GPU); until then they skip."""
import os

import pytest

from conftest import ROOT, read_fasta

ECOLI = os.path.join(ROOT, "tests", "golden", "Escherichia_coli_str_K12_MG1655.fna")
SFLEXNERI = os.path.join(ROOT, "tests", "golden", "Shigella_flexneri_2a_01.fna")
needs_files = pytest.mark.skipif(not (os.path.exists(ECOLI) and os.path.exists(SFLEXNERI)), reason="missing FastANI data files")


def _check(make_sketch, minimizers, index_size, query):
    sk = make_sketch()
    assert sk.window_size == 24
    ecoli, shigella = read_fasta(ECOLI), read_fasta(SFLEXNERI)
    sk.add_draft("Escherichia_coli_str_K12_MG1655", ecoli)
    assert minimizers(sk) == 371301
    m = sk.index()
    assert index_size(m) == 361568
    assert query(m, shigella, 4) == [("Escherichia_coli_str_K12_MG1655", 97.7507, 1303, 1608)]
    assert query(m, ecoli, 7) == [("Escherichia_coli_str_K12_MG1655", 100.0, 1547, 1547)]
    sk = make_sketch()
    sk.add_draft("Shigella_flexneri_2a_01", shigella)
    assert minimizers(sk) == 386387
    m = sk.index()
    assert index_size(m) == 347908
    assert query(m, shigella, 7) == [("Shigella_flexneri_2a_01", 100.0, 1600, 1608)]


@needs_files
def test_oracle_nucleotide_goldens():
    from oracle.oracle import OracleSketch
    _check(OracleSketch, lambda s: len(s.minimizers()[0]), lambda m: m.index_size,
           lambda m, q, places: [(h[0], round(h[1], places), h[2], h[3]) for h in m.query_draft(q, threads=os.cpu_count())])


@needs_files
@pytest.mark.gpu
def test_hip_nucleotide_goldens():
    import pyfastani_amd as pf
    _check(pf.Sketch, lambda s: len(s.minimizers), lambda m: len(m.lookup_index),
           lambda m, q, places: [(h.name, round(h.identity, places), h.matches, h.fragments) for h in m.query_draft(q)])

"""Host-side behaviour of the pyfastani surface that needs no GPU: constructor validation, value classes, pickling
of value classes.  Assertions follow the reference's src/pyfastani/tests/test_sketch.py."""
import pickle
import warnings

import pytest

import pyfastani_amd as pf


def test_init_errors():
    # test_sketch.py:12-23
    with pytest.raises(TypeError):
        pf.Sketch(k="1")
    with pytest.raises(TypeError):
        pf.Sketch(fragment_length="1")
    with pytest.raises(TypeError):
        pf.Sketch(minimum_fraction="0.5")
    with pytest.raises(OverflowError):
        pf.Sketch(k=2**32)
    with pytest.raises(ValueError):
        pf.Sketch(k=0)
    with pytest.raises(ValueError):
        pf.Sketch(p_value=-1.0)
    with pytest.raises(ValueError):
        pf.Sketch(percentage_identity=-1.0)
    with pytest.raises(ValueError):
        pf.Sketch(percentage_identity=200.0)
    with pytest.raises(ValueError):
        pf.Sketch(minimum_fraction=1.5)
    with pytest.raises(BufferError):
        pf.Sketch(k=pf.MAX_KMER_SIZE + 1)
    with pytest.raises(TypeError):
        pf.Sketch(16)  # keyword-only


def test_large_k_warns():
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        sk = pf.Sketch(k=21)
    assert len(caught) == 1 and issubclass(caught[0].category, UserWarning)
    assert sk.k == 21 and sk.window_size == 15


def test_properties():
    sk = pf.Sketch()
    assert (sk.k, sk.window_size, sk.fragment_length, sk.protein) == (16, 24, 3000, False)
    assert sk.minimum_fraction == pytest.approx(0.2) and sk.percentage_identity == 80.0 and sk.p_value == 1e-3
    assert sk.names == [] and sk.occurences_threshold == 2**31 - 1
    pk = pf.Sketch(protein=True, fragment_length=100)
    assert pk.protein and pk.window_size == 1
    assert pf.MAX_KMER_SIZE == 2048


def test_reinit_and_short_contig_warning():
    # test_sketch.py:25-52 (packing and bookkeeping are host work; no minimizers are read here)
    sk = pf.Sketch(fragment_length=100)
    sk.add_genome("test", "ATGC" * 100)
    assert sk.names == ["test"] and sk.fragment_length == 100
    sk.__init__(fragment_length=200)
    assert sk.names == [] and sk.fragment_length == 200
    sk = pf.Sketch()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        sk.add_draft("short_seq", ["ATGC" * 1000, "ATGC"])
    assert len(caught) == 1
    assert sk.names == ["short_seq"]


def test_input_carriers():
    import numpy as np
    sk = pf.Sketch()
    sk.add_draft("mixed", ["ACGT" * 100, b"ACGT" * 100, bytearray(b"ACGT" * 100), memoryview(b"ACGT" * 100),
                           np.frombuffer(b"ACGT" * 100, dtype=np.uint8), "ACGT" * 50 + "Δ" + "ACGT" * 50])
    assert sk.names == ["mixed"]
    with pytest.raises((TypeError, ValueError, BufferError)):
        sk.add_genome("bad", np.zeros(100, dtype=np.float64))
    with pytest.raises(TypeError):
        sk.add_genome("bad", 12345)


def test_mapper_cannot_be_instantiated():
    with pytest.raises(TypeError):
        pf.Mapper()


def test_value_classes():
    h = pf.Hit("a", 97.75, 10, 20)
    assert h == pf.Hit("a", 97.75, 10, 20) and not (h == pf.Hit("a", 97.75, 11, 20))
    assert repr(h) == "Hit(name='a', identity=97.75, matches=10, fragments=20)"
    assert pickle.loads(pickle.dumps(h)) == h
    assert pf.Hit("a", 0.1, 1, 1).identity == pytest.approx(0.1, rel=1e-6) and pf.Hit("a", 0.1, 1, 1).identity != 0.1
    m = pf.MinimizerInfo(5, 1, 2)
    assert pickle.loads(pickle.dumps(m)) == m and repr(m) == "MinimizerInfo(hash=5, sequence_id=1, window_position=2)"
    p = pf.Position(3, 4)
    assert pickle.loads(pickle.dumps(p)) == p and repr(p) == "Position(sequence_id=3, window_position=4)"
    idx = pf.MinimizerIndex()
    idx[7] = [p]
    assert 7 in idx and idx[7] == [p] and len(idx) == 1 and list(idx) == [7]
    del idx[7]
    with pytest.raises(KeyError):
        idx[7]
    assert len(pf.Minimizers()) == 0


def test_threads_argument_validation():
    # _fastani.pyx:1050; checked before any device work
    mapper = pf.Mapper.__new__(pf.Mapper)
    with pytest.raises(ValueError):
        mapper.query_draft([], threads=-1)
    with pytest.raises(ValueError):
        mapper.query_genome("ACGT", threads=-2)


def test_add_drafts_is_add_draft_for_every_genome():
    """`Sketch.add_drafts` (one run of the packer over all contigs, fa_sketch_add_genomes): names, the short-contig warnings and
    the argument checks of n `add_draft` calls (host side only; the records are compared on the GPU, tests/test_gpu_parity.py)."""
    import warnings
    import pyfastani_amd as pf
    sk = pf.Sketch()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert sk.add_drafts(["a", "b", "c"], [[b"ACGT" * 200, b"AC"], [], ["ACGT" * 100, bytearray(b"TTGA" * 50)]]) is sk
    assert sk.names == ["a", "b", "c"] and len(w) == 1 and "short" in str(w[0].message)
    with pytest.raises(ValueError):
        sk.add_drafts(["x"], [[b"ACGT"], [b"ACGT"]])
    with pytest.raises(TypeError):
        sk.add_drafts(["x"], [[1234]])
    assert sk.names == ["a", "b", "c"]
    mixed = pf.Sketch().add_drafts(["m"], [[b"ACGT" * 100, "ACG\u0141" * 100]])     # bytes next to a UCS2 string: genome by genome
    assert mixed.names == ["m"]

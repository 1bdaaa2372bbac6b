"""Host ingest: the FASTA reader mirrors pyfastani._fasta.Parser (_fasta.pyx:33-103).  CPU only."""
import os

import pytest

from pyfastani_amd._fasta import Parser, Record


def write(tmp_path, name, data):
    p = tmp_path / name
    p.write_bytes(data)
    return str(p)


def test_records_ids_and_upper_casing(tmp_path):
    path = write(tmp_path, "a.fna", b">seq1 some description\nacgtNNac\nGGTT\n\n>seq2\nA\n>empty\n>last\nacgu-*\nTT")
    recs = list(Parser(path))
    assert [r.id for r in recs] == ["seq1 some description", "seq2", "empty", "last"]
    assert [r.seq for r in recs] == [b"ACGTNNACGGTT", b"A", b"", b"ACGU-*TT"]
    assert all(isinstance(r, Record) and isinstance(r.seq, bytes) and isinstance(r.id, str) for r in recs)


def test_greater_than_inside_a_line_is_sequence(tmp_path):
    recs = list(Parser(write(tmp_path, "gt.fa", b">a\nAC>GT\nTT\n>b\nGG\n")))
    assert [(r.id, r.seq) for r in recs] == [("a", b"AC>GTTT"), ("b", b"GG")]


def test_add_fasta_counts_records_and_short_contigs(tmp_path):
    # host-side part of Sketch.add_fasta (no device needed until the sketch is flushed): one genome per file, a record
    # shorter than the window is reported like add_draft reports it
    import warnings
    import pyfastani_amd as pf
    body = (b"ACGTTGCA" * 10 + b"\n") * 400
    path = write(tmp_path, "g.fna", b">c1\n" + body + b">c2 short\nACGT\n>c3\n" + body)
    sk = pf.Sketch()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sk.add_fasta("genome", path)
    assert sk.names == ["genome"] and len(w) == 1 and "short" in str(w[0].message)


def test_add_fasta_many_counts_like_add_fasta(tmp_path):
    # fa_sketch_add_fasta_many: every file read + packed by its own host task, one genome per file in the order given; names,
    # short-contig warnings and errors as n calls of add_fasta (host side only: no device needed until the sketch is flushed)
    import warnings
    import pyfastani_amd as pf
    body = (b"ACGTTGCA" * 10 + b"\n") * 400
    paths = [write(tmp_path, f"g{i}.fna", b">c1\n" + body + (b">c2 short\nACGT\n" if i % 2 else b"") + b">c3\n" + body) for i in range(5)]
    sk = pf.Sketch()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert sk.add_fasta_many([f"n{i}" for i in range(5)], paths) is sk
    assert sk.names == [f"n{i}" for i in range(5)] and len(w) == 2 and all("short" in str(x.message) for x in w)
    with pytest.raises(ValueError):
        sk.add_fasta_many(["a"], paths[:2])
    with pytest.raises(OSError):
        sk.add_fasta_many(["a", "b"], [paths[0], str(tmp_path / "missing.fna")])
    assert sk.names == [f"n{i}" for i in range(5)]                      # a failed call adds nothing
    assert sk.add_fasta_many([], []).names == sk.names


def test_host_pieces_against_their_definition():
    """The host-only pieces of the library -- the 2-bit packer, the FASTA readers (the one-sweep reader of round 5 against
    the byte-by-byte definition of the store on 40 random files, prefixes of records, protein), the statistics tables, the
    workspace lease -- built from the same headers with AddressSanitizer + UBSan and run (scripts/host_sanitize/driver.cpp)."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    from conftest import ROOT
    out = os.path.join(ROOT, "build", "host_sanitize")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "driver_pytest")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread",
                    os.path.join(ROOT, "scripts", "host_sanitize", "driver.cpp"), "-o", exe], check=True, capture_output=True, text=True)
    for env in ({"FA_HOST_THREADS": "8"}, {"FA_HOST_THREADS": "3", "FA_NO_AVX2": "1"}):
        res = subprocess.run([exe], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert res.returncode == 0 and "all checks passed" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def test_crlf_is_kept_like_the_reference(tmp_path):
    # only the '\n' is stripped (_fasta.pyx:95-96): a carriage return stays in the id and in the sequence
    recs = list(Parser(write(tmp_path, "crlf.fa", b">id\r\nAC\r\nGT\r\n")))
    assert recs[0].id == "id\r" and recs[0].seq == b"AC\rGT\r"


def test_file_not_starting_with_header_yields_nothing(tmp_path):
    assert list(Parser(write(tmp_path, "x.fa", b"ACGT\n>late\nACGT\n"))) == []
    assert list(Parser(write(tmp_path, "empty.fa", b""))) == []


def test_long_lines_and_large_record(tmp_path):
    body = (b"acgt" * 5000 + b"\n") * 30 + b"ACGT" * 100        # 20 000-character lines, no trailing newline
    recs = list(Parser(write(tmp_path, "long.fa", b">big\n" + body + b"\n>next\nTT\n")))
    assert recs[0].seq == b"ACGT" * (5000 * 30 + 100) and recs[1].seq == b"TT"


def test_header_longer_than_the_line_buffer(tmp_path):
    ok = b">" + b"x" * 2045 + b"\nACGT\n"                        # 2047 characters including the newline: fits
    assert list(Parser(write(tmp_path, "ok.fa", ok)))[0].id == "x" * 2045
    with pytest.raises(BufferError):
        list(Parser(write(tmp_path, "bad.fa", b">" + b"x" * 2046 + b"\nACGT\n")))
    with pytest.raises(BufferError):
        list(Parser(write(tmp_path, "eof.fa", b">header without newline")))


def test_errors(tmp_path):
    with pytest.raises(OSError):
        Parser(os.path.join(str(tmp_path), "missing.fa"))
    with pytest.raises(TypeError):
        Parser(b"bytes-path")
    with pytest.raises(TypeError):
        Record("id", "not bytes")


def test_protein_fixture_matches_plain_python_reader(golden_dir):
    path = os.path.join(golden_dir, "BGC0001425.faa")
    want, cur = [], None
    for line in open(path, "rb"):
        if line.startswith(b">"):
            cur = [line[1:-1].decode("latin-1"), b""]
            want.append(cur)
        else:
            cur[1] += line.rstrip(b"\n").upper()
    got = [(r.id, r.seq) for r in Parser(path)]
    assert got == [tuple(w) for w in want] and len(got) > 10


def test_packed_genomes_read_once_use_twice(tmp_path):
    # PackedGenomes: files read + packed once (host only); as references they count like add_fasta_many
    import warnings
    import pyfastani_amd as pf
    body = (b"ACGTTGCA" * 10 + b"\n") * 400
    paths = [write(tmp_path, f"p{i}.fna", b">c1\n" + body + (b">c2 short\nACGT\n" if i == 1 else b"")) for i in range(4)]
    packed = pf.PackedGenomes(paths)
    sizes, records, bases = packed.info()
    assert len(packed) == 4 and records == [1, 2, 1, 1] and bases[0] == 32_000 and bases[1] == 32_004 and sizes[0] == os.path.getsize(paths[0])
    sk = pf.Sketch()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sk.add_packed(["b", "c"], packed, 1, 2)
        sk.add_packed(["a"], packed, 0, 1)
    assert sk.names == ["b", "c", "a"] and len(w) == 1
    with pytest.raises(ValueError):
        sk.add_packed(["x"], packed, 3, 2)                   # file range outside the set
    with pytest.raises(ValueError):
        pf.Sketch(protein=True, fragment_length=100).add_packed(["x"], packed, 0, 1)   # packed for the other alphabet
    with pytest.raises(OSError):
        pf.PackedGenomes([paths[0], str(tmp_path / "missing.fna")])
    grown = pf.PackedGenomes(paths[:1]).extend(paths[1:3])
    assert len(grown) == 3 and grown.info()[1] == [1, 2, 1]
    with pytest.raises(OSError):
        grown.extend([str(tmp_path / "missing.fna")])
    assert len(grown) == 3


def test_add_fasta_reads_a_fifo_to_its_end(tmp_path):
    # a path that is not a regular file (a FIFO, a process substitution) reports size 0: the reader takes it until end of file
    # instead of returning an empty genome (fa_fasta.h: slurp).  Host-side only: the short record behind 2 MB of sequence is
    # reported, so the reader got to the end of the pipe.
    import threading
    import warnings
    import pyfastani_amd as pf
    body = (b"ACGTTGCA" * 10 + b"\n") * 25000
    data = b">c1\n" + body + b">c2 short\nACGT\n"
    fifo = str(tmp_path / "genome.fifo")
    os.mkfifo(fifo)

    def feed():
        with open(fifo, "wb") as f:
            f.write(data)
    t = threading.Thread(target=feed)
    t.start()
    sk = pf.Sketch()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sk.add_fasta("piped", fifo)
    t.join(30)
    assert sk.names == ["piped"] and len(w) == 1 and "short" in str(w[0].message)
    # ... and the one-sweep reader behind add_fasta_many (read() into a buffer sized by fstat) does the same
    os.unlink(fifo)
    os.mkfifo(fifo)
    t = threading.Thread(target=feed)
    t.start()
    sk = pf.Sketch()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sk.add_fasta_many(["piped"], [fifo])
    t.join(30)
    assert sk.names == ["piped"] and len(w) == 1 and "short" in str(w[0].message)

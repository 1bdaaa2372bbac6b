"""FASTA reader with the behaviour of ``pyfastani._fasta`` (src/pyfastani/_fasta.pyx:33-103), backed by the
memory-mapped native parser of ``libfastani_hip`` (host ingest).

``Parser(path)`` iterates over ``Record(id, seq)``: ``id`` is the header line without ``>`` and the newline, ``seq``
the upper-cased sequence as ``bytes``.  A file whose first line does not start with ``>`` yields nothing; a header that
does not fit the reference's 2048-byte line buffer raises ``BufferError``; a missing file raises ``OSError``.
"""
import ctypes as C
import os

from ._lib import lib, check


class Record:
    """A FASTA record (_fasta.pyx:33-39)."""

    __slots__ = ("id", "seq")

    def __init__(self, id, seq):
        if not isinstance(id, str):
            raise TypeError(f"id must be str, not {type(id).__name__}")
        if not isinstance(seq, bytes):
            raise TypeError(f"seq must be bytes, not {type(seq).__name__}")
        self.id = id
        self.seq = seq

    def __repr__(self):
        return f"Record({self.id!r}, <{len(self.seq)} bytes>)"


class Parser:
    """An iterator over the records of a FASTA file (_fasta.pyx:41-103)."""

    def __init__(self, path):
        if not isinstance(path, str):
            raise TypeError(f"path must be str, not {type(path).__name__}")
        self.path = path
        self._h = None
        h = C.c_void_p()
        check(lib.fa_fasta_open(os.fsencode(path), C.byref(h)))
        self._h = h

    def __del__(self):
        self.close()

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.fa_fasta_close(h)

    def __iter__(self):
        return self

    def __next__(self):
        if not self._h:
            raise StopIteration
        has = C.c_int(0)
        pid, pseq = C.c_void_p(), C.c_void_p()
        nid, nseq = C.c_int64(0), C.c_int64(0)
        check(lib.fa_fasta_next(self._h, C.byref(has), C.byref(pid), C.byref(nid), C.byref(pseq), C.byref(nseq)))
        if not has.value:
            raise StopIteration
        ident = C.string_at(pid, nid.value).decode("latin-1")       # PyUnicode_1BYTE_KIND, _fasta.pyx:82
        seq = C.string_at(pseq, nseq.value) if nseq.value else b""
        return Record(ident, seq)

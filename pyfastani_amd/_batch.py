"""Resident query batches: many query genomes packed 2-bit in HBM, mapped without leaving the device.

This is the many-to-many extension of the reference's one-query-at-a-time ``Mapper.query_draft``
(_fastani.pyx:1006-1136): the same per-genome semantics, but the inputs are uploaded once and any
sub-range of genomes can be mapped with one call (``fa_mapper_query_genomes``).
"""
import ctypes as C
import warnings

import numpy as np

from . import _lib
from ._lib import lib, check
from ._api import _borrow_all


class GenomeBatch:
    def __init__(self, mapper, genomes):
        self._mapper = mapper
        self._h = None
        ptrs, lens, owners, contig_genome = [], [], [], []
        width = None
        borrowed = []
        for gi, contigs in enumerate(genomes):
            bufs, w = _borrow_all(contigs)
            borrowed.append((bufs, w))
        widths = {w for bufs, w in borrowed if any(b[1] for b in bufs)}
        if len(widths) > 1:
            raise ValueError("all genomes of a batch must use the same character width (all bytes, or all str)")
        width = widths.pop() if widths else 1
        for gi, (bufs, _) in enumerate(borrowed):
            for b in bufs:
                ptrs.append(b[0])
                lens.append(b[1])
                owners.append(b[3])
                contig_genome.append(gi)
        self.n_genomes = len(borrowed)
        n = len(ptrs)
        c_ptrs = (C.c_void_p * max(n, 1))(*ptrs)
        c_lens = (C.c_int64 * max(n, 1))(*lens)
        cg = np.asarray(contig_genome, dtype=np.int32)
        h = C.c_void_p()
        check(lib.fa_genomes_upload(mapper._h, c_ptrs, c_lens, cg.ctypes.data if n else None, n, self.n_genomes, width,
                                    C.byref(h)))
        del owners
        self._finish(h)

    @classmethod
    def from_fasta(cls, mapper, paths):
        """One genome per FASTA file, parsed and packed by the library (``fa_genomes_upload_fasta``)."""
        import os
        self = cls.__new__(cls)
        self._mapper = mapper
        self._h = None
        paths = [os.fsencode(p) for p in paths]
        self.n_genomes = len(paths)
        arr = (C.c_char_p * max(len(paths), 1))(*paths)
        h = C.c_void_p()
        check(lib.fa_genomes_upload_fasta(mapper._h, arr, len(paths), C.byref(h)))
        self._finish(h)
        return self

    def _finish(self, h):
        self._h = h
        self.total_fragments = np.zeros(self.n_genomes, np.uint64)
        self.total_length = np.zeros(self.n_genomes, np.uint64)
        self.n_short = np.zeros(self.n_genomes, np.int32)
        ng = C.c_int32(0)
        check(lib.fa_genomes_info(self._h, C.byref(ng), self.total_fragments.ctypes.data, self.total_length.ctypes.data,
                                  self.n_short.ctypes.data))
        for _ in range(int(self.n_short.sum())):
            warnings.warn(
                "Mapper received a short sequence relative to parameters, mapping will not be computed.",
                UserWarning,
            )

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib.fa_genomes_free(self._h)
            except Exception:
                pass
            self._h = None

    def __len__(self):
        return self.n_genomes

    def query_rows(self, first=0, count=None):
        """Raw cgi::CGI_Results rows (structured numpy array) for genomes [first, first+count)."""
        if count is None:
            count = self.n_genomes - first
        cap = max(1, count * max(1, len(self._mapper._names)))
        rows = np.zeros(cap, dtype=ROW_DTYPE)
        n_rows = C.c_int64(0)
        check(lib.fa_mapper_query_genomes(self._mapper._h, self._h, first, count, rows.ctypes.data, cap,
                                          C.byref(n_rows), 0))
        return rows[: n_rows.value]

    def query_rows_device(self, first, count, device_ptr, cap):
        """Same, but rows are written to a caller-owned DEVICE buffer (e.g. a torch tensor)."""
        n_rows = C.c_int64(0)
        check(lib.fa_mapper_query_genomes(self._mapper._h, self._h, first, count, device_ptr, cap, C.byref(n_rows), 1))
        return n_rows.value

    def query(self, first=0, count=None):
        """One sorted hit list per genome, exactly what ``Mapper.query_draft`` returns for each."""
        if count is None:
            count = self.n_genomes - first
        rows = self.query_rows(first, count)
        out = [[] for _ in range(count)]
        per_genome = [[] for _ in range(count)]
        for r in rows:
            per_genome[int(r["query_id"]) - first].append(_RowView(r))
        for i in range(count):
            out[i] = self._mapper._rows_to_hits(per_genome[i], int(self.total_length[first + i]))
        return out


ROW_DTYPE = np.dtype(
    [("query_id", "<i4"), ("ref_genome_id", "<i4"), ("count_seq", "<i4"), ("total_query_fragments", "<i4"),
     ("identity", "<f4")]
)
assert ROW_DTYPE.itemsize == C.sizeof(_lib.CgiRow)


class _RowView:
    __slots__ = ("query_id", "ref_genome_id", "count_seq", "total_query_fragments", "identity")

    def __init__(self, r):
        self.query_id = int(r["query_id"])
        self.ref_genome_id = int(r["ref_genome_id"])
        self.count_seq = int(r["count_seq"])
        self.total_query_fragments = int(r["total_query_fragments"])
        self.identity = float(r["identity"])

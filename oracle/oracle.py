"""ctypes driver for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package (``pyfastani_amd``) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FA_ORACLE_DEFINES="FO_SLIDE_END=1,FO_L2_CI=0.75f": one of the alternative readings named at the top of fastani_oracle.hpp,
# built into a library of its own (scripts/oracle_sensitivity.py).  Unset = the default reading, the one everything else uses.
_DEFINES = [d.strip() for d in os.environ.get("FA_ORACLE_DEFINES", "").split(",") if d.strip()]
for _d in _DEFINES:
    if not _d.startswith("FO_") or not all(ch.isalnum() or ch in "_=." for ch in _d):
        raise ValueError(f"FA_ORACLE_DEFINES: {_d!r} is not an FO_<RULE>=<value> switch")
_TAG = "".join("__" + d.replace("=", "_").replace(".", "p") for d in _DEFINES)
_SO = os.path.join(_HERE, "_build", f"libfastani_oracle{_TAG}.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("oracle_capi.cpp", "fastani_oracle.hpp")]
    stale = lambda: not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)  # noqa: E731
    if force or stale():
        import fcntl
        os.makedirs(os.path.dirname(_SO), exist_ok=True)       # a fresh checkout has no _build/ yet
        with open(_SO + ".lock", "w") as lock:                 # several processes may ask at once: one runs make
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or stale():
                if _DEFINES:
                    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-pthread", "-shared"] + ["-D" + d for d in _DEFINES]
                                          + ["-o", _SO, os.path.join(_HERE, "oracle_capi.cpp")])
                else:
                    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i32, i64, u64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double
        P = C.POINTER
        L.fo_hash.restype = C.c_uint32
        L.fo_hash.argtypes = [C.c_char_p, i32]
        L.fo_recommended_window.restype = i32
        L.fo_recommended_window.argtypes = [f64, i32, i32, f32, i32, u64]
        L.fo_min_hits_relaxed.restype = i32
        L.fo_min_hits_relaxed.argtypes = [i32, i32, f32]
        L.fo_min_hits.restype = i32
        L.fo_min_hits.argtypes = [i32, i32, f32]
        L.fo_j2md.restype = f32
        L.fo_j2md.argtypes = [f32, i32]
        L.fo_md2j.restype = f32
        L.fo_md2j.argtypes = [f32, i32]
        L.fo_md_lower_bound.restype = f32
        L.fo_md_lower_bound.argtypes = [f32, i32, i32, f32]
        L.fo_identity.restype = None
        L.fo_identity.argtypes = [i32, i32, i32, P(f32), P(f32)]
        L.fo_new.restype = vp
        L.fo_new.argtypes = [i32, i32, f32, f64, f32, u64, i32, i32]
        L.fo_free.argtypes = [vp]
        L.fo_window.restype = i32
        L.fo_window.argtypes = [vp]
        L.fo_add_contig.restype = i32
        L.fo_add_contig.argtypes = [vp, vp, i64, i32]
        L.fo_end_genome.argtypes = [vp]
        L.fo_add_genomes.restype = None
        L.fo_add_genomes.argtypes = [vp, P(vp), P(i64), vp, i64, i32, i32, i32]
        L.fo_num_minimizers.restype = i64
        L.fo_num_minimizers.argtypes = [vp]
        L.fo_get_minimizers.argtypes = [vp, vp, vp, vp]
        L.fo_sketch_sequence.restype = i64
        L.fo_sketch_sequence.argtypes = [vp, vp, i64, i32, vp, vp, i64]
        L.fo_index.argtypes = [vp]
        L.fo_freq_threshold.restype = i32
        L.fo_freq_threshold.argtypes = [vp]
        L.fo_index_size.restype = i64
        L.fo_index_size.argtypes = [vp]
        L.fo_index_count.restype = i64
        L.fo_index_count.argtypes = [vp, C.c_uint32]
        L.fo_num_genomes.restype = i64
        L.fo_num_genomes.argtypes = [vp]
        L.fo_genome_length.restype = u64
        L.fo_genome_length.argtypes = [vp, i64]
        L.fo_l1_fragment.restype = i32
        L.fo_l1_fragment.argtypes = [vp, vp, i32, P(i32), P(i32), vp, vp, vp, i32]
        L.fo_query.restype = i32
        L.fo_query.argtypes = [vp, P(vp), P(i64), i32, i32, i32, P(i32), P(u64), P(u64), P(f64)]
        L.fo_get_hits.argtypes = [vp, vp, vp, vp, vp]
        L.fo_num_rows.restype = i64
        L.fo_num_rows.argtypes = [vp]
        L.fo_get_rows.argtypes = [vp, vp, vp, vp]
        L.fo_num_mappings.restype = i64
        L.fo_num_mappings.argtypes = [vp]
        L.fo_get_mappings.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def _as_bytes(seq):
    if isinstance(seq, str):
        return seq.encode("latin-1")
    if isinstance(seq, np.ndarray):
        return np.ascontiguousarray(seq, dtype=np.uint8).tobytes()
    return bytes(seq)


def murmur_hash(kmer):
    b = _as_bytes(kmer)
    return int(lib().fo_hash(b, len(b)))


class OracleSketch:
    """Mirror of pyfastani.Sketch/Mapper driven through the CPU oracle."""

    def __init__(self, k=16, fragment_length=3000, minimum_fraction=0.2, p_value=1e-3, percentage_identity=80.0,
                 reference_size=5_000_000, protein=False, window=0):
        if not protein and window <= 0:
            window = lib().fo_recommended_window(p_value, k, 4, percentage_identity, fragment_length, reference_size)
            if window <= 0:
                # no admissible sketch size: the reference reads an uninitialised variable here;
                # both the oracle and the product clamp to the largest window instead of emulating UB
                window = fragment_length
        self._h = lib().fo_new(k, fragment_length, minimum_fraction, p_value, percentage_identity, reference_size,
                               int(protein), window)
        self.names = []
        self.fragment_length = fragment_length
        self.k = k

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fo_free(self._h)
            self._h = None

    @property
    def window_size(self):
        return lib().fo_window(self._h)

    def add_draft(self, name, contigs):
        n_short = 0
        for c in contigs:
            b = _as_bytes(c)
            n_short += 1 - lib().fo_add_contig(self._h, b, len(b), 1)
        lib().fo_end_genome(self._h)
        self.names.append(name)
        return n_short

    def add_genome(self, name, seq):
        return self.add_draft(name, [seq])

    def add_drafts(self, names, genomes, threads=0):
        """Several draft genomes at once, sketched by `threads` host threads (0 = all cores); the records are those
        of one `add_draft` call per genome (checked in tests/test_oracle_golden.py)."""
        threads = threads or (os.cpu_count() or 1)
        bufs, cg = [], []
        for gi, contigs in enumerate(genomes):
            for c in contigs:
                bufs.append(_as_bytes(c))
                cg.append(gi)
        n = len(bufs)
        keep = [np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(1, np.uint8) for b in bufs]
        arr = (C.c_void_p * max(n, 1))(*[k.ctypes.data for k in keep])
        lens = (C.c_int64 * max(n, 1))(*[len(b) for b in bufs])
        cga = np.asarray(cg, dtype=np.int32)
        lib().fo_add_genomes(self._h, arr, lens, cga.ctypes.data if n else None, n, len(genomes), 1, threads)
        self.names.extend(names)

    def minimizers(self):
        n = lib().fo_num_minimizers(self._h)
        h = np.empty(n, np.uint32)
        s = np.empty(n, np.int32)
        w = np.empty(n, np.int32)
        lib().fo_get_minimizers(self._h, h.ctypes.data, s.ctypes.data, w.ctypes.data)
        return h, s, w

    def sketch_sequence(self, seq):
        b = _as_bytes(seq)
        cap = max(len(b), 1)
        h = np.empty(cap, np.uint32)
        w = np.empty(cap, np.int32)
        n = lib().fo_sketch_sequence(self._h, b, len(b), 1, h.ctypes.data, w.ctypes.data, cap)
        return h[:n].copy(), w[:n].copy()

    def index(self):
        lib().fo_index(self._h)
        return self

    @property
    def freq_threshold(self):
        return lib().fo_freq_threshold(self._h)

    @property
    def index_size(self):
        return lib().fo_index_size(self._h)

    def index_count(self, h):
        return lib().fo_index_count(self._h, int(h))

    def l1_fragment(self, frag, cap=4096):
        b = _as_bytes(frag)
        assert len(b) == self.fragment_length
        ss, mh = C.c_int(0), C.c_int(0)
        seq = np.empty(cap, np.int32)
        st = np.empty(cap, np.int32)
        en = np.empty(cap, np.int32)
        n = lib().fo_l1_fragment(self._h, b, 1, C.byref(ss), C.byref(mh), seq.ctypes.data, st.ctypes.data,
                                 en.ctypes.data, cap)
        assert n <= cap
        return ss.value, mh.value, list(zip(seq[:n].tolist(), st[:n].tolist(), en[:n].tolist()))

    def query_draft(self, contigs, threads=1, details=False):
        bufs = [_as_bytes(c) for c in contigs]
        n = len(bufs)
        arr = (C.c_void_p * max(n, 1))()
        lens = (C.c_int64 * max(n, 1))()
        keep = []
        for i, b in enumerate(bufs):
            cb = C.create_string_buffer(b, len(b)) if len(b) else C.create_string_buffer(1)
            keep.append(cb)
            arr[i] = C.cast(cb, C.c_void_p)
            lens[i] = len(b)
        n_short, tf, tl, sec = C.c_int(0), C.c_uint64(0), C.c_uint64(0), C.c_double(0)
        nh = lib().fo_query(self._h, arr, lens, n, 1, threads, C.byref(n_short), C.byref(tf), C.byref(tl), C.byref(sec))
        g = np.empty(nh, np.int32)
        ident = np.empty(nh, np.float32)
        m = np.empty(nh, np.int32)
        f = np.empty(nh, np.int32)
        lib().fo_get_hits(self._h, g.ctypes.data, ident.ctypes.data, m.ctypes.data, f.ctypes.data)
        hits = [(self.names[g[i]], float(ident[i]), int(m[i]), int(f[i])) for i in range(nh)]
        if not details:
            return hits
        nm = lib().fo_num_mappings(self._h)
        cols = [np.empty(nm, np.int32) for _ in range(5)] + [np.empty(nm, np.float32)]
        lib().fo_get_mappings(self._h, *[c.ctypes.data for c in cols])
        nr = lib().fo_num_rows(self._h)
        rg = np.empty(nr, np.int32)
        ri = np.empty(nr, np.float32)
        rc = np.empty(nr, np.int32)
        lib().fo_get_rows(self._h, rg.ctypes.data, ri.ctypes.data, rc.ctypes.data)
        return hits, {
            "n_short": n_short.value, "total_fragments": tf.value, "total_length": tl.value, "seconds": sec.value,
            "mappings": dict(zip(["qseq", "rseq", "rstart", "sketch", "shared", "identity"], cols)),
            "rows": {"genome": rg, "identity": ri, "count": rc},
        }

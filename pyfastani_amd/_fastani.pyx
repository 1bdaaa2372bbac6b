# coding: utf-8
# cython: language_level=3, language=c++, binding=False
# distutils: language = c++
"""pyfastani's Python surface compiled against the MI355X engine.

This is the Cython module a pyfastani maintainer obtains by applying INTEGRATION.md to
``src/pyfastani/_fastani.pyx``: the classes keep the reference's names, signatures, exceptions and warnings
(line numbers of the reference cited per method), but where the reference ``cimport``s FastANI's C++
(``_fastani.pyx:51-84``) this module ``cimport``s ``fastani_hip.pxd`` -- the C ABI of ``libfastani_hip.so``
(``include/fastani_hip.h``), whose entry points run hand-written HIP kernels on gfx950.  There is no CPU fallback:
without the library the import fails, without a GPU every compute call raises ``RuntimeError``.

Nothing in this module needs PyTorch; the few methods that hand device tensors to the multi-GPU layer
(``pyfastani_amd.sharding``) import it when they are called.
"""

cimport cython
from cpython.unicode cimport PyUnicode_DATA, PyUnicode_KIND, PyUnicode_GET_LENGTH
from libc.stdint cimport int32_t, int64_t, uint32_t, uint64_t, uintptr_t
from libc.string cimport memcpy
from libcpp.vector cimport vector

cimport fastani_hip as hip

import operator
import os
import threading
import warnings

MAX_KMER_SIZE = 2048           # _fastani.pyx:103-107
cdef int _INT_MAX = 2147483647

# concurrent query calls each use their own HIP stream; the runtime's default of 4 hardware queues would put several of
# them on one queue, so ask for more before the runtime starts (harmless if HIP is already initialised)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


# --- error plumbing -----------------------------------------------------------------------------------------------
cdef int _check(int code) except -1:
    """Map a C status code onto the exception the reference would raise (INTEGRATION.md section 2)."""
    if code == 0:
        return 0
    cdef const char* raw = hip.fa_last_error()
    msg = raw.decode("utf-8", "replace") if raw != NULL else ""
    if code == 1:
        raise ValueError(msg)
    if code == 3:
        raise MemoryError(msg)
    if code == 4:
        raise NotImplementedError(msg)
    if code == 6:
        raise OSError(msg)
    if code == 7:
        raise BufferError(msg)
    raise RuntimeError(msg)


def device_count():
    cdef int n = 0
    hip.fa_device_count(&n)
    return n


def device_trim():
    """Return the device memory the library keeps for reuse (freed indices, batches, workspaces) to the runtime; returns the
    number of bytes it held (``fa_device_trim``)."""
    cdef uint64_t held = 0
    _check(hip.fa_device_trim(&held))
    return int(held)


def set_device(int device):
    """Select the GPU for sketches and mappers created afterwards (one process per GPU: call once per rank)."""
    _check(hip.fa_set_device(device))


# --- argument coercion with the exceptions Cython's typed signatures raise (tests/test_sketch.py:12-23) -------------
cdef object _as_uint(object value, str name, object bits):
    try:
        v = operator.index(value)
    except TypeError:
        raise TypeError(f"an integer is required for {name!r}, got {type(value).__name__}") from None
    if v < 0:
        raise OverflowError(f"can't convert negative value to unsigned int ({name})")
    if v >= 1 << bits:
        raise OverflowError(f"value too large to convert to unsigned int ({name})")
    return v


cdef double _as_float(object value, str name) except? -1.0:
    if isinstance(value, (str, bytes, bytearray)) or not hasattr(value, "__float__"):
        raise TypeError(f"a float is required for {name!r}, got {type(value).__name__}")
    return float(value)


# --- borrowed contigs (_fastani.pyx:633-645 / :1073-1092) ---------------------------------------------------------
cdef struct _Span:
    const void* data
    int64_t length
    int kind


cdef int _borrow(object contig, _Span* out, list keep) except -1:
    """Pointer, length and character width of a contig given as ``str`` (read through its canonical UCS1/2/4
    representation, no copy) or as any C-contiguous ``unsigned char`` buffer.  `keep` receives what must stay alive."""
    cdef const unsigned char[::1] view
    if isinstance(contig, str):
        out.kind = PyUnicode_KIND(contig)
        out.data = PyUnicode_DATA(contig)
        out.length = PyUnicode_GET_LENGTH(contig)
        keep.append(contig)
        return 0
    mv = memoryview(contig)
    if mv.ndim != 1 or mv.itemsize != 1 or not mv.c_contiguous:
        raise BufferError("contig must be a C-contiguous buffer of bytes")
    if mv.format not in ("B", "b", "c"):
        raise ValueError(f"Buffer dtype mismatch, expected 'const unsigned char' but got format {mv.format!r}")
    if mv.format != "B":
        mv = mv.cast("B")
    view = mv
    out.kind = 1
    out.length = view.shape[0]
    out.data = <const void*> &view[0] if view.shape[0] > 0 else NULL
    keep.append(mv)
    return 0


cdef int _borrow_all(object contigs, vector[const void*]& ptrs, vector[int64_t]& lens, list keep) except -1:
    """Borrow every contig of a genome with ONE character width (the C ABI takes one per call); returns that width.
    Mixed ``str`` kinds are widened to UCS4 -- same characters, _fastani.pyx:144-148 reads them one by one anyway."""
    cdef _Span sp
    cdef vector[_Span] spans
    cdef int width = 0
    cdef bint mixed = False
    cdef size_t i
    items = list(contigs)
    for contig in items:
        _borrow(contig, &sp, keep)
        spans.push_back(sp)
        if sp.length > 0:
            if width == 0:
                width = sp.kind
            elif width != sp.kind:
                mixed = True
    if mixed:
        import numpy as np
        spans.clear()
        for contig in items:
            text = contig if isinstance(contig, str) else bytes(memoryview(contig)).decode("latin-1")
            arr = np.frombuffer(text.encode("utf-32-le", "surrogatepass"), dtype=np.uint32)
            keep.append(arr)
            sp.kind = 4
            sp.length = len(text)
            sp.data = <const void*> <uintptr_t> (arr.ctypes.data if arr.size else 0)
            spans.push_back(sp)
        width = 4
    for i in range(spans.size()):
        ptrs.push_back(spans[i].data)
        lens.push_back(spans[i].length)
    return width if width else 1


# --- value classes (_fastani.pyx:1271-1428) -----------------------------------------------------------------------
@cython.freelist(16)
cdef class Hit:
    """A single hit found when querying a `Mapper` with a genome (_fastani.pyx:1271-1324)."""
    cdef readonly object name
    cdef readonly int64_t matches
    cdef readonly int64_t fragments
    cdef readonly float identity

    def __init__(self, object name, float identity, int64_t matches, int64_t fragments):
        self.name = name
        self.matches = matches
        self.fragments = fragments
        self.identity = identity

    def __repr__(self):
        return "{}(name={!r}, identity={!r}, matches={!r}, fragments={!r})".format(
            type(self).__name__, self.name, self.identity, self.matches, self.fragments)

    def __eq__(self, Hit other):
        return (self.name == other.name and self.matches == other.matches and self.fragments == other.fragments
                and self.identity == other.identity)

    __hash__ = None

    def __reduce__(self):
        return (Hit, (self.name, self.identity, self.matches, self.fragments))


cdef class MinimizerInfo:
    """The information about a single minimizer (_fastani.pyx:1327-1379)."""
    cdef readonly uint32_t hash
    cdef readonly int32_t sequence_id
    cdef readonly int32_t window_position

    def __init__(self, uint32_t hash, int32_t sequence_id, int32_t window_position):
        self.hash = hash
        self.sequence_id = sequence_id
        self.window_position = window_position

    def __repr__(self):
        return "{}(hash={!r}, sequence_id={!r}, window_position={!r})".format(
            type(self).__name__, self.hash, self.sequence_id, self.window_position)

    def __eq__(self, MinimizerInfo other):
        return (self.hash == other.hash and self.sequence_id == other.sequence_id
                and self.window_position == other.window_position)

    __hash__ = None

    def __reduce__(self):
        return (MinimizerInfo, (self.hash, self.sequence_id, self.window_position))


cdef class Position:
    """A (sequence, window) position of a minimizer (_fastani.pyx:1382-1428)."""
    cdef readonly int32_t sequence_id
    cdef readonly int32_t window_position

    def __init__(self, int32_t sequence_id, int32_t window_position):
        self.sequence_id = sequence_id
        self.window_position = window_position

    def __repr__(self):
        return "{}(sequence_id={!r}, window_position={!r})".format(type(self).__name__, self.sequence_id, self.window_position)

    def __eq__(self, Position other):
        return self.sequence_id == other.sequence_id and self.window_position == other.window_position

    __hash__ = None

    def __reduce__(self):
        return (Position, (self.sequence_id, self.window_position))


class Minimizers:
    """A read-only view over the minimizers of a `Sketch` or a `Mapper` (_fastani.pyx:1203-1268).

    The records live in HBM; they are read back once per owner state and cached."""

    def __init__(self, owner=None):
        self._owner = owner
        self._cache = None
        self._detached = None  # (hashes, ids, offsets) when unpickled on its own

    def _arrays(self):
        import numpy as np
        if self._owner is None:
            if self._detached is None:
                z = np.zeros(0, np.uint32)
                return z, z.astype(np.int32), z.astype(np.int32)
            return self._detached
        token = self._owner._state_token()
        if self._cache is None or self._cache[0] != token:
            self._cache = (token, self._owner._read_minimizers())
        return self._cache[1]

    def __len__(self):
        if self._owner is not None:
            return self._owner._num_minimizers()
        return len(self._arrays()[0])

    def __getitem__(self, index):
        h, s, w = self._arrays()
        n = len(h)
        i = operator.index(index)
        if i < 0:
            i += n
        if i < 0 or i >= n:
            raise IndexError(index)
        return MinimizerInfo(int(h[i]), int(s[i]), int(w[i]))

    def __getstate__(self):
        h, s, w = self._arrays()
        return {"hashes": h.tolist(), "ids": s.tolist(), "offsets": w.tolist(), "length": len(h)}

    def __setstate__(self, state):
        import numpy as np
        n = state["length"]
        self._owner = None
        self._cache = None
        self._detached = (
            np.asarray(state["hashes"][:n], dtype=np.uint32),
            np.asarray(state["ids"][:n], dtype=np.int32),
            np.asarray(state["offsets"][:n], dtype=np.int32),
        )


cdef class MinimizerIndex:
    """The index mapping minimizer hash values to their positions (_fastani.pyx:1431-1539): a dict-like, read-only view
    over the device-resident lookup index of a `Mapper`."""
    cdef readonly object owner
    cdef dict _own            # stand-alone instances (the reference allows constructing an empty index)

    def __init__(self, owner=None):
        self.owner = owner
        self._own = {}

    cdef hip.fa_mapper* _hm(self) except NULL:
        cdef Mapper m = <Mapper> self.owner
        if m._hm == NULL:
            raise RuntimeError("the Mapper was released")
        return m._hm

    def __len__(self):
        cdef int64_t n = 0
        if self.owner is None:
            return len(self._own)
        _check(hip.fa_mapper_lookup_size(self._hm(), &n))
        return n

    def _keys(self):
        import numpy as np
        n = len(self)
        keys = np.empty(n, np.uint32)
        cdef uintptr_t p = keys.ctypes.data
        if n:
            _check(hip.fa_mapper_lookup_keys(self._hm(), <uint32_t*> p))
        return keys

    def __iter__(self):
        if self.owner is None:
            return iter(list(self._own))
        return iter(self._keys().tolist())

    def __contains__(self, item):
        cdef int64_t n = 0
        item = _as_uint(item, "item", 32)
        if self.owner is None:
            return item in self._own
        _check(hip.fa_mapper_lookup_count(self._hm(), <uint32_t> item, &n))
        return n >= 0

    def __getitem__(self, item):
        cdef int64_t n = 0
        cdef vector[int32_t] seq, pos
        cdef int64_t i
        item = _as_uint(item, "item", 32)
        if self.owner is None:
            return list(self._own[item])
        _check(hip.fa_mapper_lookup_count(self._hm(), <uint32_t> item, &n))
        if n < 0:
            raise KeyError(item)
        seq.resize(max(n, 1))
        pos.resize(max(n, 1))
        _check(hip.fa_mapper_lookup_get(self._hm(), <uint32_t> item, seq.data(), pos.data(), n))
        return [Position(seq[i], pos[i]) for i in range(n)]

    def __setitem__(self, item, value):
        if self.owner is not None:
            raise TypeError("the lookup index of a Mapper lives in device memory and is read-only")
        self._own[_as_uint(item, "item", 32)] = [Position(p.sequence_id, p.window_position) for p in value]

    def __delitem__(self, item):
        if self.owner is not None:
            raise TypeError("the lookup index of a Mapper lives in device memory and is read-only")
        del self._own[_as_uint(item, "item", 32)]

    def items(self):
        for key in self:
            yield key, self[key]

    def __reduce__(self):
        return (MinimizerIndex, (), None, None, self.items())


# --- _Parameterized (_fastani.pyx:364-446) --------------------------------------------------------------------------
cdef class _Parameterized:
    cdef hip.fa_params _p
    cdef int _threads

    def __cinit__(self):
        self._p.kmer_size = 16
        self._p.window_size = 24
        self._p.fragment_length = 3000
        self._p.alphabet_size = 4
        self._p.min_fraction = 0.2
        self._p.percentage_identity = 80.0
        self._p.p_value = 1e-3
        self._p.reference_size = 5000000
        self._threads = 1

    cdef dict _params_getstate(self):
        return {
            "kmerSize": self._p.kmer_size,
            "windowSize": self._p.window_size,
            "minReadLength": self._p.fragment_length,
            "minFraction": self._p.min_fraction,
            "threads": self._threads,
            "alphabetSize": self._p.alphabet_size,
            "referenceSize": self._p.reference_size,
            "percentageIdentity": self._p.percentage_identity,
            "p_value": self._p.p_value,
        }

    cdef int _params_setstate(self, dict state) except -1:
        self._p.kmer_size = state["kmerSize"]
        self._p.window_size = state["windowSize"]
        self._p.fragment_length = state["minReadLength"]
        self._p.min_fraction = state["minFraction"]
        self._threads = state["threads"]
        self._p.alphabet_size = state["alphabetSize"]
        self._p.reference_size = state["referenceSize"]
        self._p.percentage_identity = state["percentageIdentity"]
        self._p.p_value = state["p_value"]
        return 0

    @property
    def _param(self):
        """The parameters as the ``ctypes`` structure of ``pyfastani_amd._lib`` (debug entry points take it by address)."""
        from . import _lib
        return _lib.Params(self._p.kmer_size, self._p.window_size, self._p.fragment_length, self._p.alphabet_size,
                           self._p.min_fraction, self._p.percentage_identity, self._p.p_value, self._p.reference_size)

    @property
    def k(self):
        """`int`: The k-mer size used for sketching."""
        return self._p.kmer_size

    @property
    def window_size(self):
        """`int`: The window size used for sketching."""
        return self._p.window_size

    @property
    def fragment_length(self):
        """`int`: The minimum read length to use for mapping."""
        return self._p.fragment_length

    @property
    def minimum_fraction(self):
        """`float`: The minimum genome fraction required to trust ANI values."""
        return self._p.min_fraction

    @property
    def percentage_identity(self):
        """`float`: The identity threshold for similarity when estimating hits."""
        return self._p.percentage_identity

    @property
    def p_value(self):
        """`float`: The p-value threshold for similarity when estimating hits."""
        return self._p.p_value

    @property
    def protein(self):
        """`bool`: Whether or not the object expects peptides or nucleotides."""
        return self._p.alphabet_size == 20


cdef tuple _download_minimizers_sketch(hip.fa_sketch* s):
    import numpy as np
    cdef int64_t n = 0
    _check(hip.fa_sketch_num_minimizers(s, &n))
    h = np.empty(n, np.uint32)
    q = np.empty(n, np.int32)
    w = np.empty(n, np.int32)
    cdef uintptr_t ph = h.ctypes.data, pq = q.ctypes.data, pw = w.ctypes.data
    if n:
        _check(hip.fa_sketch_get_minimizers(s, <uint32_t*> ph, <int32_t*> pq, <int32_t*> pw))
    return h, q, w


# --- FASTA files packed once (host ingest; no reference analogue) -----------------------------------------------------------------
cdef class PackedGenomes:
    """FASTA files (one genome each) read and 2-bit packed ONCE by the library's host threads, to be used as references
    (`Sketch.add_packed`) and as queries (`Mapper.query_fasta_stream(packed)`): an all-vs-all reads every file one time."""
    cdef hip.fa_packed* _hp
    cdef readonly list paths
    cdef readonly bint protein

    def __cinit__(self):
        self._hp = NULL

    def __init__(self, paths, protein=False):
        cdef vector[const char*] arr
        cdef int code, prot = 1 if protein else 0
        cdef int32_t n
        self.paths = [os.fspath(p) for p in paths]
        self.protein = bool(protein)
        encoded = [os.fsencode(p) for p in self.paths]
        for p in encoded:
            arr.push_back(<const char*> p)
        n = <int32_t> len(encoded)
        if arr.empty():
            arr.push_back(NULL)
        with nogil:
            code = hip.fa_packed_read(arr.data(), n, prot, &self._hp)
        _check(code)

    def __dealloc__(self):
        if self._hp != NULL:
            hip.fa_packed_free(self._hp)
            self._hp = NULL

    def __len__(self):
        return len(self.paths)

    def extend(self, paths):
        """Read and pack more files behind the ones the set holds (all of them, or none if one fails)."""
        cdef vector[const char*] arr
        cdef int code
        cdef int32_t n
        more = [os.fspath(p) for p in paths]
        encoded = [os.fsencode(p) for p in more]
        for p in encoded:
            arr.push_back(<const char*> p)
        n = <int32_t> len(encoded)
        if arr.empty():
            arr.push_back(NULL)
        with nogil:
            code = hip.fa_packed_append(self._hp, arr.data(), n)
        _check(code)
        self.paths.extend(more)
        return self

    def info(self):
        """Per file: ``(bytes, records, bases)`` as three lists."""
        cdef vector[uint64_t] fb
        cdef vector[int64_t] rec, bases
        cdef int32_t n = 0
        cdef size_t k = max(len(self.paths), 1)
        fb.resize(k); rec.resize(k); bases.resize(k)
        _check(hip.fa_packed_info(self._hp, &n, fb.data(), rec.data(), bases.data()))
        return [fb[i] for i in range(n)], [rec[i] for i in range(n)], [bases[i] for i in range(n)]


# --- Sketch (_fastani.pyx:449-806) ------------------------------------------------------------------------------------
cdef class Sketch(_Parameterized):
    """An index computing minimizers over the reference genomes.

    Use `add_genome` / `add_draft` to add reference genomes, then `index` to obtain a `Mapper`.  Minimizers are
    extracted on the GPU, lazily: contigs are packed to 2 bits per base when added and sketched in one batch when the
    minimizers are first needed (``len(sketch.minimizers)``, `index`)."""
    cdef hip.fa_sketch* _hs                 # instead of the reference's `Sketch_t* _sk`
    cdef list _names
    cdef int64_t _version
    cdef readonly object minimizers
    cdef readonly object _lock

    def __cinit__(self):
        self._hs = NULL
        self._names = []
        self._version = 0
        self._lock = threading.Lock()
        self.minimizers = Minimizers(self)

    def __init__(self, *, k=16, fragment_length=3000, minimum_fraction=0.2, p_value=1e-03, percentage_identity=80.0,
                 reference_size=5_000_000, protein=False):
        cdef int w = 0
        k = _as_uint(k, "k", 32)
        fragment_length = _as_uint(fragment_length, "fragment_length", 32)
        minimum_fraction = _as_float(minimum_fraction, "minimum_fraction")
        p_value = _as_float(p_value, "p_value")
        percentage_identity = _as_float(percentage_identity, "percentage_identity")
        reference_size = _as_uint(reference_size, "reference_size", 64)
        # _fastani.pyx:523-539
        if minimum_fraction > 1 or minimum_fraction < 0:
            raise ValueError(f"minimum_fraction must be between 0 and 1, got {minimum_fraction!r}")
        if fragment_length <= 0:
            raise ValueError(f"fragment_length must be strictly positive, got {fragment_length!r}")
        if p_value <= 0:
            raise ValueError(f"p_value must be positive, got {p_value!r}")
        if percentage_identity > 100 or percentage_identity < 0:
            raise ValueError(f"percentage_identity must be between 0 and 100, got {percentage_identity!r}")
        if k <= 0:
            raise ValueError(f"k must be strictly positive, got {k!r}")
        elif k > MAX_KMER_SIZE:
            raise BufferError(f"k must be smaller than {MAX_KMER_SIZE}, got {k}")
        elif k > 16:
            warnings.warn(f"Using k-mer size greater than 16 ({k!r}), accuracy will be degraded.", UserWarning)
        if fragment_length >= 1 << 31:
            raise OverflowError("fragment_length too large")
        self._p.kmer_size = k
        self._p.fragment_length = fragment_length
        self._p.min_fraction = minimum_fraction
        self._p.p_value = p_value
        self._p.percentage_identity = percentage_identity
        self._p.reference_size = reference_size
        if protein:
            self._p.alphabet_size = 20
            self._p.window_size = 1
        else:
            self._p.alphabet_size = 4
            _check(hip.fa_recommended_window_size(self._p.p_value, self._p.kmer_size, 4, self._p.percentage_identity,
                                                  self._p.fragment_length, self._p.reference_size, &w))   # :553-560
            # the reference reads an uninitialised sketch size when no candidate meets the p-value cut-off
            # (e.g. fragment_length=100); clamp to the largest admissible window instead of emulating UB
            self._p.window_size = w if w > 0 else self._p.fragment_length
        self._release()
        self._names = []
        self._version = 0
        _check(hip.fa_sketch_new(&self._p, &self._hs))                   # `new Sketch_t(param)`, :476

    cdef void _release(self) noexcept:
        if self._hs != NULL:
            hip.fa_sketch_free(self._hs)
            self._hs = NULL

    def __dealloc__(self):
        self._release()                                                      # :569-570

    @property
    def _h(self):
        """Address of the ``fa_sketch`` handle (for the ctypes debug entry points of ``pyfastani_amd._lib``)."""
        return <uintptr_t> self._hs

    def _state_token(self):
        return ("sketch", self._version)

    def _num_minimizers(self):
        cdef int64_t n = 0
        _check(hip.fa_sketch_num_minimizers(self._hs, &n))
        return n

    def _read_minimizers(self):
        return _download_minimizers_sketch(self._hs)

    cdef tuple _host_state(self):
        import numpy as np
        cdef int64_t n = 0, counter = 0
        _check(hip.fa_sketch_num_genomes(self._hs, &n))
        lengths = np.zeros(n, np.uint64)
        sbf = np.zeros(n, np.int32)
        cdef uintptr_t pl = lengths.ctypes.data, ps = sbf.ctypes.data
        _check(hip.fa_sketch_get_state(self._hs, <uint64_t*> pl, <int32_t*> ps, &counter))
        return lengths, sbf, counter

    # -- pickling (_fastani.pyx:572-591) -----------------------------------------------------------------------
    def __getstate__(self):
        lengths, sbf, counter = self._host_state()
        return {
            "parameters": self._params_getstate(),
            "counter": counter,
            "lengths": lengths.tolist(),
            "names": list(self._names),
            "sketch": {"sequencesByFileInfo": sbf.tolist(), "minimizers": self.minimizers.__getstate__()},
        }

    def __setstate__(self, state):
        import numpy as np
        self._params_setstate(state["parameters"])
        self._release()
        self._version = 0
        self._names = list(state["names"])
        _check(hip.fa_sketch_new(&self._p, &self._hs))
        mins = state["sketch"]["minimizers"]
        cdef int64_t n = mins["length"]
        lengths = np.ascontiguousarray(state["lengths"], dtype=np.uint64)
        sbf = np.ascontiguousarray(state["sketch"]["sequencesByFileInfo"], dtype=np.int32)
        h = np.ascontiguousarray(mins["hashes"][:n], dtype=np.uint32)
        s = np.ascontiguousarray(mins["ids"][:n], dtype=np.int32)
        w = np.ascontiguousarray(mins["offsets"][:n], dtype=np.int32)
        cdef uintptr_t pl = lengths.ctypes.data, ps = sbf.ctypes.data, ph = h.ctypes.data, pq = s.ctypes.data, pw = w.ctypes.data
        cdef int64_t counter = state["counter"]
        _check(hip.fa_sketch_set_state(self._hs, len(lengths), <const uint64_t*> pl, <const int32_t*> ps, counter, n,
                                       <const uint32_t*> ph, <const int32_t*> pq, <const int32_t*> pw))

    def __reduce__(self):
        return (_unpickle_sketch, (self.__getstate__(),))

    # -- record exchange for the multi-GPU index build (pyfastani_amd.sharding, SURVEY.md 8e) --------------------
    def _export_records(self, device):
        """Minimizer records as one ``int32`` torch tensor ``[3, n]`` (hash bits, contig id, window position) on
        ``device`` plus the host-side state ``(lengths, sequencesByFileInfo, counter)``.  On a CUDA/HIP device the
        records are copied HBM to HBM."""
        import numpy as np
        import torch
        cdef int64_t n = self._num_minimizers()
        cdef uintptr_t base
        cdef int64_t row
        lengths, sbf, counter = self._host_state()
        dev = torch.device(device)
        if dev.type == "cuda":
            rec = torch.empty((3, max(n, 1)), dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)
            base, row = rec.data_ptr(), rec.stride(0) * 4
            _check(hip.fa_sketch_get_minimizers_device(self._hs, rec.shape[1], <uint32_t*> base, <int32_t*> (base + row),
                                                       <int32_t*> (base + 2 * row)))
            rec = rec[:, :n]
        else:
            h, sq, w = self._read_minimizers()
            rec = torch.from_numpy(np.stack([h.view(np.int32), sq, w]))
        return rec, (lengths, sbf, counter)

    def _import_records(self, names, lengths, sbf, counter, rec):
        """Replace the content of this sketch by merged records (``rec`` as returned by `_export_records`)."""
        import numpy as np
        lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
        sbf = np.ascontiguousarray(sbf, dtype=np.int32)
        cdef int64_t n = int(rec.shape[1]), ctr = int(counter)
        cdef uintptr_t pl = lengths.ctypes.data, ps = sbf.ctypes.data, base, ph, pq, pw
        cdef int64_t row
        with self._lock:
            if rec.device.type == "cuda":
                import torch
                rec = rec.contiguous()
                torch.cuda.synchronize(rec.device)
                base, row = rec.data_ptr(), rec.stride(0) * 4
                _check(hip.fa_sketch_set_state_device(self._hs, len(lengths), <const uint64_t*> pl, <const int32_t*> ps, ctr, n,
                                                      <const uint32_t*> base, <const int32_t*> (base + row), <const int32_t*> (base + 2 * row)))
            else:
                a = np.ascontiguousarray(rec.numpy())
                h, sq, w = np.ascontiguousarray(a[0]).view(np.uint32), np.ascontiguousarray(a[1]), np.ascontiguousarray(a[2])
                ph, pq, pw = h.ctypes.data, sq.ctypes.data, w.ctypes.data
                _check(hip.fa_sketch_set_state(self._hs, len(lengths), <const uint64_t*> pl, <const int32_t*> ps, ctr, n,
                                               <const uint32_t*> ph, <const int32_t*> pq, <const int32_t*> pw))
            self._names = list(names)
            self._version += 1

    # -- properties ----------------------------------------------------------------------------------------------
    @property
    def occurences_threshold(self):
        """`int`: The occurence threshold above which minimizers are ignored (INT_MAX until indexed, :596-600)."""
        return _INT_MAX

    @property
    def names(self):
        """`list`: The names of the sequences currently sketched."""
        return self._names[:]

    # -- methods -------------------------------------------------------------------------------------------------
    cdef int _add_draft(self, object name, object contigs) except 1:
        # _fastani.pyx:610-690: one fa_sketch_add_contig per contig (addMinimizers runs on the device, lazily), then the
        # genome is closed.  An exception while reading a contig abandons the genome the way the reference's local
        # `total` does: the carried length is dropped, contigs already added stay (they count towards the next genome).
        cdef _Span sp
        cdef int added = 0
        cdef list keep
        try:
            for contig in contigs:
                keep = []
                _borrow(contig, &sp, keep)
                with nogil:
                    code = hip.fa_sketch_add_contig(self._hs, sp.data, sp.length, sp.kind, &added)
                _check(code)
                if not added:
                    warnings.warn("Sketch received a short contig relative to parameters, minimizers will not be added.",
                                  UserWarning)                                 # :670-677
        except BaseException:
            hip.fa_sketch_abort_genome(self._hs)
            raise
        self._names.append(name)
        _check(hip.fa_sketch_end_genome(self._hs))                             # :686-690
        self._version += 1
        return 0

    cpdef Sketch add_draft(self, object name, object contigs):
        """Add a reference draft genome to the sketcher (_fastani.pyx:692-717)."""
        with self._lock:
            self._add_draft(name, contigs)
        return self

    cpdef Sketch add_genome(self, object name, object sequence):
        """Add a reference genome to the sketcher (_fastani.pyx:719-744)."""
        with self._lock:
            self._add_draft(name, (sequence,))
        return self

    def add_drafts(self, names, genomes):
        """`add_draft` for many reference genomes at once: ``genomes[i]`` is the iterable of contigs of ``names[i]``.  One call of
        the library's packer over all contigs (``fa_sketch_add_genomes``) instead of one per genome -- the host side of a
        thousand-genome index drops from 0.4-0.5 s to the packing itself.  All contigs must share one character width (all
        bytes-like, or all ``str`` of one kind); mixed input goes genome by genome through `add_draft`."""
        cdef vector[const void*] ptrs
        cdef vector[int64_t] lens
        cdef vector[int32_t] cg, n_short
        cdef _Span sp
        cdef list keep = []
        cdef int width = 0, code
        cdef int32_t gi = 0
        cdef bint mixed = False
        names = list(names)
        genomes = [list(contigs) for contigs in genomes]
        if len(names) != len(genomes):
            raise ValueError("names and genomes differ in length")
        for contigs in genomes:
            for contig in contigs:
                _borrow(contig, &sp, keep)
                if sp.length > 0:
                    if width == 0:
                        width = sp.kind
                    elif width != sp.kind:
                        mixed = True
                ptrs.push_back(sp.data); lens.push_back(sp.length); cg.push_back(gi)
            gi += 1
        if mixed:
            for name, contigs in zip(names, genomes):
                self.add_draft(name, contigs)
            return self
        if width == 0:
            width = 1
        n_short.resize(max(gi, 1))
        if ptrs.empty():
            ptrs.push_back(NULL); lens.push_back(0); cg.push_back(0)
        cdef int64_t n = <int64_t> len(keep)
        with self._lock:
            with nogil:
                code = hip.fa_sketch_add_genomes(self._hs, ptrs.data(), lens.data(), cg.data(), n, gi, width, n_short.data())
            _check(code)
            self._names.extend(names)
            self._version += 1
        for i in range(gi):
            for _ in range(n_short[i]):
                warnings.warn("Sketch received a short contig relative to parameters, minimizers will not be added.", UserWarning)
        return self

    def add_fasta(self, name, path):
        """Add every record of a FASTA file as the contigs of ONE reference genome (`add_draft` semantics), read and
        packed natively without building Python objects (host ingest, SURVEY.md 8f-2)."""
        cdef int64_t n_rec = 0, n_short = 0
        cdef bytes p = os.fsencode(path)
        cdef const char* cp = p
        with self._lock:
            with nogil:
                code = hip.fa_sketch_add_fasta(self._hs, cp, &n_rec, &n_short)
            _check(code)
            self._names.append(name)
            self._version += 1
        for _ in range(n_short):
            warnings.warn("Sketch received a short contig relative to parameters, minimizers will not be added.", UserWarning)
        return self

    def add_fasta_many(self, names, paths):
        """`add_fasta` for many reference genomes at once (one per file, in order): the files are read and packed
        concurrently by the library's host threads -- the native form of the reference benchmark's loading loop
        (``benches/mapping/bench.py:41-47``)."""
        cdef vector[const char*] arr
        cdef vector[int64_t] n_rec, n_short
        cdef int code
        cdef int32_t n
        names = list(names)
        encoded = [os.fsencode(p) for p in paths]
        if len(names) != len(encoded):
            raise ValueError("names and paths differ in length")
        for p in encoded:
            arr.push_back(<const char*> p)
        n = <int32_t> len(encoded)
        n_rec.resize(max(n, 1)); n_short.resize(max(n, 1))
        if arr.empty():
            arr.push_back(NULL)
        with self._lock:
            with nogil:
                code = hip.fa_sketch_add_fasta_many(self._hs, arr.data(), n, n_rec.data(), n_short.data())
            _check(code)
            self._names.extend(names)
            self._version += 1
        for i in range(n):
            for _ in range(n_short[i]):
                warnings.warn("Sketch received a short contig relative to parameters, minimizers will not be added.", UserWarning)
        return self

    def add_packed(self, names, PackedGenomes packed, int first=0, count=None):
        """Files ``[first, first + count)`` of a `PackedGenomes` as that many reference genomes (`add_fasta_many` without
        reading the files again)."""
        cdef vector[int64_t] n_rec, n_short
        cdef int code
        cdef int32_t c = <int32_t> (len(packed) - first if count is None else count)
        names = list(names)
        if len(names) != c:
            raise ValueError("names and the file range differ in length")
        n_rec.resize(max(c, 1)); n_short.resize(max(c, 1))
        with self._lock:
            with nogil:
                code = hip.fa_sketch_add_packed(self._hs, packed._hp, first, c, n_rec.data(), n_short.data())
            _check(code)
            self._names.extend(names)
            self._version += 1
        for i in range(c):
            for _ in range(n_short[i]):
                warnings.warn("Sketch received a short contig relative to parameters, minimizers will not be added.", UserWarning)
        return self

    def flush(self):
        """Sketch the contigs added so far on the device now (it happens by itself when the minimizers are first needed:
        `index`, `minimizers`, pickling).  Releases the GIL; other threads may go on adding genomes -- their host work (reading,
        packing) overlaps with the device work here, only the final append waits."""
        cdef int64_t n = 0
        cdef int code
        with nogil:
            code = hip.fa_sketch_num_minimizers(self._hs, &n)
        _check(code)
        return self

    def add_fasta_stream(self, names, paths, int chunk=128, stats=None, PackedGenomes keep=None):
        """`add_fasta_many` in chunks of `chunk` files with the device working behind the host: while the files of chunk c + 1 are
        read and packed by the host pool, a second thread has the device sketch chunk c (`flush`).  Same sketch as one
        `add_fasta_many` call; the reference-side half of a files-to-table run (``bench.py``: ``fasta_to_table``).  ``stats`` (a
        dict) receives ``add_s`` (the `add_fasta_many` calls: reading, packing and the wait for a sketch in flight before the
        append) and ``sketch_s`` (the `flush` calls on the second thread).  ``keep``: a `PackedGenomes` that receives the files as
        they are read (`PackedGenomes.extend`), so that they can be mapped later without being read again
        (``Mapper.query_fasta_stream(keep)``: an all-vs-all)."""
        import threading
        import time
        names, paths = list(names), list(paths)
        if len(names) != len(paths):
            raise ValueError("names and paths differ in length")
        chunk = max(1, chunk)
        worker, failure = None, []
        if stats is None:
            stats = {}
        stats.update(add_s=0.0, sketch_s=0.0, chunks=(len(paths) + chunk - 1) // chunk)

        def sketch_pending():
            t0 = time.perf_counter()
            try:
                self.flush()
            except BaseException as exc:      # re-raised on the calling thread
                failure.append(exc)
            stats["sketch_s"] += time.perf_counter() - t0

        for i in range(0, len(paths), chunk):
            t0 = time.perf_counter()
            if keep is None:
                self.add_fasta_many(names[i:i + chunk], paths[i:i + chunk])      # (its append waits for a flush in flight)
            else:
                at = len(keep)
                keep.extend(paths[i:i + chunk])
                self.add_packed(names[i:i + chunk], keep, at, len(paths[i:i + chunk]))
            stats["add_s"] += time.perf_counter() - t0
            if worker is not None:
                worker.join()
            if failure:
                raise failure[0]
            worker = threading.Thread(target=sketch_pending, name="pyfastani-amd-sketch", daemon=True)
            worker.start()
        if worker is not None:
            worker.join()
        if failure:
            raise failure[0]
        return self

    cpdef Sketch clear(self):
        """Reset the `Sketch`, removing any reference genome it may contain (_fastani.pyx:746-767)."""
        with self._lock:                       # (add_draft / add_genome / add_fasta hold it while they append contigs)
            self._names.clear()
            _check(hip.fa_sketch_clear(self._hs))
            self._version += 1
        return self

    cpdef Mapper index(self):
        """Index the reference genomes for fast lookups using the minimizers (_fastani.pyx:769-806).

        Ownership of the data moves to the returned `Mapper`; this `Sketch` is cleared but stays usable."""
        cdef Mapper mapper = Mapper.__new__(Mapper)
        cdef int code
        # under the lock of the add_* calls: a genome that another thread is half-way through adding must not be split
        # between the mapper and the emptied sketch, and the names move in the same critical section as the records
        with self._lock:
            with nogil:
                code = hip.fa_sketch_index(self._hs, &mapper._hm)            # Sketch_t::index + computeFreqHist, :790-791
            _check(code)
            mapper._p = self._p
            mapper._threads = self._threads
            mapper._names = self._names.copy()
            self._names.clear()
            self._version += 1
        return mapper


def _unpickle_sketch(state):
    cdef Sketch sk = Sketch.__new__(Sketch)
    sk.__setstate__(state)
    return sk


# --- Mapper (_fastani.pyx:809-1200) -----------------------------------------------------------------------------------
cdef class Mapper(_Parameterized):
    """A genome mapper using Murmur3 hashes and k-mers to compute ANI, resident on one MI355X."""
    cdef hip.fa_mapper* _hm                # instead of the reference's `Sketch_t* _sk` + `Map_t`
    cdef list _names
    cdef vector[uint64_t] _lengths
    cdef bint _have_lengths
    cdef int64_t _version                  # bumped by __setstate__: the `Minimizers` view caches per (object, version)
    cdef readonly object minimizers

    def __cinit__(self):
        self._hm = NULL
        self._names = []
        self._have_lengths = False
        self._version = 0
        self.minimizers = Minimizers(self)

    def __init__(self, *args, **kwargs):
        raise TypeError("Mapper cannot be instantiated, use `Sketch.index` instead.")   # :836-837

    def __dealloc__(self):
        if self._hm != NULL:
            hip.fa_mapper_free(self._hm)                                     # :839-840
            self._hm = NULL

    @property
    def _h(self):
        """Address of the ``fa_mapper`` handle (for the ctypes debug / timing entry points of ``pyfastani_amd._lib``)."""
        return <uintptr_t> self._hm

    def _state_token(self):
        return ("mapper", id(self), self._version)

    def _num_minimizers(self):
        cdef int64_t n = 0
        _check(hip.fa_mapper_num_minimizers(self._hm, &n))
        return n

    def _read_minimizers(self):
        import numpy as np
        cdef int64_t n = self._num_minimizers()
        h = np.empty(n, np.uint32)
        s = np.empty(n, np.int32)
        w = np.empty(n, np.int32)
        cdef uintptr_t ph = h.ctypes.data, ps = s.ctypes.data, pw = w.ctypes.data
        if n:
            _check(hip.fa_mapper_get_minimizers(self._hm, <uint32_t*> ph, <int32_t*> ps, <int32_t*> pw))
        return h, s, w

    def _state_arrays(self):
        import numpy as np
        cdef int64_t n = 0
        _check(hip.fa_mapper_num_genomes(self._hm, &n))
        lengths = np.zeros(n, np.uint64)
        sbf = np.zeros(n, np.int32)
        cdef uintptr_t pl = lengths.ctypes.data, ps = sbf.ctypes.data
        _check(hip.fa_mapper_get_state(self._hm, <uint64_t*> pl, <int32_t*> ps))
        return lengths, sbf

    cdef int _load_lengths(self) except -1:
        cdef int64_t n = 0
        cdef vector[int32_t] sbf
        if self._have_lengths:
            return 0
        _check(hip.fa_mapper_num_genomes(self._hm, &n))
        self._lengths.resize(max(n, 1))
        sbf.resize(max(n, 1))
        _check(hip.fa_mapper_get_state(self._hm, self._lengths.data(), sbf.data()))
        self._lengths.resize(n)
        self._have_lengths = True
        return 0

    @property
    def _genome_lengths(self):
        self._load_lengths()
        return [self._lengths[i] for i in range(self._lengths.size())]

    # -- pickling (_fastani.pyx:842-865): the index is rebuilt on load ------------------------------------------
    def __getstate__(self):
        lengths, sbf = self._state_arrays()
        return {
            "parameters": self._params_getstate(),
            "lengths": lengths.tolist(),
            "names": list(self._names),
            "sketch": {"sequencesByFileInfo": sbf.tolist(), "minimizers": self.minimizers.__getstate__()},
        }

    def __setstate__(self, state):
        cdef Sketch sk = Sketch.__new__(Sketch)
        sbf = state["sketch"]["sequencesByFileInfo"]
        sk.__setstate__({
            "parameters": state["parameters"],
            "counter": sbf[len(sbf) - 1] if sbf else 0,
            "lengths": state["lengths"],
            "names": state["names"],
            "sketch": state["sketch"],
        })
        cdef Mapper other = sk.index()
        if self._hm != NULL:
            hip.fa_mapper_free(self._hm)
        self._hm = other._hm
        other._hm = NULL
        self._p = other._p
        self._threads = other._threads
        self._names = other._names
        self._have_lengths = False
        self._version += 1

    def __reduce__(self):
        return (_unpickle_mapper, (self.__getstate__(),))

    # -- properties ------------------------------------------------------------------------------------------------
    @property
    def lookup_index(self):
        """`MinimizerIndex`: the table of minimizer positions in the reference genomes (_fastani.pyx:869-881)."""
        return MinimizerIndex(self)

    def _torch_device(self):
        import torch
        cdef int d = -1
        _check(hip.fa_mapper_device(self._hm, &d))
        return torch.device("cuda", d if d >= 0 else torch.cuda.current_device())

    def _export_lookup(self, device="cuda"):
        """Distinct hashes of this index (ascending, ``int32`` bit patterns) and the lengths of their position lists, as
        two torch tensors on ``device``: what a rank contributes to the global frequency threshold of a
        reference-sharded index (`sharding.global_frequency`)."""
        import torch
        cdef int64_t n = 0
        cdef uintptr_t pk, pc
        _check(hip.fa_mapper_lookup_size(self._hm, &n))
        dev = torch.device(device)
        own = self._torch_device()                                           # the GPU this mapper is bound to, not torch's current one
        keys = torch.empty(max(n, 1), dtype=torch.int32, device=own)
        counts = torch.empty(max(n, 1), dtype=torch.int32, device=own)
        torch.cuda.synchronize(own)
        pk, pc = keys.data_ptr(), counts.data_ptr()
        _check(hip.fa_mapper_lookup_export_device(self._hm, keys.shape[0], <uint32_t*> pk, <int32_t*> pc))
        return keys[:n].to(dev), counts[:n].to(dev)

    def _set_global_frequency(self, threshold, drop_keys):
        """Install the frequency threshold taken over all shards of a reference-sharded index and the hashes (torch
        ``int32`` bit patterns) whose summed list length reaches it."""
        import torch
        own = self._torch_device()
        drop = drop_keys.to(device=own, dtype=torch.int32).contiguous()
        torch.cuda.synchronize(own)
        cdef uintptr_t p = drop.data_ptr() if drop.numel() else 0
        _check(hip.fa_mapper_set_global_frequency(self._hm, int(threshold), int(drop.numel()), <const uint32_t*> p))

    @property
    def occurences_threshold(self):
        cdef int t = 0
        _check(hip.fa_mapper_freq_threshold(self._hm, &t))
        return t

    @property
    def names(self):
        return self._names[:]

    # -- queries -------------------------------------------------------------------------------------------------------
    cdef list _hits_of_rows(self, const hip.fa_cgi_row* rows, int64_t n_rows, uint64_t total_length):
        # _fastani.pyx:1121-1136; `uint64 >= uint64 * float` is evaluated in float exactly like the C expression
        cdef int64_t i
        cdef uint64_t min_length, shared_length
        cdef list hits = []
        self._load_lengths()
        for i in range(n_rows):
            min_length = min(total_length, self._lengths[rows[i].ref_genome_id])
            shared_length = <uint64_t> rows[i].count_seq * <uint64_t> self._p.fragment_length
            if <float> shared_length >= <float> min_length * self._p.min_fraction:
                hits.append(Hit(self._names[rows[i].ref_genome_id], rows[i].identity, rows[i].count_seq, rows[i].total_query_fragments))
        hits.sort(key=_hit_identity, reverse=True)                            # stable, :1135
        return hits

    def _rows_to_hits(self, rows, total_length):
        """`Hit` list of raw rows given as objects with the ``fa_cgi_row`` attributes (the batch path and the tests)."""
        cdef vector[hip.fa_cgi_row] buf
        cdef hip.fa_cgi_row r
        for x in rows:
            r.query_id = x.query_id
            r.ref_genome_id = x.ref_genome_id
            r.count_seq = x.count_seq
            r.total_query_fragments = x.total_query_fragments
            r.identity = x.identity
            buf.push_back(r)
        return self._hits_of_rows(buf.data(), <int64_t> buf.size(), total_length)

    cdef list _query_draft(self, object contigs, int threads=0):
        # _fastani.pyx:1006-1136.  `threads` is validated for signature compatibility; fragment-level parallelism is the
        # GPU's job (one call = one pass of the device pipeline over every fragment of the genome).
        cdef vector[const void*] ptrs
        cdef vector[int64_t] lens
        cdef vector[hip.fa_cgi_row] rows
        cdef list keep = []
        cdef int64_t n_rows = 0
        cdef int n_short = 0, width, code
        cdef uint64_t total_fragments = 0, total_length = 0
        if threads < 0:
            raise ValueError(f"`threads` must be positive or null, got {threads!r}")   # :1050
        width = _borrow_all(contigs, ptrs, lens, keep)
        rows.resize(max(len(self._names), 1))
        if ptrs.empty():
            ptrs.push_back(NULL)
            lens.push_back(0)
            with nogil:
                code = hip.fa_mapper_query(self._hm, ptrs.data(), lens.data(), 0, width, rows.data(), <int64_t> rows.size(),
                                           &n_rows, &n_short, &total_fragments, &total_length)
        else:
            with nogil:                                                       # re-entrant, GIL released: :1158-1161
                code = hip.fa_mapper_query(self._hm, ptrs.data(), lens.data(), <int> ptrs.size(), width, rows.data(),
                                           <int64_t> rows.size(), &n_rows, &n_short, &total_fragments, &total_length)
        _check(code)
        for _ in range(n_short):
            warnings.warn("Mapper received a short sequence relative to parameters, mapping will not be computed.",
                          UserWarning)                                        # :1063-1069
        return self._hits_of_rows(rows.data(), n_rows, total_length)

    def query_draft(self, object contigs, int threads=0):
        """Query the mapper for a complete genome given as contigs (_fastani.pyx:1138-1168)."""
        return self._query_draft(contigs, threads)

    def query_genome(self, object sequence, int threads=0):
        """Query the mapper for a complete, closed genome (_fastani.pyx:1170-1200)."""
        return self._query_draft((sequence,), threads)

    # -- many-to-many extension (no reference analogue): resident batches ------------------------------------------
    def upload_genomes(self, genomes):
        """Pack a list of draft genomes (each an iterable of contigs) into HBM and return a `GenomeBatch`."""
        return GenomeBatch(self, genomes)

    def upload_fasta(self, paths):
        """One query genome per FASTA file (its records are the contigs), read, packed and uploaded natively."""
        return GenomeBatch.from_fasta(self, paths)

    def query_fasta(self, path):
        """`query_draft` for a genome stored as a FASTA file (its records are the contigs)."""
        return self.upload_fasta([path]).query()[0]

    def query_fasta_stream(self, paths, chunk=None, rows=False, uintptr_t device_ptr=0, int64_t device_cap=0, stats=None):
        """Map the genomes stored in `paths` (one FASTA file each; or a `PackedGenomes`: files that are packed already -- the
        chunks are then refilled without reading anything) in chunks of `chunk` files (default: as many files as make
        ~120 MB of FASTA -- two dozen 5 Mb genomes, one device pass -- and at most 4096), yielding ``(first, result)``
        per chunk -- ``first`` is the number of the chunk's first genome in `paths`, ``result`` one hit list per genome, or the
        raw row array with ``rows=True`` (``query_id`` counts from 0 inside the chunk).  With ``device_ptr`` / ``device_cap`` (a caller-owned HBM table of 20-byte rows, e.g. a
        torch tensor) the rows never leave the device: chunk after chunk is written behind the rows of the chunks before it
        and ``result`` is ``(row offset, row count)``.

        While chunk c is mapped on the device, a second host thread reads, packs and uploads chunk c + 1 into the other of
        two recycled batches (``fa_genomes_reload_fasta``: its own stream, a pinned staging image), so the file-to-hits path
        is bound by the slower of the two sides, not by their sum.  ``stats`` (a dict) receives ``ingest_s`` (read + pack +
        upload, summed over the chunks, on the loader thread), ``map_s`` (the mapping calls) and ``wait_s`` (the consumer
        waiting for a chunk): the overlap is what ``ingest_s + map_s`` exceeds the wall clock by."""
        import queue
        import threading
        import time
        packed = paths if isinstance(paths, PackedGenomes) else None     # (files packed already: chunks are refilled without reading)
        sizes = None
        if packed is not None:
            sizes = packed.info()[0]
            paths = list(packed.paths)
        else:
            paths = list(paths)
        if chunk is None:
            chunks, cur, cur_bytes = [], [], 0
            for j, p in enumerate(paths):
                try:
                    size = sizes[j] if sizes is not None else os.path.getsize(p)
                except OSError:
                    size = 0                          # (the reader reports a missing file when its chunk is loaded)
                if cur and (cur_bytes + size > 120_000_000 or len(cur) >= 4096):
                    chunks.append(cur)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += size
            if cur:
                chunks.append(cur)
        else:
            chunk = max(1, int(chunk))
            chunks = [paths[i:i + chunk] for i in range(0, len(paths), chunk)]
        firsts = [0]
        for c in chunks:
            firsts.append(firsts[-1] + len(c))
        if stats is None:
            stats = {}
        stats.update(ingest_s=0.0, map_s=0.0, wait_s=0.0, chunks=len(chunks))
        if not chunks:
            return
        free = queue.Queue()
        ready = queue.Queue()
        cdef GenomeBatch b
        cdef int64_t written = 0, n_rows
        for _ in range(2):
            free.put(None)

        def loader():
            try:
                for i, c in enumerate(chunks):
                    slot = free.get()
                    if slot is False:                     # the consumer gave up
                        return
                    t0 = time.perf_counter()
                    if slot is None:
                        slot = GenomeBatch.from_fasta(self, [], True)       # an empty batch that is refilled from now on
                    if packed is not None:
                        slot.reload_packed(packed, firsts[i], len(c))
                    else:
                        slot.reload_fasta(c)
                    stats["ingest_s"] += time.perf_counter() - t0
                    ready.put((i, slot))
            except BaseException as exc:  # handed to the consumer
                ready.put((-1, exc))

        t = threading.Thread(target=loader, name="pyfastani-amd-ingest", daemon=True)
        t.start()
        try:
            for _ in range(len(chunks)):
                t0 = time.perf_counter()
                i, slot = ready.get()
                stats["wait_s"] += time.perf_counter() - t0
                if i < 0:
                    raise slot
                b = slot
                t0 = time.perf_counter()
                if device_ptr:
                    n_rows = b.query_rows_device(0, b.n_genomes, device_ptr + 20 * written, device_cap - written)
                    result = (written, n_rows)
                    written += n_rows
                else:
                    result = b.query_rows(0, b.n_genomes) if rows else b.query(0, b.n_genomes)
                stats["map_s"] += time.perf_counter() - t0
                free.put(slot)
                yield firsts[i], result
        finally:
            free.put(False); free.put(False)
            t.join(timeout=60)

    def query_batch(self, batch, first=0, count=None):
        """Map genomes ``[first, first+count)`` of a resident batch; returns one hit list per genome."""
        return batch.query(first, count)


def _hit_identity(Hit hit):
    return hit.identity


def _unpickle_mapper(state):
    cdef Mapper m = Mapper.__new__(Mapper)
    m.__setstate__(state)
    return m


# --- resident query batches ----------------------------------------------------------------------------------------------
cdef object _ROW_DTYPE = None


def _row_dtype():
    global _ROW_DTYPE
    if _ROW_DTYPE is None:
        import numpy as np
        _ROW_DTYPE = np.dtype([("query_id", "<i4"), ("ref_genome_id", "<i4"), ("count_seq", "<i4"),
                               ("total_query_fragments", "<i4"), ("identity", "<f4")])
        assert _ROW_DTYPE.itemsize == sizeof(hip.fa_cgi_row)
    return _ROW_DTYPE


cdef class GenomeBatch:
    """Many query genomes packed 2-bit in HBM, mapped without leaving the device: the many-to-many extension of the
    reference's one-query-at-a-time ``Mapper.query_draft`` (_fastani.pyx:1006-1136) -- the same per-genome semantics,
    but the inputs are uploaded once and any sub-range of genomes is mapped with one call."""
    cdef hip.fa_genomes* _hg
    cdef readonly Mapper _mapper
    cdef readonly int n_genomes
    cdef readonly object total_fragments
    cdef readonly object total_length
    cdef readonly object n_short

    def __cinit__(self):
        self._hg = NULL

    def __init__(self, Mapper mapper, genomes):
        cdef vector[const void*] ptrs
        cdef vector[int64_t] lens
        cdef vector[int32_t] cg
        cdef list keep = []
        cdef int width = 0, w, code
        cdef size_t before
        cdef int32_t gi = 0
        self._mapper = mapper
        for contigs in genomes:
            before = ptrs.size()
            w = _borrow_all(contigs, ptrs, lens, keep)
            has_data = False
            for j in range(before, ptrs.size()):
                cg.push_back(gi)
                if lens[j] > 0:
                    has_data = True
            if has_data:
                if width == 0:
                    width = w
                elif width != w:
                    raise ValueError("all genomes of a batch must use the same character width (all bytes, or all str)")
            gi += 1
        self.n_genomes = gi
        if width == 0:
            width = 1
        cdef int64_t n = <int64_t> ptrs.size()
        if n == 0:
            ptrs.push_back(NULL)
            lens.push_back(0)
            cg.push_back(0)
        with nogil:
            code = hip.fa_genomes_upload(mapper._hm, ptrs.data(), lens.data(), cg.data(), n, gi, width, &self._hg)
        _check(code)
        self._finish()

    @classmethod
    def from_fasta(cls, Mapper mapper, paths, recyclable=False):
        """One genome per FASTA file, parsed and packed by the library (``fa_genomes_upload_fasta``).  ``recyclable``: the
        batch is created empty and filled through `reload_fasta`, i.e. with a pinned staging image that later refills reuse."""
        cdef GenomeBatch self = GenomeBatch.__new__(GenomeBatch)
        cdef vector[const char*] arr
        cdef int code
        self._mapper = mapper
        if recyclable:
            arr.push_back(NULL)
            self.n_genomes = 0
            with nogil:
                code = hip.fa_genomes_upload_fasta(mapper._hm, arr.data(), 0, &self._hg)
            _check(code)
            self.reload_fasta(paths)
            return self
        encoded = [os.fsencode(p) for p in paths]
        for p in encoded:
            arr.push_back(<const char*> p)
        self.n_genomes = len(encoded)
        if arr.empty():
            arr.push_back(NULL)
        with nogil:
            code = hip.fa_genomes_upload_fasta(mapper._hm, arr.data(), self.n_genomes, &self._hg)
        _check(code)
        self._finish()
        return self

    def reload_packed(self, PackedGenomes packed, int first, int count):
        """Replace the contents of the batch by files ``[first, first + count)`` of a `PackedGenomes` (`reload_fasta`
        without reading the files again)."""
        cdef int code
        with nogil:
            code = hip.fa_genomes_reload_packed(self._mapper._hm, self._hg, packed._hp, first, count)
        if code != 0:
            self.n_genomes = 0
        _check(code)
        self.n_genomes = count
        self._finish()
        return self

    def reload_fasta(self, paths):
        """Replace the contents of the batch by the genomes of other FASTA files (one per file), recycling its device
        buffers, pinned staging image and upload stream (``fa_genomes_reload_fasta``)."""
        cdef vector[const char*] arr
        cdef int code
        cdef int32_t n
        encoded = [os.fsencode(p) for p in paths]
        for p in encoded:
            arr.push_back(<const char*> p)
        n = <int32_t> len(encoded)
        if arr.empty():
            arr.push_back(NULL)
        with nogil:
            code = hip.fa_genomes_reload_fasta(self._mapper._hm, self._hg, arr.data(), n)
        if code != 0:
            self.n_genomes = 0
        _check(code)
        self.n_genomes = n
        self._finish()
        return self

    cdef int _finish(self) except -1:
        import numpy as np
        cdef int32_t ng = 0
        self.total_fragments = np.zeros(self.n_genomes, np.uint64)
        self.total_length = np.zeros(self.n_genomes, np.uint64)
        self.n_short = np.zeros(self.n_genomes, np.int32)
        cdef uintptr_t pf = self.total_fragments.ctypes.data, pl = self.total_length.ctypes.data, ps = self.n_short.ctypes.data
        _check(hip.fa_genomes_info(self._hg, &ng, <uint64_t*> pf, <uint64_t*> pl, <int32_t*> ps))
        for _ in range(int(self.n_short.sum())):
            warnings.warn("Mapper received a short sequence relative to parameters, mapping will not be computed.", UserWarning)
        return 0

    def __dealloc__(self):
        if self._hg != NULL:
            hip.fa_genomes_free(self._hg)
            self._hg = NULL

    @property
    def _h(self):
        return <uintptr_t> self._hg

    def __len__(self):
        return self.n_genomes

    def query_rows(self, int first=0, count=None):
        """Raw cgi::CGI_Results rows (structured numpy array) for genomes [first, first+count)."""
        import numpy as np
        cdef int c = self.n_genomes - first if count is None else count
        cdef int64_t cap = max(1, <int64_t> c * max(1, len(self._mapper._names)))
        cdef int64_t n_rows = 0
        cdef int code
        rows = np.zeros(cap, dtype=_row_dtype())
        cdef uintptr_t p = rows.ctypes.data
        with nogil:
            code = hip.fa_mapper_query_genomes(self._mapper._hm, self._hg, first, c, <hip.fa_cgi_row*> p, cap, &n_rows, 0)
        _check(code)
        return rows[:n_rows]

    def query_rows_device(self, int first, int count, uintptr_t device_ptr, int64_t cap):
        """Same, but the rows are written to a caller-owned DEVICE buffer (e.g. a torch tensor feeding an RCCL all-gather).
        The library writes on its own stream: whatever the caller queued on the buffer (its allocation's memset, say)
        must have completed (``torch.cuda.synchronize()``) before the call, and the rows are complete when it returns."""
        cdef int64_t n_rows = 0
        cdef int code
        with nogil:
            code = hip.fa_mapper_query_genomes(self._mapper._hm, self._hg, first, count, <hip.fa_cgi_row*> device_ptr, cap, &n_rows, 1)
        _check(code)
        return n_rows

    def query(self, int first=0, count=None):
        """One sorted hit list per genome, exactly what ``Mapper.query_draft`` returns for each."""
        cdef int c = self.n_genomes - first if count is None else count
        cdef int64_t i, lo, n
        cdef uintptr_t p
        rows = self.query_rows(first, c)
        n = len(rows)
        p = rows.ctypes.data if n else 0
        cdef const hip.fa_cgi_row* r = <const hip.fa_cgi_row*> p
        out = [[] for _ in range(c)]
        lo = 0
        # rows come grouped by query genome, in (query, reference) order
        while lo < n:
            i = lo
            while i < n and r[i].query_id == r[lo].query_id:
                i += 1
            out[r[lo].query_id - first] = self._mapper._hits_of_rows(r + lo, i - lo, int(self.total_length[r[lo].query_id]))
            lo = i
        return out

"""Host-side mirror of pyfastani's Python surface, backed by the HIP engine.

Every class and method keeps the reference's name, argument meaning and error behaviour
(``src/pyfastani/_fastani.pyx`` of the reference; line numbers cited per method).  The compute is
done by ``libfastani_hip.so`` through the C ABI of ``include/fastani_hip.h``; this module only
validates arguments, borrows buffers, keeps the genome names and turns rows into ``Hit`` objects.
"""
import ctypes as C
import operator
import threading
import warnings

import numpy as np

from . import _lib
from ._lib import lib, check

MAX_KMER_SIZE = 2048  # _fastani.pyx:103-107


# --------------------------------------------------------------------------------------------------
# argument coercion with the exceptions Cython's typed signatures raise (test_sketch.py:12-23)
# --------------------------------------------------------------------------------------------------
def _as_uint(value, name, bits):
    try:
        v = operator.index(value)
    except TypeError:
        raise TypeError(f"an integer is required for {name!r}, got {type(value).__name__}") from None
    if v < 0:
        raise OverflowError(f"can't convert negative value to unsigned int ({name})")
    if v >= 1 << bits:
        raise OverflowError(f"value too large to convert to unsigned int ({name})")
    return v


def _as_float(value, name):
    if isinstance(value, (str, bytes, bytearray)) or not hasattr(value, "__float__"):
        raise TypeError(f"a float is required for {name!r}, got {type(value).__name__}")
    return float(value)


def _borrow(contig):
    """(address, length, char_width, keepalive) of a contig given as ``str`` or a byte buffer.

    Mirrors _fastani.pyx:633-645 / :1073-1092: ``str`` is read through its canonical UCS1/2/4
    representation, anything else must expose a C-contiguous ``unsigned char`` buffer.
    """
    if isinstance(contig, str):
        n = len(contig)
        if n == 0:
            return 0, 0, 1, None
        maxc = max(map(ord, contig)) if not contig.isascii() else 0x7F
        if maxc < 256:
            b = contig.encode("latin-1")
            return C.cast(C.c_char_p(b), C.c_void_p).value, n, 1, b
        if maxc < 65536:
            a = np.frombuffer(contig.encode("utf-16-le", "surrogatepass"), dtype=np.uint16)
            return a.ctypes.data, n, 2, a
        a = np.frombuffer(contig.encode("utf-32-le", "surrogatepass"), dtype=np.uint32)
        return a.ctypes.data, n, 4, a
    mv = memoryview(contig)
    if mv.ndim != 1 or mv.itemsize != 1 or not mv.c_contiguous:
        raise BufferError("contig must be a C-contiguous buffer of bytes")
    if mv.format not in ("B", "b", "c"):
        raise ValueError(f"Buffer dtype mismatch, expected 'const unsigned char' but got format {mv.format!r}")
    a = np.frombuffer(mv, dtype=np.uint8)
    return (a.ctypes.data if a.size else 0), a.size, 1, (a, mv)


def _borrow_all(contigs):
    """Borrow every contig of a genome with one common character width."""
    contigs = list(contigs)
    bufs = [_borrow(c) for c in contigs]
    widths = {b[2] for b in bufs if b[1] > 0}
    if len(widths) > 1:
        # mixed str kinds: widen everything to UCS4 (same characters, _fastani.pyx:144-148)
        bufs = []
        for c in contigs:
            text = c if isinstance(c, str) else bytes(memoryview(c)).decode("latin-1")
            a = np.frombuffer(text.encode("utf-32-le", "surrogatepass"), dtype=np.uint32)
            bufs.append((a.ctypes.data if a.size else 0, len(text), 4, a))
        return bufs, 4
    return bufs, (widths.pop() if widths else 1)


# --------------------------------------------------------------------------------------------------
# value classes (_fastani.pyx:1271-1428)
# --------------------------------------------------------------------------------------------------
class Hit:
    """A single hit found when querying a `Mapper` with a genome (_fastani.pyx:1271-1324)."""

    __slots__ = ("name", "matches", "fragments", "identity")

    def __init__(self, name, identity, matches, fragments):
        self.name = name
        self.matches = int(matches)
        self.fragments = int(fragments)
        self.identity = float(np.float32(identity))  # C `float` attribute

    def __repr__(self):
        return "{}(name={!r}, identity={!r}, matches={!r}, fragments={!r})".format(
            type(self).__name__, self.name, self.identity, self.matches, self.fragments
        )

    def __eq__(self, other):
        if not isinstance(other, Hit):
            raise TypeError(f"Argument 'other' has incorrect type (expected Hit, got {type(other).__name__})")
        return (
            self.name == other.name
            and self.matches == other.matches
            and self.fragments == other.fragments
            and self.identity == other.identity
        )

    __hash__ = None

    def __reduce__(self):
        return (Hit, (self.name, self.identity, self.matches, self.fragments))


class MinimizerInfo:
    """The information about a single minimizer (_fastani.pyx:1327-1379)."""

    __slots__ = ("hash", "sequence_id", "window_position")

    def __init__(self, hash, sequence_id, window_position):
        self.hash = int(hash)
        self.sequence_id = int(sequence_id)
        self.window_position = int(window_position)

    def __repr__(self):
        return "{}(hash={!r}, sequence_id={!r}, window_position={!r})".format(
            type(self).__name__, self.hash, self.sequence_id, self.window_position
        )

    def __eq__(self, other):
        if not isinstance(other, MinimizerInfo):
            raise TypeError("expected MinimizerInfo")
        return (self.hash, self.sequence_id, self.window_position) == (
            other.hash, other.sequence_id, other.window_position)

    __hash__ = None

    def __reduce__(self):
        return (MinimizerInfo, (self.hash, self.sequence_id, self.window_position))


class Position:
    """A (sequence, window) position of a minimizer (_fastani.pyx:1382-1428)."""

    __slots__ = ("sequence_id", "window_position")

    def __init__(self, sequence_id, window_position):
        self.sequence_id = int(sequence_id)
        self.window_position = int(window_position)

    def __repr__(self):
        return "{}(sequence_id={!r}, window_position={!r})".format(
            type(self).__name__, self.sequence_id, self.window_position)

    def __eq__(self, other):
        if not isinstance(other, Position):
            raise TypeError("expected Position")
        return (self.sequence_id, self.window_position) == (other.sequence_id, other.window_position)

    __hash__ = None

    def __reduce__(self):
        return (Position, (self.sequence_id, self.window_position))


class Minimizers:
    """A read-only view over the minimizers of a `Sketch` or a `Mapper` (_fastani.pyx:1203-1268).

    The records live in HBM; they are read back once per owner state and cached.
    """

    def __init__(self, owner=None):
        self._owner = owner
        self._cache = None
        self._detached = None  # (hashes, ids, offsets) when unpickled on its own

    def _arrays(self):
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
        n = state["length"]
        self._owner = None
        self._cache = None
        self._detached = (
            np.asarray(state["hashes"][:n], dtype=np.uint32),
            np.asarray(state["ids"][:n], dtype=np.int32),
            np.asarray(state["offsets"][:n], dtype=np.int32),
        )


class MinimizerIndex:
    """The index mapping minimizer hash values to their positions (_fastani.pyx:1431-1539).

    A dict-like, read-only view over the device-resident lookup index of a `Mapper`.
    """

    def __init__(self, owner=None):
        self.owner = owner
        self._own = {}  # stand-alone instances (reference allows constructing an empty index)

    def __len__(self):
        if self.owner is None:
            return len(self._own)
        n = C.c_int64(0)
        check(lib.fa_mapper_lookup_size(self.owner._h, C.byref(n)))
        return n.value

    def _keys(self):
        n = len(self)
        keys = np.empty(n, np.uint32)
        if n:
            check(lib.fa_mapper_lookup_keys(self.owner._h, keys.ctypes.data))
        return keys

    def __iter__(self):
        if self.owner is None:
            return iter(list(self._own))
        return iter(self._keys().tolist())

    def __contains__(self, item):
        item = _as_uint(item, "item", 32)
        if self.owner is None:
            return item in self._own
        n = C.c_int64(0)
        check(lib.fa_mapper_lookup_count(self.owner._h, item, C.byref(n)))
        return n.value >= 0

    def __getitem__(self, item):
        item = _as_uint(item, "item", 32)
        if self.owner is None:
            return list(self._own[item])
        n = C.c_int64(0)
        check(lib.fa_mapper_lookup_count(self.owner._h, item, C.byref(n)))
        if n.value < 0:
            raise KeyError(item)
        seq = np.empty(n.value, np.int32)
        pos = np.empty(n.value, np.int32)
        check(lib.fa_mapper_lookup_get(self.owner._h, item, seq.ctypes.data, pos.ctypes.data, n.value))
        return [Position(int(a), int(b)) for a, b in zip(seq, pos)]

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


# --------------------------------------------------------------------------------------------------
# _Parameterized (_fastani.pyx:364-446)
# --------------------------------------------------------------------------------------------------
class _Parameterized:
    def _init_params(self):
        self._param = _lib.Params(16, 24, 3000, 4, 0.2, 80.0, 1e-3, 5_000_000)
        self._threads = 1

    def _params_getstate(self):
        p = self._param
        return {
            "kmerSize": p.kmer_size,
            "windowSize": p.window_size,
            "minReadLength": p.fragment_length,
            "minFraction": p.min_fraction,
            "threads": self._threads,
            "alphabetSize": p.alphabet_size,
            "referenceSize": p.reference_size,
            "percentageIdentity": p.percentage_identity,
            "p_value": p.p_value,
        }

    def _params_setstate(self, state):
        self._init_params()
        p = self._param
        p.kmer_size = state["kmerSize"]
        p.window_size = state["windowSize"]
        p.fragment_length = state["minReadLength"]
        p.min_fraction = state["minFraction"]
        self._threads = state["threads"]
        p.alphabet_size = state["alphabetSize"]
        p.reference_size = state["referenceSize"]
        p.percentage_identity = state["percentageIdentity"]
        p.p_value = state["p_value"]

    @property
    def k(self):
        """`int`: The k-mer size used for sketching."""
        return self._param.kmer_size

    @property
    def window_size(self):
        """`int`: The window size used for sketching."""
        return self._param.window_size

    @property
    def fragment_length(self):
        """`int`: The minimum read length to use for mapping."""
        return self._param.fragment_length

    @property
    def minimum_fraction(self):
        """`float`: The minimum genome fraction required to trust ANI values."""
        return self._param.min_fraction

    @property
    def percentage_identity(self):
        """`float`: The identity threshold for similarity when estimating hits."""
        return self._param.percentage_identity

    @property
    def p_value(self):
        """`float`: The p-value threshold for similarity when estimating hits."""
        return self._param.p_value

    @property
    def protein(self):
        """`bool`: Whether or not the object expects peptides or nucleotides."""
        return self._param.alphabet_size == 20


# --------------------------------------------------------------------------------------------------
# Sketch (_fastani.pyx:449-806)
# --------------------------------------------------------------------------------------------------
class Sketch(_Parameterized):
    """An index computing minimizers over the reference genomes.

    Use `add_genome` / `add_draft` to add reference genomes, then `index` to obtain a `Mapper`.
    Minimizers are extracted on the GPU, lazily: contigs are packed to 2 bits per base when added and
    sketched in one batch when the minimizers are first needed (``len(sketch.minimizers)``, `index`).
    """

    def __init__(self, *, k=16, fragment_length=3000, minimum_fraction=0.2, p_value=1e-03, percentage_identity=80.0,
                 reference_size=5_000_000, protein=False):
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

        self._init_params()
        p = self._param
        p.kmer_size = k
        p.fragment_length = fragment_length
        p.min_fraction = minimum_fraction
        p.p_value = p_value
        p.percentage_identity = percentage_identity
        p.reference_size = reference_size
        if protein:
            p.alphabet_size = 20
            p.window_size = 1
        else:
            p.alphabet_size = 4
            w = C.c_int(0)
            check(lib.fa_recommended_window_size(p.p_value, k, 4, p.percentage_identity, fragment_length, reference_size,
                                                 C.byref(w)))
            # the reference reads an uninitialised sketch size when no candidate meets the p-value cut-off
            # (e.g. fragment_length=100); clamp to the largest admissible window instead of emulating UB
            p.window_size = w.value if w.value > 0 else fragment_length
        self._lock = threading.Lock()
        self._release()
        self._h = None
        self._names = []
        self._version = 0
        self.minimizers = Minimizers(self)
        self._new_handle()

    # -- handle management ---------------------------------------------------------------------
    def _new_handle(self):
        h = C.c_void_p()
        check(lib.fa_sketch_new(C.byref(self._param), C.byref(h)))
        self._h = h

    def _release(self):
        h = getattr(self, "_h", None)
        if h:
            lib.fa_sketch_free(h)
        self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _state_token(self):
        return ("sketch", self._version)

    def _num_minimizers(self):
        n = C.c_int64(0)
        check(lib.fa_sketch_num_minimizers(self._h, C.byref(n)))
        return n.value

    def _read_minimizers(self):
        n = self._num_minimizers()
        h = np.empty(n, np.uint32)
        s = np.empty(n, np.int32)
        w = np.empty(n, np.int32)
        if n:
            check(lib.fa_sketch_get_minimizers(self._h, h.ctypes.data, s.ctypes.data, w.ctypes.data))
        return h, s, w

    def add_fasta(self, name, path):
        """Add every record of a FASTA file as the contigs of ONE reference genome (`add_draft` semantics,
        _fastani.pyx:692-717), read and packed natively without building Python objects (host ingest)."""
        import os
        n_rec, n_short = C.c_int64(0), C.c_int64(0)
        with self._lock:
            check(lib.fa_sketch_add_fasta(self._h, os.fsencode(path), C.byref(n_rec), C.byref(n_short)))
            self._names.append(name)
            self._version += 1
        for _ in range(n_short.value):
            warnings.warn(
                "Sketch received a short contig relative to parameters, minimizers will not be added.",
                UserWarning,
            )
        return self

    # -- record exchange for the multi-GPU index build (pyfastani_amd.sharding) ------
    def _export_records(self, device):
        """Minimizer records as one ``int32`` torch tensor ``[3, n]`` (hash bits, contig id, window position) on
        ``device`` plus the host-side state ``(lengths, sequencesByFileInfo, counter)``.  On a CUDA/HIP device the
        records are copied HBM to HBM."""
        import torch
        n = self._num_minimizers()
        ng = C.c_int64(0)
        check(lib.fa_sketch_num_genomes(self._h, C.byref(ng)))
        lengths = np.zeros(ng.value, np.uint64)
        sbf = np.zeros(ng.value, np.int32)
        counter = C.c_int64(0)
        check(lib.fa_sketch_get_state(self._h, lengths.ctypes.data, sbf.ctypes.data, C.byref(counter)))
        dev = torch.device(device)
        if dev.type == "cuda":
            rec = torch.empty((3, max(n, 1)), dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)
            base, row = rec.data_ptr(), rec.stride(0) * 4
            check(lib.fa_sketch_get_minimizers_device(self._h, rec.shape[1], base, base + row, base + 2 * row))
            rec = rec[:, :n]
        else:
            h, sq, w = self._read_minimizers()
            rec = torch.from_numpy(np.stack([h.view(np.int32), sq, w]))
        return rec, (lengths, sbf, counter.value)

    def _import_records(self, names, lengths, sbf, counter, rec):
        """Replace the content of this sketch by merged records (``rec`` as returned by `_export_records`)."""
        lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
        sbf = np.ascontiguousarray(sbf, dtype=np.int32)
        n = int(rec.shape[1])
        with self._lock:
            if rec.device.type == "cuda":
                import torch
                rec = rec.contiguous()
                torch.cuda.synchronize(rec.device)
                base, row = rec.data_ptr(), rec.stride(0) * 4
                check(lib.fa_sketch_set_state_device(self._h, len(lengths), lengths.ctypes.data, sbf.ctypes.data, int(counter),
                                                     n, base, base + row, base + 2 * row))
            else:
                a = np.ascontiguousarray(rec.numpy())
                h, sq, w = np.ascontiguousarray(a[0]).view(np.uint32), np.ascontiguousarray(a[1]), np.ascontiguousarray(a[2])
                check(lib.fa_sketch_set_state(self._h, len(lengths), lengths.ctypes.data, sbf.ctypes.data, int(counter), n,
                                              h.ctypes.data, sq.ctypes.data, w.ctypes.data))
            self._names = list(names)
            self._version += 1

    # -- pickling (_fastani.pyx:572-591) -----------------------------------------------------------
    def __getstate__(self):
        n = C.c_int64(0)
        check(lib.fa_sketch_num_genomes(self._h, C.byref(n)))
        lengths = np.zeros(n.value, np.uint64)
        sbf = np.zeros(n.value, np.int32)
        counter = C.c_int64(0)
        check(lib.fa_sketch_get_state(self._h, lengths.ctypes.data, sbf.ctypes.data, C.byref(counter)))
        return {
            "parameters": self._params_getstate(),
            "counter": counter.value,
            "lengths": lengths.tolist(),
            "names": list(self._names),
            "sketch": {"sequencesByFileInfo": sbf.tolist(), "minimizers": self.minimizers.__getstate__()},
        }

    def __setstate__(self, state):
        self._params_setstate(state["parameters"])
        self._lock = threading.Lock()
        self._h = None
        self._version = 0
        self._names = list(state["names"])
        self.minimizers = Minimizers(self)
        self._new_handle()
        mins = state["sketch"]["minimizers"]
        n = mins["length"]
        lengths = np.asarray(state["lengths"], dtype=np.uint64)
        sbf = np.asarray(state["sketch"]["sequencesByFileInfo"], dtype=np.int32)
        h = np.asarray(mins["hashes"][:n], dtype=np.uint32)
        s = np.asarray(mins["ids"][:n], dtype=np.int32)
        w = np.asarray(mins["offsets"][:n], dtype=np.int32)
        check(lib.fa_sketch_set_state(self._h, len(lengths), lengths.ctypes.data, sbf.ctypes.data, state["counter"], n,
                                      h.ctypes.data, s.ctypes.data, w.ctypes.data))

    # -- properties ------------------------------------------------------------------------------------
    @property
    def occurences_threshold(self):
        """`int`: The occurence threshold above which minimizers are ignored (INT_MAX until indexed)."""
        return 2**31 - 1

    @property
    def names(self):
        """`list`: The names of the sequences currently sketched."""
        return self._names[:]

    # -- methods ---------------------------------------------------------------------------------------
    def _add_draft(self, name, contigs):
        # _fastani.pyx:610-690
        for contig in contigs:
            addr, n, width, keep = _borrow(contig)
            added = C.c_int(0)
            check(lib.fa_sketch_add_contig(self._h, addr, n, width, C.byref(added)))
            del keep
            if not added.value:
                warnings.warn(
                    "Sketch received a short contig relative to parameters, minimizers will not be added.",
                    UserWarning,
                )
        self._names.append(name)
        check(lib.fa_sketch_end_genome(self._h))
        self._version += 1

    def add_draft(self, name, contigs):
        """Add a reference draft genome to the sketcher (_fastani.pyx:692-717)."""
        with self._lock:
            self._add_draft(name, contigs)
        return self

    def add_genome(self, name, sequence):
        """Add a reference genome to the sketcher (_fastani.pyx:719-744)."""
        with self._lock:
            self._add_draft(name, (sequence,))
        return self

    def clear(self):
        """Reset the `Sketch`, removing any reference genome it may contain (_fastani.pyx:746-767)."""
        self._names.clear()
        check(lib.fa_sketch_clear(self._h))
        self._version += 1
        return self

    def index(self):
        """Index the reference genomes for fast lookups using the minimizers (_fastani.pyx:769-806).

        Ownership of the data moves to the returned `Mapper`; this `Sketch` is cleared but stays usable.
        """
        h = C.c_void_p()
        check(lib.fa_sketch_index(self._h, C.byref(h)))
        mapper = Mapper.__new__(Mapper)
        mapper._adopt(h, self._param, self._threads, self._names.copy())
        self._names.clear()
        self._version += 1
        return mapper


# --------------------------------------------------------------------------------------------------
# Mapper (_fastani.pyx:809-1200)
# --------------------------------------------------------------------------------------------------
class Mapper(_Parameterized):
    """A genome mapper using Murmur3 hashes and k-mers to compute ANI, resident on one MI355X."""

    def __init__(self, *args, **kwargs):
        raise TypeError("Mapper cannot be instantiated, use `Sketch.index` instead.")  # :836-837

    def _adopt(self, handle, param, threads, names):
        self._init_params()
        C.memmove(C.byref(self._param), C.byref(param), C.sizeof(_lib.Params))
        self._threads = threads
        self._h = handle
        self._names = names
        self._lengths = None
        self.minimizers = Minimizers(self)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib.fa_mapper_free(h)
            except Exception:
                pass
            self._h = None

    def _state_token(self):
        return ("mapper", id(self))

    def _num_minimizers(self):
        n = C.c_int64(0)
        check(lib.fa_mapper_num_minimizers(self._h, C.byref(n)))
        return n.value

    def _read_minimizers(self):
        n = self._num_minimizers()
        h = np.empty(n, np.uint32)
        s = np.empty(n, np.int32)
        w = np.empty(n, np.int32)
        if n:
            check(lib.fa_mapper_get_minimizers(self._h, h.ctypes.data, s.ctypes.data, w.ctypes.data))
        return h, s, w

    def _state_arrays(self):
        n = C.c_int64(0)
        check(lib.fa_mapper_num_genomes(self._h, C.byref(n)))
        lengths = np.zeros(n.value, np.uint64)
        sbf = np.zeros(n.value, np.int32)
        check(lib.fa_mapper_get_state(self._h, lengths.ctypes.data, sbf.ctypes.data))
        return lengths, sbf

    @property
    def _genome_lengths(self):
        if self._lengths is None:
            self._lengths = self._state_arrays()[0]
        return self._lengths

    # -- pickling (_fastani.pyx:842-865): the index is rebuilt on load ----------------------------------
    def __getstate__(self):
        lengths, sbf = self._state_arrays()
        return {
            "parameters": self._params_getstate(),
            "lengths": lengths.tolist(),
            "names": list(self._names),
            "sketch": {"sequencesByFileInfo": sbf.tolist(), "minimizers": self.minimizers.__getstate__()},
        }

    def __setstate__(self, state):
        sk = Sketch.__new__(Sketch)
        sbf = state["sketch"]["sequencesByFileInfo"]
        sk.__setstate__({
            "parameters": state["parameters"],
            "counter": sbf[-1] if sbf else 0,
            "lengths": state["lengths"],
            "names": state["names"],
            "sketch": state["sketch"],
        })
        other = sk.index()
        self._adopt(other._h, other._param, other._threads, other._names)
        other._h = None

    # -- properties ---------------------------------------------------------------------------------------
    @property
    def lookup_index(self):
        """`MinimizerIndex`: the table of minimizer positions in the reference genomes (_fastani.pyx:869-881)."""
        return MinimizerIndex(self)

    def _export_lookup(self, device="cuda"):
        """Distinct hashes of this index (ascending, ``int32`` bit patterns) and the lengths of their position lists, as
        two torch tensors on ``device``: what a rank contributes to the global frequency threshold of a
        reference-sharded index (`sharding.global_frequency`)."""
        import torch
        n = C.c_int64(0)
        check(lib.fa_mapper_lookup_size(self._h, C.byref(n)))
        dev = torch.device(device)
        keys = torch.empty(max(n.value, 1), dtype=torch.int32, device="cuda")
        counts = torch.empty(max(n.value, 1), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        check(lib.fa_mapper_lookup_export_device(self._h, keys.shape[0], keys.data_ptr(), counts.data_ptr()))
        return keys[: n.value].to(dev), counts[: n.value].to(dev)

    def _set_global_frequency(self, threshold, drop_keys):
        """Install the frequency threshold taken over all shards of a reference-sharded index and the hashes (torch
        ``int32`` bit patterns) whose summed list length reaches it."""
        import torch
        drop = drop_keys.to(device="cuda", dtype=torch.int32).contiguous()
        torch.cuda.synchronize()
        check(lib.fa_mapper_set_global_frequency(self._h, int(threshold), int(drop.numel()), drop.data_ptr() if drop.numel() else None))

    @property
    def occurences_threshold(self):
        t = C.c_int(0)
        check(lib.fa_mapper_freq_threshold(self._h, C.byref(t)))
        return t.value

    @property
    def names(self):
        return self._names[:]

    # -- queries ------------------------------------------------------------------------------------------
    def _rows_to_hits(self, rows, total_length):
        # _fastani.pyx:1121-1136; the comparison is evaluated in float32 exactly like the C expression
        # `uint64 >= uint64 * float`
        lengths = self._genome_lengths
        frag = self._param.fragment_length
        min_fraction = np.float32(self._param.min_fraction)
        hits = []
        for r in rows:
            min_length = min(int(total_length), int(lengths[r.ref_genome_id]))
            shared_length = r.count_seq * frag
            if np.float32(shared_length) >= np.float32(min_length) * min_fraction:
                hits.append(Hit(self._names[r.ref_genome_id], r.identity, r.count_seq, r.total_query_fragments))
        hits.sort(key=lambda hit: hit.identity, reverse=True)  # stable, :1135
        return hits

    def _query_draft(self, contigs, threads=0):
        # _fastani.pyx:1006-1136.  `threads` is validated for signature compatibility; fragment-level
        # parallelism is the GPU's job.
        threads = operator.index(threads)
        if threads < 0:
            raise ValueError(f"`threads` must be positive or null, got {threads!r}")
        bufs, width = _borrow_all(contigs)
        n = len(bufs)
        ptrs = (C.c_void_p * max(n, 1))(*[b[0] for b in bufs])
        lens = (C.c_int64 * max(n, 1))(*[b[1] for b in bufs])
        n_genomes = len(self._names)
        rows = (_lib.CgiRow * max(n_genomes, 1))()
        n_rows, n_short = C.c_int64(0), C.c_int(0)
        total_fragments, total_length = C.c_uint64(0), C.c_uint64(0)
        check(lib.fa_mapper_query(self._h, ptrs, lens, n, width, rows, max(n_genomes, 1), C.byref(n_rows),
                                  C.byref(n_short), C.byref(total_fragments), C.byref(total_length)))
        for _ in range(n_short.value):
            warnings.warn(
                "Mapper received a short sequence relative to parameters, mapping will not be computed.",
                UserWarning,
            )
        return self._rows_to_hits(rows[: n_rows.value], total_length.value)

    def query_draft(self, contigs, threads=0):
        """Query the mapper for a complete genome given as contigs (_fastani.pyx:1138-1168)."""
        return self._query_draft(contigs, threads=threads)

    def query_genome(self, sequence, threads=0):
        """Query the mapper for a complete, closed genome (_fastani.pyx:1170-1200)."""
        return self._query_draft((sequence,), threads=threads)

    # -- many-to-many extension (no reference analogue): resident batches ------------------------------
    def upload_genomes(self, genomes):
        """Pack a list of draft genomes (each an iterable of contigs) into HBM and return a `GenomeBatch`."""
        from ._batch import GenomeBatch
        return GenomeBatch(self, genomes)

    def query_fasta(self, path):
        """`query_draft` for a genome stored as a FASTA file (its records are the contigs): read, packed and mapped
        without creating a Python object per contig.  Returns the same `Hit` list as `query_draft`."""
        return self.upload_fasta([path]).query()[0]

    def upload_fasta(self, paths):
        """One query genome per FASTA file (its records are the contigs), read, packed and uploaded natively."""
        from ._batch import GenomeBatch
        return GenomeBatch.from_fasta(self, paths)

    def query_batch(self, batch, first=0, count=None):
        """Map genomes ``[first, first+count)`` of a resident batch; returns one hit list per genome."""
        return batch.query(first, count)

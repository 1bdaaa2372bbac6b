"""ctypes binding of ``libfastani_hip.so`` (the C ABI declared in ``include/fastani_hip.h``).

The product binding is the Cython module ``pyfastani_amd._fastani``; this table of every exported symbol serves the
tests (symbol / layout checks), the debug and timing entry points used by ``bench.py`` and ``scripts/``, and as the
ctypes stub shown in INTEGRATION.md.  The library is the only compute backend: if it is missing, importing this module
raises ``ImportError`` with build instructions -- there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfastani_hip.so")

FA_OK, FA_ERR_INVALID, FA_ERR_NO_DEVICE, FA_ERR_NOMEM, FA_ERR_UNSUPPORTED, FA_ERR_INTERNAL, FA_ERR_IO, FA_ERR_BUFFER = range(8)


class Params(C.Structure):
    # skch::Parameters, include/fastani/map/map_parameters.pxd:9-24 of the reference
    _fields_ = [
        ("kmer_size", C.c_int32),
        ("window_size", C.c_int32),
        ("fragment_length", C.c_int32),
        ("alphabet_size", C.c_int32),
        ("min_fraction", C.c_float),
        ("percentage_identity", C.c_float),
        ("p_value", C.c_double),
        ("reference_size", C.c_uint64),
    ]


class CgiRow(C.Structure):
    # cgi::CGI_Results, include/fastani/cgi/cgid_types.pxd:19-27
    _fields_ = [
        ("query_id", C.c_int32),
        ("ref_genome_id", C.c_int32),
        ("count_seq", C.c_int32),
        ("total_query_fragments", C.c_int32),
        ("identity", C.c_float),
    ]


class Mapping(C.Structure):
    _fields_ = [
        ("query_seq_id", C.c_int32),
        ("ref_seq_id", C.c_int32),
        ("ref_start_pos", C.c_int32),
        ("sketch_size", C.c_int32),
        ("conserved", C.c_int32),
        ("query_id", C.c_int32),
    ]


# every symbol of include/fastani_hip.h: (restype, argtypes)
_vp, _i32, _i64, _u64, _u32, _f32, _f64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint32, C.c_float, C.c_double
_P = C.POINTER
SIGNATURES = {
    "fa_last_error": (C.c_char_p, []),
    "fa_version": (_i32, []),
    "fa_device_trim": (_i32, [_P(C.c_uint64)]),
    "fa_device_count": (_i32, [_P(_i32)]),
    "fa_set_device": (_i32, [_i32]),
    "fa_recommended_window_size": (_i32, [_f64, _i32, _i32, _f32, _i32, _u64, _P(_i32)]),
    "fa_estimate_minimum_hits_relaxed": (_i32, [_i32, _i32, _f32, _P(_i32)]),
    "fa_mapping_identity": (_i32, [_i32, _i32, _i32, _P(_f32), _P(_f32)]),
    "fa_hash": (_u32, [C.c_char_p, _i32]),
    "fa_sketch_new": (_i32, [_P(Params), _P(_vp)]),
    "fa_sketch_free": (None, [_vp]),
    "fa_sketch_add_contig": (_i32, [_vp, _vp, _i64, _i32, _P(_i32)]),
    "fa_sketch_end_genome": (_i32, [_vp]),
    "fa_sketch_abort_genome": (_i32, [_vp]),
    "fa_sketch_clear": (_i32, [_vp]),
    "fa_sketch_num_minimizers": (_i32, [_vp, _P(_i64)]),
    "fa_sketch_get_minimizers": (_i32, [_vp, _vp, _vp, _vp]),
    "fa_sketch_num_genomes": (_i32, [_vp, _P(_i64)]),
    "fa_sketch_get_state": (_i32, [_vp, _vp, _vp, _P(_i64)]),
    "fa_sketch_set_state": (_i32, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "fa_sketch_get_minimizers_device": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "fa_sketch_set_state_device": (_i32, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "fa_sketch_index": (_i32, [_vp, _P(_vp)]),
    "fa_mapper_free": (None, [_vp]),
    "fa_mapper_freq_threshold": (_i32, [_vp, _P(_i32)]),
    "fa_mapper_lookup_export_device": (_i32, [_vp, _i64, _vp, _vp]),
    "fa_mapper_set_global_frequency": (_i32, [_vp, _i32, _i64, _vp]),
    "fa_mapper_lookup_size": (_i32, [_vp, _P(_i64)]),
    "fa_mapper_device": (_i32, [_vp, _P(_i32)]),
    "fa_mapper_lookup_keys": (_i32, [_vp, _vp]),
    "fa_mapper_lookup_count": (_i32, [_vp, _u32, _P(_i64)]),
    "fa_mapper_lookup_get": (_i32, [_vp, _u32, _vp, _vp, _i64]),
    "fa_mapper_num_minimizers": (_i32, [_vp, _P(_i64)]),
    "fa_mapper_get_minimizers": (_i32, [_vp, _vp, _vp, _vp]),
    "fa_mapper_num_genomes": (_i32, [_vp, _P(_i64)]),
    "fa_mapper_get_state": (_i32, [_vp, _vp, _vp]),
    "fa_mapper_query": (_i32, [_vp, _P(_vp), _P(_i64), _i32, _i32, _vp, _i64, _P(_i64), _P(_i32), _P(_u64), _P(_u64)]),
    "fa_genomes_upload": (_i32, [_vp, _P(_vp), _P(_i64), _vp, _i64, _i32, _i32, _P(_vp)]),
    "fa_genomes_free": (None, [_vp]),
    "fa_genomes_info": (_i32, [_vp, _P(_i32), _vp, _vp, _vp]),
    "fa_mapper_query_genomes": (_i32, [_vp, _vp, _i32, _i32, _vp, _i64, _P(_i64), _i32]),
    "fa_mapper_debug_mappings": (_i32, [_vp, _vp, _i64, _P(_i64)]),
    "fa_mapper_debug_l1": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _P(_i64)]),
    "fa_mapper_debug_query_sketch": (_i32, [_vp, _i64, _vp, _i32, _P(_i32)]),
    "fa_debug_sketch_sequence": (_i32, [_P(Params), _vp, _i64, _i32, _vp, _vp, _i64, _P(_i64)]),
    "fa_fasta_open": (_i32, [C.c_char_p, _P(_vp)]),
    "fa_fasta_next": (_i32, [_vp, _P(_i32), _P(_vp), _P(_i64), _P(_vp), _P(_i64)]),
    "fa_fasta_close": (None, [_vp]),
    "fa_sketch_add_fasta": (_i32, [_vp, C.c_char_p, _P(_i64), _P(_i64)]),
    "fa_genomes_upload_fasta": (_i32, [_vp, _P(C.c_char_p), _i32, _P(_vp)]),
    "fa_packed_read": (_i32, [_P(C.c_char_p), _i32, _i32, _P(_vp)]),
    "fa_packed_append": (_i32, [_vp, _P(C.c_char_p), _i32]),
    "fa_packed_free": (None, [_vp]),
    "fa_packed_info": (_i32, [_vp, _P(_i32), _P(C.c_uint64), _P(_i64), _P(_i64)]),
    "fa_sketch_add_packed": (_i32, [_vp, _vp, _i32, _i32, _P(_i64), _P(_i64)]),
    "fa_genomes_reload_packed": (_i32, [_vp, _vp, _vp, _i32, _i32]),
    "fa_genomes_reload_fasta": (_i32, [_vp, _vp, _P(C.c_char_p), _i32]),
    "fa_sketch_add_genomes": (_i32, [_vp, _P(_vp), _P(_i64), _P(_i32), _i64, _i32, _i32, _P(_i32)]),
    "fa_sketch_add_fasta_many": (_i32, [_vp, _P(C.c_char_p), _i32, _P(_i64), _P(_i64)]),
    "fa_debug_probe_occupancy": (_i32, [_i32, _P(_i32)]),
    "fa_mapper_debug_items": (_i32, [_vp, _vp, _i64]),
    "fa_mapper_debug_locus_events": (_i32, [_vp, _vp, _i64, _vp]),
    "fa_mapper_debug_links": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "fa_mapper_last_timings": (_i32, [_vp, _P(_f32), _i32]),
    "fa_mapper_set_stage_events": (_i32, [_vp, _i32]),
    "fa_mapper_stream": (_i32, [_vp, _P(_vp)]),
    "fa_bench_sketch_kernel": (_i32, [_vp, _vp, _i32, _P(_f32), _P(_u64), _P(_u64)]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950). pyfastani_amd has no CPU fallback."
    )



# Concurrent query calls each use their own stream; the runtime's default of 4 hardware queues would put several of them
# on one queue (measured: two part streams serialised), so ask for more before HIP starts.  Nothing else is touched at
# import: in particular PyTorch is NOT imported here (a process that uses both imports torch first, see sharding.py).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here means the library and the header disagree
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    msg = lib.fa_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(code):
    """Map a C status code onto the exception the reference would raise."""
    if code == FA_OK:
        return
    msg = last_error()
    if code == FA_ERR_INVALID:
        raise ValueError(msg)
    if code == FA_ERR_NOMEM:
        raise MemoryError(msg)
    if code == FA_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if code == FA_ERR_IO:
        raise OSError(msg)
    if code == FA_ERR_BUFFER:
        raise BufferError(msg)
    raise RuntimeError(msg)


def device_count():
    n = C.c_int(0)
    lib.fa_device_count(C.byref(n))
    return n.value

"""The C-ABI library loads and exports exactly what include/fastani_hip.h declares; host-only entry points agree
with the oracle (no GPU compute here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu
from oracle.oracle import lib as olib, murmur_hash
from pyfastani_amd import _lib
from pyfastani_amd._lib import lib, check


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fastani_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fa_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    names = declared_symbols()
    assert len(names) >= 35
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/fastani_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_struct_layouts():
    assert C.sizeof(_lib.Params) == 40
    assert C.sizeof(_lib.CgiRow) == 20
    assert C.sizeof(_lib.Mapping) == 24


def test_host_hash_matches_oracle():
    rng = np.random.default_rng(0)
    for k in list(range(1, 40)) + [64, 200]:
        kmer = bytes(rng.integers(65, 91, k, dtype=np.uint8))
        assert lib.fa_hash(kmer, k) == murmur_hash(kmer)


def test_window_and_min_hits_match_oracle():
    w = C.c_int(0)
    for k, frag in [(16, 3000), (14, 1000), (16, 5000), (21, 3000), (21, 1000), (5, 3000)]:
        check(lib.fa_recommended_window_size(1e-3, k, 4, 80.0, frag, 5_000_000, C.byref(w)))
        assert w.value == olib().fo_recommended_window(1e-3, k, 4, 80.0, frag, 5_000_000)
    h = C.c_int(0)
    for s in list(range(1, 400, 7)) + [1000, 2962]:
        for pid in (80.0, 90.0, 95.0, 70.0):
            check(lib.fa_estimate_minimum_hits_relaxed(s, 16, pid, C.byref(h)))
            assert h.value == olib().fo_min_hits_relaxed(s, 16, pid)


def test_identity_table_matches_oracle():
    a, b = C.c_float(0), C.c_float(0)
    oa, ob = C.c_float(0), C.c_float(0)
    for k in (14, 16, 21):
        for s in (1, 2, 17, 85, 150, 233, 240, 256, 300):
            prev_upper = -1.0
            for c in range(0, s + 1):
                check(lib.fa_mapping_identity(c, s, k, C.byref(a), C.byref(b)))
                olib().fo_identity(c, s, k, C.byref(oa), C.byref(ob))
                assert a.value == oa.value and b.value == ob.value  # bit-exact float32
                assert b.value >= prev_upper  # the pass filter is monotone in the shared count
                prev_upper = b.value


def test_invalid_params_are_reported_not_thrown():
    p = _lib.Params(0, 24, 3000, 4, 0.2, 80.0, 1e-3, 5_000_000)
    h = C.c_void_p()
    assert lib.fa_sketch_new(C.byref(p), C.byref(h)) == _lib.FA_ERR_INVALID
    assert b"kmer_size" in lib.fa_last_error()


@pytest.mark.skipif(has_gpu(), reason="CPU-only behaviour")
def test_compute_fails_loudly_without_device():
    import pyfastani_amd as pf
    sk = pf.Sketch()
    sk.add_genome("x", "ACGT" * 1000)  # packing is host work
    with pytest.raises(RuntimeError, match="no HIP device"):
        len(sk.minimizers)
    with pytest.raises(RuntimeError, match="no HIP device"):
        sk.index()


def test_no_kernel_of_the_library_needs_scratch_memory():
    """The HIP runtime keeps a scratch allocation per stream for good once any kernel has asked for one (a mapper life cycle
    then leaks a few MB of HBM: scripts/check_leaks.py); register spills also cost the hot kernels their occupancy targets.
    The build keeps the compiler's resource remarks (`__graft_entry__._write_kernel_resources`): every `fa::` kernel must
    report zero scratch (rocPRIM's radix sort at index time is the one exception, on the sketch's own stream)."""
    import json
    path = os.path.join(ROOT, "pyfastani_amd", "lib", "libfastani_hip.so.kernels.json")
    if not os.path.exists(path):
        import __graft_entry__ as entry
        entry.build(force=True)
    kernels = json.load(open(path))
    ours = {k: v for k, v in kernels.items() if k.startswith("_ZN2fa")}
    assert len(ours) > 40, len(ours)
    spilled = {k: v["scratch_bytes_per_lane"] for k, v in ours.items() if v["scratch_bytes_per_lane"]}
    assert not spilled, spilled
    hot = [v for k, v in ours.items() if "k_l1ILi" in k and "ELi16E" in k]
    assert hot and all(v["occupancy_waves_per_simd"] == 8 for v in hot), hot


def test_the_slide_microbenchmark_still_compiles_against_the_kernels():
    """`scripts/ubench/slide_chain.hip` launches the PRODUCT `k_l2_scan` on synthetic event streams (the numbers of
    profiles/r03_valu_rates.txt and of every slide experiment come from it): it fills `L2Args` by hand, so a change of the
    kernel's arguments must reach it (syntax check only: seconds, no GPU)."""
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "scripts", "ubench", "slide_chain.hip")
    proc = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-w", src],
                          cwd=os.path.dirname(src), stderr=subprocess.PIPE, text=True)
    assert proc.returncode == 0, proc.stderr[-2000:]
    text = open(src).read()
    assert "a.loci.count" in text and "a.loci.n" in text and "a.loci.shift" in text    # (the arguments k_l2_scan finds its loci by)

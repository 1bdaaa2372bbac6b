"""The host layer is a compiled Cython module over the C ABI, importable without PyTorch."""
import os
import subprocess
import sys

import pytest

import pyfastani_amd as pf
from conftest import ROOT


def test_classes_come_from_the_compiled_binding():
    mod = sys.modules["pyfastani_amd._fastani"]
    assert mod.__file__.endswith(".so")                      # not a Python module: the extension built by build()
    for cls in (pf.Sketch, pf.Mapper, pf.Hit, pf.GenomeBatch, pf.MinimizerInfo, pf.Position, pf.MinimizerIndex):
        assert cls.__module__ == "pyfastani_amd._fastani"
    # the extension links the library the header describes (no second copy of the engine inside it)
    out = subprocess.run(["ldd", mod.__file__], capture_output=True, text=True).stdout
    assert "libfastani_hip.so" in out and "not found" not in out.split("libfastani_hip.so")[1].split("\n")[0]


def test_import_needs_neither_torch_nor_numpy_and_is_fast():
    code = ("import time, sys; t = time.perf_counter(); import pyfastani_amd as pf; dt = time.perf_counter() - t; "
            "sk = pf.Sketch(); assert sk.window_size == 24; "
            "assert 'torch' not in sys.modules and 'numpy' not in sys.modules, sorted(m for m in sys.modules if m in ('torch', 'numpy')); "
            "print(dt)")
    env = dict(os.environ, PYTHONPATH=ROOT)
    best = min(float(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout)
               for _ in range(3))
    assert best < 0.3, best                                    # "import time without torch below 0.3 s"


def test_reference_signatures():
    sk = pf.Sketch(k=16, fragment_length=3000, minimum_fraction=0.2, p_value=1e-3, percentage_identity=80.0,
                   reference_size=5_000_000, protein=False)          # _fastani.pyx:484-494, keyword-only
    assert sk.add_genome("g", "ACGT" * 1000) is sk and sk.add_draft("d", ["ACGT" * 1000]) is sk and sk.clear() is sk
    with pytest.raises(TypeError):
        pf.Hit("a", 1.0, 1, 1) == 3                                 # typed __eq__, _fastani.pyx:1300

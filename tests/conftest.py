import os
import sys

import pytest

# A process that uses both PyTorch (the multi-GPU tests) and libfastani_hip.so must load torch first, so that both sit on
# the HIP runtime torch bundles (pyfastani_amd/sharding.py); the package itself never imports torch.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # make sure the native pieces exist before any test imports them
    import __graft_entry__ as entry
    entry.build()


def has_gpu():
    from pyfastani_amd import _lib
    return _lib.device_count() > 0


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def read_fasta(path):
    records, cur = [], None
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                cur = []
                records.append(cur)
            elif line and cur is not None:
                cur.append(line)
    return ["".join(r) for r in records]

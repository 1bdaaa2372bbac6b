"""Differential fuzzing of the HIP path against the oracle (scripts/fuzz_parity.py): random genomes with tandem
repeats, dispersed repeats, inversions, low-complexity runs, N runs, IUPAC codes, lower case, drafts and short contigs,
under random (k, fragment_length, percentage_identity, minimum_fraction).  Every L2 mapping, the index size, the
frequency threshold and every hit must match.  A 10 000-case campaign (seeds 2-5) ran clean in round 1."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_fuzz_against_oracle():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "400", "11"],
                         capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "0 mismatches" in res.stdout

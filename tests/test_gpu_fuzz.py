"""Differential fuzzing of the HIP path against the oracle (scripts/fuzz_parity.py): random genomes with tandem
repeats, dispersed repeats, inversions, low-complexity runs, N runs, IUPAC codes, lower case, drafts and short contigs,
under random (k, fragment_length, percentage_identity, minimum_fraction).  Every L2 mapping, the index size, the
frequency threshold and every hit must match.  A 10 000-case campaign (seeds 2-5) ran clean in round 1.

The seed is NOT fixed: it is derived from the kernel and oracle sources, so every change of either draws a fresh set of
cases (the same sources always replay the same set; the seed is printed on failure and `scripts/fuzz_parity.py <cases>
<seed>` replays it).  The run is a fixed number of cases -- about 100 s on an idle box -- so that a loaded machine makes the
test slower, not red."""
import hashlib
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def source_seed():
    h = hashlib.sha256()
    for d in ("pyfastani_amd/csrc", "oracle"):
        for name in sorted(os.listdir(os.path.join(ROOT, d))):
            if name.endswith((".h", ".hip", ".hpp", ".cpp")):
                h.update(open(os.path.join(ROOT, d, name), "rb").read())
    return int.from_bytes(h.digest()[:4], "little") & 0x7FFFFFFF


def test_fuzz_against_oracle():
    seed = source_seed()
    cases = 400
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), str(cases), str(seed)],
                         capture_output=True, text=True, timeout=2400)
    assert res.returncode == 0, f"seed {seed}\n" + res.stdout[-3000:] + res.stderr[-3000:]
    assert "0 mismatches" in res.stdout and f"seed {seed}" in res.stdout
    assert int(res.stdout.strip().splitlines()[-1].split()[0]) == cases, res.stdout[-500:]


def test_fuzz_with_the_coordinate_filter_of_large_indices_forced_on():
    """k_l1 skips the coordinate fetch of hits that cannot be an end of a candidate only on indices of 3 x 10^8 records and
    more (where it pays); FA_L1_NEAR=1 forces it onto the small indices of the fuzzer, so that its loci are compared with the
    oracle's mapping for mapping -- chance hits, repeats, tandem duplications, minimum hit counts from 1 up."""
    seed = (source_seed() ^ 0x5A5A5A) & 0x7FFFFFFF
    cases = 250
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), str(cases), str(seed)],
                         env=dict(os.environ, FA_L1_NEAR="1"), capture_output=True, text=True, timeout=2400)
    assert res.returncode == 0, f"seed {seed}\n" + res.stdout[-3000:] + res.stderr[-3000:]
    assert "0 mismatches" in res.stdout and int(res.stdout.strip().splitlines()[-1].split()[0]) == cases, res.stdout[-500:]

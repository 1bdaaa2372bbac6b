"""FastANI-style outputs from a hit table (CPU only, fabricated rows)."""
import numpy as np

from pyfastani_amd import outputs
from pyfastani_amd._batch import ROW_DTYPE


def rows_of(*t):
    return np.array(list(t), dtype=ROW_DTYPE)


def test_filter_matrix_and_files(tmp_path):
    rows = rows_of((0, 0, 100, 100, 100.0), (0, 1, 90, 100, 95.5), (1, 0, 88, 100, 95.0), (1, 1, 100, 100, 100.0),
                   (2, 2, 100, 100, 100.0), (2, 0, 5, 100, 79.0))
    lengths = [300_000, 300_000, 300_000]
    kept = outputs.filter_rows(rows, lengths, lengths, 3000, 0.2)
    assert len(kept) == 5 and not any((kept["query_id"] == 2) & (kept["ref_genome_id"] == 0))   # 5 x 3000 < 20 % of 300 kb
    m = outputs.identity_matrix(kept, 3, 3, symmetric=True)
    assert m[0, 1] == m[1, 0] == (95.5 + 95.0) / 2 and np.isnan(m[2, 0]) and m[2, 2] == 100.0
    p = tmp_path / "out.matrix"
    outputs.write_matrix(str(p), ["a", "b", "c"], m)
    assert p.read_text().splitlines() == ["3", "a", "b\t95.250000", "c\tNA\tNA"]
    h = tmp_path / "out.tsv"
    outputs.write_hits(str(h), ["a", "b", "c"], ["a", "b", "c"], kept)
    lines = h.read_text().splitlines()
    assert lines[0].split("\t") == ["a", "a", "100", "100", "100"] and lines[1].split("\t")[:3] == ["a", "b", "95.5"]
    assert len(lines) == 5

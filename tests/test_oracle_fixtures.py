"""The oracle still reproduces the committed synthetic fixtures (guards the checker itself against drift)."""
import json
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_synthetic_goldens import build_case  # noqa: E402
from oracle.oracle import OracleSketch  # noqa: E402


def test_oracle_reproduces_fixtures(golden_dir):
    fixtures = json.load(open(os.path.join(golden_dir, "synthetic_goldens.json")))
    for fx in fixtures[:3]:
        case = fx["case"]
        refs, queries = build_case(case)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sk = OracleSketch(**case["params"])
        for i, r in enumerate(refs):
            sk.add_draft(f"ref{i}", r)
        assert sk.window_size == fx["window"] and len(sk.minimizers()[0]) == fx["n_minimizers"]
        sk.index()
        assert sk.index_size == fx["index_size"] and sk.freq_threshold == fx["freq_threshold"]
        for q, want in zip(queries, fx["queries"]):
            hits = sk.query_draft(q)
            assert [[h[0], float(np.float32(h[1])), h[2], h[3]] for h in hits] == want["hits"]

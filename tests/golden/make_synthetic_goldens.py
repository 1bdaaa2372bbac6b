"""Generates tests/golden/synthetic_goldens.json with the CPU oracle (the real reference cannot be built or
imported: its C++ is an empty submodule).  Inputs are regenerated from seeds at test time; only
parameters and expected outputs are stored."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from oracle.oracle import OracleSketch
from pyfastani_amd import synthetic as syn


def build_case(case):
    g = syn.rng(case["seed"])
    anc = syn.random_codes(g, case["length"])
    refs = []
    for i, d in enumerate(case["ref_div"]):
        seq = syn.to_ascii(syn.mutate_codes(g, anc, d)) if d is not None else syn.to_ascii(syn.random_codes(g, case["length"]))
        refs.append(syn.split_contigs(g, seq, case["contigs"]) if case["contigs"] > 1 else [seq])
    queries = []
    for d in case["query_div"]:
        seq = syn.to_ascii(syn.mutate_codes(g, anc, d))
        if case.get("rc_query"):
            seq = syn.to_ascii(syn.reverse_complement_codes(np.searchsorted(syn.ACGT, seq).astype(np.uint8)))
        queries.append(syn.split_contigs(g, seq, case["contigs"]) if case["contigs"] > 1 else [seq])
    return refs, queries


CASES = [
    {"name": "default_300k", "seed": 101, "length": 300_000, "ref_div": [0.01, 0.05, 0.10, 0.15, 0.20, None], "query_div": [0.03, 0.12],
     "contigs": 1, "params": {}},
    {"name": "drafts_400k", "seed": 102, "length": 400_000, "ref_div": [0.02, 0.08, None, 0.15], "query_div": [0.05], "contigs": 12,
     "params": {}},
    {"name": "k14_frag1000", "seed": 103, "length": 200_000, "ref_div": [0.03, 0.10, None], "query_div": [0.06], "contigs": 1,
     "params": {"k": 14, "fragment_length": 1000}},
    {"name": "k21_frag5000", "seed": 104, "length": 300_000, "ref_div": [0.03, 0.10, None], "query_div": [0.06], "contigs": 1,
     "params": {"k": 21, "fragment_length": 5000}},
    {"name": "pid90_rc", "seed": 105, "length": 250_000, "ref_div": [0.01, 0.04, 0.12], "query_div": [0.02], "contigs": 3,
     "params": {"percentage_identity": 90.0, "minimum_fraction": 0.5}, "rc_query": True},
]

if __name__ == "__main__":
    import warnings
    out = []
    for case in CASES:
        refs, queries = build_case(case)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sk = OracleSketch(**case["params"])
        for i, r in enumerate(refs):
            sk.add_draft(f"ref{i}", r)
        n_min = len(sk.minimizers()[0])
        sk.index()
        expected = []
        for q in queries:
            hits, det = sk.query_draft(q, details=True)
            expected.append({"hits": [[h[0], float(np.float32(h[1])), h[2], h[3]] for h in hits], "n_mappings": int(len(det["mappings"]["qseq"]))})
        out.append({"case": case, "window": sk.window_size, "n_minimizers": n_min, "index_size": sk.index_size,
                    "freq_threshold": sk.freq_threshold, "queries": expected})
        print(case["name"], sk.window_size, n_min, sk.index_size, expected)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "synthetic_goldens.json"), "w") as f:
        json.dump(out, f, indent=1)

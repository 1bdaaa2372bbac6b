#!/usr/bin/env python3
"""What each OPEN reading of the absent upstream C++ would move (CPU only, oracle only -- test infrastructure).

The oracle (oracle/fastani_oracle.hpp) keeps every rule that no in-tree golden with inputs present can tell from its
alternatives behind a named compile-time switch FO_<RULE>; 0 is the reading SURVEY.md 8a wrote, which the HIP path follows.
This script builds each alternative into a library of its own (FA_ORACLE_DEFINES, oracle/oracle.py), and for each one

  * runs the three in-tree pins: the protein golden (test_ani.py:96-115: 130/176 x2, order), window_size == 24
    (test_ani.py:60,80) and the self-query invariant (test_ani.py:66-71: exactly 100.0, every fragment matched; also for the
    reverse-complemented genome and for a genome with repeats) -- which alternatives do the pins already exclude?
  * sizes what it moves against the default reading on BASELINE config 2 at full size (1 query x 100 references of 5 Mb), on a
    genome-like all-vs-all (12 genomes x 1 Mb with repeats, indels, an inversion) and on the committed synthetic goldens
    (tests/golden/synthetic_goldens.json): L2 mappings, CGI rows and final hits that change, max |dANI|, max |dmatches|.

    python scripts/oracle_sensitivity.py                 # all variants -> profiles/r06_open_rule_sensitivity.json
    python scripts/oracle_sensitivity.py --quick         # config 2 at 10 x 1 Mb (minutes, for a look)
"""
import argparse
import json
import os
import pickle
import subprocess
import sys
import tempfile
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

VARIANTS = [
    ("default", "", "the reading SURVEY.md 8a wrote; the HIP path follows it"),
    ("L2_CI=0.75", "FO_L2_CI=0.75f", "confidence interval 0.75 at the doL2Mapping site (0.9 at the S6b site stays pinned by window_size == 24)"),
    ("SLIDE_END=rangeEndPos+Q.len", "FO_SLIDE_END=1", "the slide ends at searchIndex(rangeEndPos + Q.len) instead of rangeEndPos + cmw: (w-1)+(k-1) more window positions"),
    ("SLIDE_ADVANCE=one record per step", "FO_SLIDE_ADVANCE=1", "window = records with wpos in [front.wpos, front.wpos + cmw), the front record leaves at every step"),
    ("SLIDE_EVAL=after drop and after admit", "FO_SLIDE_EVAL=1", "a step that drops and admits reads the counter twice"),
    ("BEST_INIT=only > resets", "FO_BEST_INIT=1", "sharedSketchSize starts at 0 and the first placement is not special"),
    ("CGI_BIN=fragLen", "FO_CGI_BIN=1", "reference bin refStartPos / fragLen instead of / (fragLen - 20)"),
    ("MD2J_EXP=double", "FO_MD2J_EXP=1", "exp in double inside md2j"),
    ("CGI_TIES=largest", "FO_CGI_TIES=1", "equal-identity ties in computeCGI go to the largest (refSeqId, refStartPos) / querySeqId"),
]


def _detail(det):
    m, r = det["mappings"], det["rows"]
    maps = sorted(zip(m["qseq"].tolist(), m["rseq"].tolist(), m["rstart"].tolist(), m["shared"].tolist(), m["sketch"].tolist()))
    rows = {int(g): (int(c), float(i)) for g, c, i in zip(r["genome"].tolist(), r["count"].tolist(), r["identity"].tolist())}
    return maps, rows


def worker(args):
    """One variant (the library is chosen by FA_ORACLE_DEFINES in this process's environment): pins + the three data sets."""
    import numpy as np
    from oracle.oracle import OracleSketch
    from pyfastani_amd import synthetic as syn, workloads
    from make_synthetic_goldens import build_case
    cores = os.cpu_count() or 1
    out = {"defines": os.environ.get("FA_ORACLE_DEFINES", "")}
    t_all = time.time()
    # ---- pin 1: window_size == 24 (default parameters) and the restated window table ----
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out["window_default"] = OracleSketch().window_size
        out["window_table"] = {f"{k},{f}": OracleSketch(k=k, fragment_length=f).window_size for k, f in workloads.CONFIG5_CELLS}
    # ---- pin 2: the protein golden ----
    def faa(name):
        seqs, cur = [], []
        for line in open(os.path.join(ROOT, "tests", "golden", name + ".faa")):
            if line.startswith(">"):
                if cur:
                    seqs.append("".join(cur))
                cur = []
            else:
                cur.append(line.strip())
        if cur:
            seqs.append("".join(cur))
        return seqs
    sk = OracleSketch(protein=True, fragment_length=100)
    sk.add_draft("BGC0001425", faa("BGC0001425"))
    sk.add_draft("BGC0001427", faa("BGC0001425"))             # (test_ani.py:103 feeds bgc1 twice)
    sk.index()
    out["protein_hits"] = [[h[0], h[2], h[3]] for h in sk.query_draft(faa("BGC0001428"))]
    # ---- pin 3: the self-query invariant (random genome, its reverse complement, a genome with repeats) ----
    g = syn.rng(31337)
    codes = syn.random_codes(g, 300_000)
    like = workloads.genome_like(31338, 1, 1, 300_000)[0][0][0]
    selfq = []
    for name, ref, query in (("random", syn.to_ascii(codes), syn.to_ascii(codes)),
                             ("reverse complement", syn.to_ascii(codes), syn.to_ascii(syn.reverse_complement_codes(codes))),
                             ("with repeats", like, like)):
        sk = OracleSketch()
        sk.add_genome("self", ref)
        sk.index()
        h = sk.query_draft([query])
        selfq.append({"case": name, "identity": h[0][1] if h else None, "matches": h[0][2] if h else 0, "fragments": h[0][3] if h else 0})
    out["self_query"] = selfq
    # ---- data set 1: BASELINE config 2 ----
    n_refs, length = (10, 1_000_000) if args.quick else (100, 5_000_000)
    t0 = time.time()
    anc, names, refs = workloads.config2_references(n_refs, length)
    query = workloads.config2_query(anc, 0, 1)[0]
    sk = OracleSketch()
    sk.add_drafts(names, refs, threads=cores)
    sk.index()
    hits, det = sk.query_draft(query, threads=cores, details=True)
    maps, rows = _detail(det)
    out["config2"] = {"workload": f"1 query x {n_refs} refs of {length / 1e6:g} Mb", "hits": hits, "mappings": maps, "rows": rows, "seconds": time.time() - t0}
    del sk, refs
    # ---- data set 2: genome-like all-vs-all ----
    t0 = time.time()
    genomes, fam = workloads.genome_like(6000, 3, 4, 300_000 if args.quick else 1_000_000)
    sk = OracleSketch()
    sk.add_drafts(list(range(len(genomes))), genomes, threads=cores)
    sk.index()
    gl = []
    for q in genomes:
        hits, det = sk.query_draft(q, threads=cores, details=True)
        maps, rows = _detail(det)
        gl.append({"hits": hits, "mappings": maps, "rows": rows})
    out["genome_like"] = {"workload": f"{len(genomes)} x {len(genomes)} genome-like of {len(genomes[0][0]) / 1e6:g} Mb", "queries": gl, "seconds": time.time() - t0}
    # ---- data set 3: the committed synthetic goldens ----
    fixtures = json.load(open(os.path.join(ROOT, "tests", "golden", "synthetic_goldens.json")))
    fx_out = []
    for fx in fixtures:
        case = fx["case"]
        refs, queries = build_case(case)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sk = OracleSketch(**case["params"])
        for i, r in enumerate(refs):
            sk.add_draft(f"ref{i}", r)
        sk.index()
        got = []
        for q in queries:
            hits, det = sk.query_draft(q, details=True)
            maps, rows = _detail(det)
            got.append({"hits": hits, "mappings": maps, "rows": rows})
        fx_out.append({"name": case["name"], "window": sk.window_size, "queries": got,
                       "matches_fixture": all([[h[0], float(np.float32(h[1])), h[2], h[3]] for h in g_["hits"]] == w["hits"] for g_, w in zip(got, fx["queries"]))})
    out["goldens"] = fx_out
    out["seconds"] = time.time() - t_all
    with open(args.worker, "wb") as f:
        pickle.dump(out, f)


def _diff_query(a, b):
    """Counts of what differs between two result sets of one query: mappings as multisets, rows by genome, hits by name."""
    from collections import Counter
    ma, mb = Counter(a["mappings"]), Counter(b["mappings"])
    only_a, only_b = sum((ma - mb).values()), sum((mb - ma).values())
    ra, rb = a["rows"], b["rows"]
    common = set(ra) & set(rb)
    rows_changed = sum(ra[g] != rb[g] for g in common) + len(set(ra) ^ set(rb))
    d_ani = max([abs(ra[g][1] - rb[g][1]) for g in common], default=0.0)
    d_cnt = max([abs(ra[g][0] - rb[g][0]) for g in common], default=0)
    ha, hb = {h[0]: h for h in a["hits"]}, {h[0]: h for h in b["hits"]}
    hits_changed = sum(tuple(ha[n]) != tuple(hb[n]) for n in set(ha) & set(hb)) + len(set(ha) ^ set(hb))
    order_changed = [h[0] for h in a["hits"] if h[0] in hb] != [h[0] for h in b["hits"] if h[0] in ha]
    return {"mappings": len(a["mappings"]), "mappings_changed": max(only_a, only_b), "rows": len(ra), "rows_changed": rows_changed,
            "hits": len(a["hits"]), "hits_changed": hits_changed, "hit_order_changed": bool(order_changed),
            "hits_appeared_or_vanished": len(set(ha) ^ set(hb)), "max_abs_dANI": d_ani, "max_abs_dmatches": d_cnt}


def _merge(diffs):
    out = {k: sum(d[k] for d in diffs) for k in ("mappings", "mappings_changed", "rows", "rows_changed", "hits", "hits_changed", "hits_appeared_or_vanished")}
    out["hit_order_changed"] = any(d["hit_order_changed"] for d in diffs)
    out["max_abs_dANI"] = max(d["max_abs_dANI"] for d in diffs)
    out["max_abs_dmatches"] = max(d["max_abs_dmatches"] for d in diffs)
    return out


def pins(r):
    protein = r["protein_hits"] == [["BGC0001425", 130, 176], ["BGC0001427", 130, 176]]
    window = r["window_default"] == 24
    # (a genome with EXACT repeats loses matches to itself -- fragments inside copies 2..n tie at 100.0 and fall into the first
    # copy's reference bin; the reference's own Shigella golden shows the effect, 1600/1608 at identity 100.0, test_ani.py:86-91 --
    # so for that case only the identity is a pin)
    selfq = all(s["identity"] == 100.0 and s["fragments"] > 0 and (s["matches"] == s["fragments"] or s["case"] == "with repeats")
                for s in r["self_query"])
    return {"protein_golden_130_176_x2": protein, "window_size_24": window, "self_query_exactly_100": selfq,
            "excluded_by_a_pin": not (protein and window and selfq),
            "protein_hits": r["protein_hits"], "window_default": r["window_default"], "self_query": r["self_query"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", help="(internal) run ONE variant under FA_ORACLE_DEFINES and pickle its results here")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", help="comma-separated variant names (default: all)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_open_rule_sensitivity.json"))
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    tmp = tempfile.mkdtemp(prefix="fa_sens_")
    results = {}
    for name, defines, _ in VARIANTS:
        if args.only and name != "default" and name not in args.only.split(","):
            continue
        path = os.path.join(tmp, f"v{len(results)}.pkl")
        env = dict(os.environ, FA_ORACLE_DEFINES=defines)
        t0 = time.time()
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", path] + (["--quick"] if args.quick else []), env=env, cwd=ROOT)
        with open(path, "rb") as f:
            results[name] = pickle.load(f)
        os.unlink(path)
        print(f"[sensitivity] {name}: {time.time() - t0:.0f} s", file=sys.stderr)
    base = results["default"]
    doc = {"what": "each OPEN reading of the absent upstream C++ (oracle/fastani_oracle.hpp, FO_* switches) against the default reading: the three in-tree pins, and what moves",
           "script": "scripts/oracle_sensitivity.py" + (" --quick" if args.quick else ""),
           "default_matches_committed_goldens": all(g["matches_fixture"] for g in base["goldens"]),
           "workloads": {"config2": base["config2"]["workload"], "genome_like": base["genome_like"]["workload"],
                         "goldens": [g["name"] for g in base["goldens"]]},
           "variants": []}
    for name, defines, text in VARIANTS:
        if name not in results:
            continue
        r = results[name]
        entry = {"variant": name, "defines": defines, "meaning": text, "pins": pins(r), "seconds": r["seconds"]}
        if name != "default":
            entry["config2"] = _diff_query(base["config2"], r["config2"])
            entry["genome_like"] = _merge([_diff_query(a, b) for a, b in zip(base["genome_like"]["queries"], r["genome_like"]["queries"])])
            entry["goldens"] = _merge([_diff_query(a, b) for ga, gb in zip(base["goldens"], r["goldens"]) for a, b in zip(ga["queries"], gb["queries"])])
            entry["window_table_changed"] = {k: [base["window_table"][k], v] for k, v in r["window_table"].items() if base["window_table"][k] != v}
        else:
            entry["config2"] = {"mappings": len(base["config2"]["mappings"]), "rows": len(base["config2"]["rows"]), "hits": len(base["config2"]["hits"])}
            entry["genome_like"] = {"mappings": sum(len(q["mappings"]) for q in base["genome_like"]["queries"]),
                                    "rows": sum(len(q["rows"]) for q in base["genome_like"]["queries"]),
                                    "hits": sum(len(q["hits"]) for q in base["genome_like"]["queries"])}
        doc["variants"].append(entry)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({v["variant"]: {"excluded": v["pins"]["excluded_by_a_pin"], **({k: v["config2"][k] for k in ("mappings_changed", "rows_changed", "hits_changed", "max_abs_dANI", "max_abs_dmatches")} if v["variant"] != "default" else {})}
                      for v in doc["variants"]}, indent=1))


if __name__ == "__main__":
    main()

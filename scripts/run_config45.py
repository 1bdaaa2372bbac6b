"""BASELINE configs 4 and 5 on one MI355X, checked through oracle-free properties.

  python scripts/run_config45.py 4 [families members]      500 draft assemblies (50 log-normal contigs, ~5 Mb), all-vs-all
  python scripts/run_config45.py 5 [families members len]  200 genomes, all-vs-all for k in {14,16,21} x frag in {1000,3000,5000}

Properties: every genome hits itself at exactly 100.0 with (nearly) all of its fragments, hits stay inside the family,
hit membership is symmetric ("exactly" up to the end-of-contig effect the oracle shows too: >= 99.999).  (k=21, frag=1000) is the degenerate cell: no window fits a fragment, nothing maps."""
import sys, os, time, json, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import synthetic as syn

which = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def all_vs_all(genomes, fam, params, chunk=24):
    """genomes: list of contig lists.  Returns a dict of timings, counts and property verdicts."""
    n = len(genomes)
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sk = pf.Sketch(**params)
        for i, contigs in enumerate(genomes):
            sk.add_draft(i, contigs)
        t_pack = time.time() - t0
        t0 = time.time()
        mapper = sk.index()
        t_index = time.time() - t0
        t0 = time.time()
        batch = mapper.upload_genomes(genomes)
        t_upload = time.time() - t0
        t0 = time.time()
        rows = [batch.query_rows(i, min(chunk, n - i)) for i in range(0, n, chunk)]
        t_map = time.time() - t0
    rows = np.concatenate(rows)
    frag = mapper.fragment_length
    qlen = batch.total_length.astype(np.float64)
    rlen = np.array([sum((len(c) // frag) * frag for c in contigs) for contigs in genomes], dtype=np.float64)
    # the reference's minimum_fraction filter (_fastani.pyx:1121-1132), float32 like the product path
    min_len = np.minimum(qlen[rows["query_id"]], rlen[rows["ref_genome_id"]]).astype(np.float32)
    keep = (rows["count_seq"].astype(np.float32) * np.float32(frag)) >= min_len * np.float32(mapper.minimum_fraction)
    hits = rows[keep]
    fam = np.asarray(fam)
    self_rows = rows[rows["query_id"] == rows["ref_genome_id"]]
    with_frags = int((batch.total_fragments > 0).sum())
    # a self mapping is exact except for the oracle-confirmed end-of-contig effect (the slide stops when the last record
    # is admitted, so a fragment that ends exactly at the contig end can miss one minimizer) and bin collisions
    ok_self = (len(self_rows) == with_frags and bool(np.all(self_rows["identity"] >= 99.999))
               and bool(np.all(self_rows["count_seq"] >= 0.98 * self_rows["total_query_fragments"])))
    ok_family = bool(np.all(fam[hits["query_id"]] == fam[hits["ref_genome_id"]]))
    pairs = set(zip(hits["query_id"].tolist(), hits["ref_genome_id"].tolist()))
    asym = sum((b, a) not in pairs for a, b in pairs)
    return {
        "window_size": mapper.window_size, "pairs": n * n, "rows": int(len(rows)), "hits_after_min_fraction": int(len(hits)),
        "index_minimizers": len(mapper.minimizers), "threshold": mapper.occurences_threshold,
        "host_pack_s": t_pack, "sketch_index_s": t_index, "upload_queries_s": t_upload, "map_s": t_map,
        "pairs_per_s_map_only": n * n / t_map if t_map > 0 else None,
        "self_rows": int(len(self_rows)), "self_identity_min": float(self_rows["identity"].min()) if len(self_rows) else None,
        "self_fraction_min": float((self_rows["count_seq"] / np.maximum(self_rows["total_query_fragments"], 1)).min()) if len(self_rows) else None,
        "self_hits_exact": ok_self, "hits_within_family": ok_family, "asymmetric_pairs": asym,
    }


if which == 4:
    families = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    members = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    length = int(sys.argv[4]) if len(sys.argv) > 4 else 5_000_000
    g = syn.rng(3000)
    t0 = time.time()
    genomes, fam = [], []
    for f in range(families):
        anc = syn.random_codes(g, length)
        for m in range(members):
            d = 0.0 if m == 0 else syn.DIVERGENCES[m % len(syn.DIVERGENCES)]
            seq = syn.to_ascii(syn.mutate_codes(g, anc, d) if d else anc)
            genomes.append(syn.split_contigs(g, seq, 50))
            fam.append(f)
    t_gen = time.time() - t0
    out = all_vs_all(genomes, fam, {})
    out = {"config": f"4: {len(genomes)} draft assemblies ({families} families x {members}), 50 contigs each, {length / 1e6:g} Mb, add_draft path",
           "generate_s": t_gen, **out}
    print(json.dumps(out))
else:
    families = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    members = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    length = int(sys.argv[4]) if len(sys.argv) > 4 else 5_000_000
    g = syn.rng(4000)
    t0 = time.time()
    genomes, fam = [], []
    for f in range(families):
        anc = syn.random_codes(g, length)
        for m in range(members):
            d = 0.0 if m == 0 else syn.DIVERGENCES[m % len(syn.DIVERGENCES)]
            genomes.append([syn.to_ascii(syn.mutate_codes(g, anc, d) if d else anc)])
            fam.append(f)
    t_gen = time.time() - t0
    cells = []
    only = os.environ.get("CELLS")          # e.g. CELLS=21:1000,16:3000
    for k in (14, 16, 21):
        for frag in (1000, 3000, 5000):
            if only and f"{k}:{frag}" not in only.split(","):
                continue
            r = all_vs_all(genomes, fam, {"k": k, "fragment_length": frag})
            if r["window_size"] >= frag:      # degenerate cell: nothing can map
                r["self_hits_exact"] = r["rows"] == 0
            cells.append({"k": k, "fragment_length": frag, **r})
            print(json.dumps(cells[-1]), file=sys.stderr, flush=True)
    print(json.dumps({"config": f"5: {len(genomes)} genomes ({families} families x {members}) of {length / 1e6:g} Mb, all-vs-all per (k, fragment_length) cell",
                      "generate_s": t_gen, "cells": cells}))

"""The BASELINE.json workloads (configs 2-5) as seeded generators, plus the all-vs-all runner with the
oracle-free property checks that hold at full size.  Shared by ``bench.py``, ``scripts/run_config*.py`` and
``tests/test_gpu_fullsize.py`` so that the driver-run suite maps exactly what the bench and the scripts map.

SURVEY.md section 8d fixes the shapes: config 2 = seed 1000, one query (d = 0.05 from ancestor A) x 100 references
(60 of family A at mixed divergence, 40 unrelated); config 3 = seed 2000, families x members genomes all-vs-all;
config 4 = seed 3000, draft assemblies of 50 log-normal contigs; config 5 = seed 4000, the (k, fragment_length) grid.
"""
import time
import warnings

import numpy as np

from . import synthetic as syn

CONFIG5_CELLS = [(k, frag) for k in (14, 16, 21) for frag in (1000, 3000, 5000)]


def config2_references(n_refs=100, length=5_000_000, seed=1000):
    """(ancestor codes, names, reference genomes as one-contig lists) of BASELINE config 2 -- the bench workload."""
    n_related = int(round(n_refs * 0.6))
    g = syn.rng(seed)
    anc = syn.random_codes(g, length)
    names, refs = [], []
    for i in range(n_refs):
        if i < n_related:
            d = syn.DIVERGENCES[i % len(syn.DIVERGENCES)]
            names.append(f"A{i:03d}")
            refs.append([syn.to_ascii(syn.mutate_codes(g, anc, d))])
        else:
            names.append(f"U{i:03d}")
            refs.append([syn.to_ascii(syn.random_codes(g, length))])
    return anc, names, refs


def config2_query(anc, rank=0, count=1):
    """The query genome(s) of rank `rank`: d = 0.05 from the ancestor, seed 5000 + rank."""
    gq = syn.rng(5000 + rank)
    return [[syn.to_ascii(syn.mutate_codes(gq, anc, 0.05))] for _ in range(count)]


def families(seed, n_families, n_members, length, contigs=0):
    """n_families x n_members genomes (member 0 is the ancestor itself); returns (genomes as contig lists, family ids)."""
    g = syn.rng(seed)
    genomes, fam = [], []
    for f in range(n_families):
        anc = syn.random_codes(g, length)
        for m in range(n_members):
            d = 0.0 if m == 0 else syn.DIVERGENCES[m % len(syn.DIVERGENCES)]
            seq = syn.to_ascii(syn.mutate_codes(g, anc, d) if d else anc)
            genomes.append(syn.split_contigs(g, seq, contigs) if contigs else [seq])
            fam.append(f)
    return genomes, np.asarray(fam)


def config3(n_families=20, n_members=50, length=5_000_000):
    return families(2000, n_families, n_members, length)


def config4(n_families=10, n_members=50, length=5_000_000):
    return families(3000, n_families, n_members, length, contigs=50)


def config5(n_families=10, n_members=20, length=5_000_000):
    return families(4000, n_families, n_members, length)


def _plant(g, anc, element, copies, half_reversed, taken):
    """Overwrite `copies` non-overlapping places of `anc` with `element` (every second copy reverse-complemented if asked)."""
    n, m = len(anc), len(element)
    placed = 0
    while placed < copies:
        at = int(g.integers(0, n - m))
        if any(at < b and a < at + m for a, b in taken):
            continue
        taken.append((at, at + m))
        anc[at: at + m] = syn.reverse_complement_codes(element) if (half_reversed and placed % 2) else element
        placed += 1


def genome_like_ancestor(g, length):
    """A random ancestor with what real bacterial chromosomes have and i.i.d. sequence has not: 7 copies of a 5 kb element
    (rRNA-operon-like), 30 copies of a 1.3 kb element, half of them on the other strand (IS-like), and three 300 bp
    low-complexity tracts (a homopolymer, a dinucleotide whose k-mers are their own reverse complement, a trinucleotide)."""
    anc = syn.random_codes(g, length)
    taken = []
    _plant(g, anc, syn.random_codes(g, 5000), 7, False, taken)
    _plant(g, anc, syn.random_codes(g, 1300), 30, True, taken)
    for unit in ([0], [0, 3], [1, 0, 2]):                     # A..., ATAT..., CAGCAG...
        _plant(g, anc, np.resize(np.asarray(unit, np.uint8), 300), 1, False, taken)
    return anc


def indel_codes(g, codes, n_events, max_len=50):
    """`n_events` insertions / deletions (equally likely) of length 1..max_len at random places."""
    if n_events <= 0:
        return codes
    at = np.sort(g.integers(0, len(codes), n_events))
    lens = g.integers(1, max_len + 1, n_events)
    ins = g.random(n_events) < 0.5
    pieces, a = [], 0
    for p, l, i in zip(at.tolist(), lens.tolist(), ins.tolist()):
        if p < a:
            continue                                          # (inside the stretch the event before it deleted)
        pieces.append(codes[a:p])
        if i:
            pieces.append(syn.random_codes(g, l))
            a = p
        else:
            a = min(p + l, len(codes))
    pieces.append(codes[a:])
    return np.concatenate(pieces)


def genome_like(seed, n_families, n_members, length, indel_share=0.01, inversion=100_000):
    """Genome-LIKE families (the reference's benchmark runs on real assemblies, benches/mapping/bench.py:25-29; SURVEY.md 8d offers
    "optional 1 % indel events of length 1-50 and one inversion of 100 kb"): every ancestor carries repeats and low-complexity
    tracts (`genome_like_ancestor`); member 0 is the ancestor, the others are substituted at the usual divergences, then one
    mutation event in a hundred (`indel_share` of the d x length substitutions) is an insertion or deletion of 1-50 bases, then
    one `inversion`-long segment is reverse-complemented.  Returns (genomes as one-contig lists, family ids)."""
    g = syn.rng(seed)
    genomes, fam = [], []
    for f in range(n_families):
        anc = genome_like_ancestor(g, length)
        for m in range(n_members):
            d = 0.0 if m == 0 else syn.DIVERGENCES[m % len(syn.DIVERGENCES)]
            codes = anc
            if d:
                codes = indel_codes(g, syn.mutate_codes(g, anc, d), int(round(d * length * indel_share)))
                inv = min(inversion, len(codes) // 10)
                if inv > 0:
                    at = int(g.integers(0, len(codes) - inv))
                    codes = codes.copy()
                    codes[at: at + inv] = syn.reverse_complement_codes(codes[at: at + inv])
            genomes.append([syn.to_ascii(codes)])
            fam.append(f)
    return genomes, np.asarray(fam)


def write_fasta(path, contigs, width=60, prefix="contig"):
    """One genome as a FASTA file: a record per contig, `width`-column lines (numpy: no Python loop over lines)."""
    with open(path, "wb") as f:
        for i, c in enumerate(contigs):
            a = np.frombuffer(bytes(c), np.uint8) if not isinstance(c, np.ndarray) else c.astype(np.uint8, copy=False)
            f.write(b">%s_%d\n" % (prefix.encode(), i))
            whole = len(a) // width * width
            if whole:
                lines = np.empty((whole // width, width + 1), np.uint8)
                lines[:, :width] = a[:whole].reshape(-1, width)
                lines[:, width] = 10
                f.write(lines.tobytes())
            if whole < len(a):
                f.write(a[whole:].tobytes() + b"\n")


def write_fasta_set(directory, genomes, width=60):
    """Every genome of a workload as ``directory/g<i>.fna``; returns (paths, total bytes)."""
    import os
    os.makedirs(directory, exist_ok=True)
    paths = []
    for i, contigs in enumerate(genomes):
        p = os.path.join(directory, f"g{i:05d}.fna")
        write_fasta(p, contigs, width, prefix=f"g{i}")
        paths.append(p)
    return paths, sum(os.path.getsize(p) for p in paths)


def row_properties(rows, batch, mapper, genomes, fam):
    """What must hold for the hit rows of an all-vs-all run without an oracle (see `all_vs_all`): counts and verdicts."""
    n = len(genomes)
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
    single = all(len(c) == 1 for c in genomes)
    ident_ok = bool(np.all(self_rows["identity"] == 100.0)) if single else bool(np.all(self_rows["identity"] >= 99.999))
    ok_self = (len(self_rows) == with_frags and bool(np.all(self_rows["identity"] >= 99.999))
               and bool(np.all(self_rows["count_seq"] >= 0.98 * self_rows["total_query_fragments"])))
    ok_family = bool(np.all(fam[hits["query_id"]] == fam[hits["ref_genome_id"]]))
    pairs = set(zip(hits["query_id"].tolist(), hits["ref_genome_id"].tolist()))
    asym = sum((b, a) not in pairs for a, b in pairs)
    return {
        "window_size": mapper.window_size, "pairs": n * n, "rows": int(len(rows)), "hits_after_min_fraction": int(len(hits)),
        "index_minimizers": len(mapper.minimizers), "threshold": mapper.occurences_threshold,
        "self_rows": int(len(self_rows)), "self_identity_min": float(self_rows["identity"].min()) if len(self_rows) else None,
        "self_identity_all_exact": ident_ok,
        "self_fraction_min": float((self_rows["count_seq"] / np.maximum(self_rows["total_query_fragments"], 1)).min()) if len(self_rows) else None,
        "self_hits_exact": ok_self, "hits_within_family": ok_family, "asymmetric_pairs": asym,
    }


def all_vs_all(genomes, fam, params=None, chunk=24, timings=True):
    """Index `genomes`, map every genome against the index and check what must hold without an oracle:

    * every genome with at least one fragment hits itself with identity >= 99.999 (exactly 100.0 for one-contig
      genomes) and (nearly) all of its fragments -- the self-query invariant of test_ani.py:66-71; the slack covers
      the end-of-contig effect the oracle shows too (the slide stops when the last record is admitted) and two
      fragments falling into one reference bin;
    * hits that survive the minimum-fraction filter stay inside the family, and hit membership is symmetric.

    Returns a dict of counts, timings and verdicts; the caller asserts the verdicts that apply to its cell."""
    import ctypes as C
    import pyfastani_amd as pf
    from ._lib import lib
    params = params or {}
    n = len(genomes)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.time()
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
        rows, retries, phase = [], 0, np.zeros(5)
        for i in range(0, n, chunk):
            rows.append(batch.query_rows(i, min(chunk, n - i)))
            ms = (C.c_float * 16)()
            lib.fa_mapper_last_timings(mapper._h, ms, 16)
            retries += int(ms[9])
            phase += np.array(list(ms)[:5])
        t_map = time.time() - t0
    rows = np.concatenate(rows)
    out = row_properties(rows, batch, mapper, genomes, fam)
    out["repeated_attempts"] = retries
    if timings:
        out.update({"host_pack_s": t_pack, "sketch_index_s": t_index, "upload_queries_s": t_upload, "map_s": t_map,
                    "pairs_per_s_map_only": n * n / t_map if t_map > 0 else None,
                    "pairs_per_s_with_index": n * n / (t_map + t_index + t_upload + t_pack),
                    "device_phase_ms": dict(zip(["sketch", "lookup_l1", "l2", "cgi", "total"], [float(x) for x in phase]))})
    out["_rows"] = rows
    out["_mapper"] = mapper
    return out

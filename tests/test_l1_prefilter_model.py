"""The block pre-filter designed for k_l1 (profiles/EXPERIMENTS.md, "Design note for the block pre-filter"), as a CPU model against
the oracle: dropping the seed hits that cannot belong to a candidate region must leave the candidate regions unchanged.

Claim: computeL1CandidateRegions flags seed i when seed i + minHits - 1 lies in the same contig less than a fragment length of
window positions further on.  Window positions grow by at least one per reference record, so the minHits seeds of such a run lie
within a fragment length of RECORD NUMBERS -- inside one cell of one of two staggered grids of cells twice that wide.  A seed whose
cells (one per grid) both hold fewer than minHits seeds is dead; removing dead seeds removes no run and creates none.

The model applies exactly that filter (exact cell counts here; the kernel design uses hashed bit arrays, which can only keep more) to
the seed hits of every fragment of a query and compares the candidate regions -- the oracle's rule restated over numpy arrays, first
checked against the oracle's own L1 output -- with and without it, on an index whose position lists also produce chance hits."""
import numpy as np

from oracle.oracle import OracleSketch
from pyfastani_amd import synthetic as syn


def _candidates(seeds, s, w, min_hits, qlen):
    """computeL1CandidateRegions over sorted record numbers (records are ordered by (contig, window position))."""
    out = []
    m = max(min_hits, 1)
    for i in range(len(seeds) - m + 1):
        a, b = seeds[i], seeds[i + m - 1]
        if s[a] == s[b] and w[b] - w[a] < qlen:
            c = [int(s[a]), max(0, int(w[b]) - qlen + 1), int(w[a])]
            if out and out[-1][0] == c[0] and out[-1][2] >= c[1]:
                out[-1][2] = max(out[-1][2], c[2])
            else:
                out.append(c)
    return [tuple(c) for c in out]


def _live(seeds, min_hits, qlen):
    half = 1
    while half < qlen:
        half *= 2
    keep = np.zeros(len(seeds), bool)
    for g in (0, 1):
        cell = (seeds + g * half) // (2 * half)
        _, inverse, counts = np.unique(cell, return_inverse=True, return_counts=True)
        keep |= counts[inverse] >= max(min_hits, 2)
    return keep


def test_dropping_dead_seeds_leaves_the_candidate_regions_unchanged():
    g = syn.rng(8101)
    k, flen = 14, 1000                                                   # k = 14: 28-bit hashes, chance hits in a small index
    anc = syn.random_codes(g, 150_000)
    refs = [[syn.to_ascii(syn.mutate_codes(g, anc, d))] for d in (0.02, 0.06, 0.1)]
    refs += [[syn.to_ascii(syn.random_codes(g, 2_000_000))] for _ in range(5)]        # unrelated genomes: chance hits only
    osk = OracleSketch(k=k, fragment_length=flen)
    for i, r in enumerate(refs):
        osk.add_draft(f"r{i}", r)
    osk.index()
    h, s, w = osk.minimizers()
    order = np.argsort(h, kind="stable")
    hs = h[order]
    thr = osk.freq_threshold
    query = syn.to_ascii(syn.mutate_codes(g, anc, 0.04))
    fragments = dropped = with_regions = 0
    for f in range(0, len(query) // flen, 3):
        frag = query[f * flen:(f + 1) * flen]
        ssize, min_hits, l1 = osk.l1_fragment(frag)
        qh, _ = osk.sketch_sequence(frag)
        seeds = []
        for x in sorted(set(int(v) for v in qh)):
            lo, hi = np.searchsorted(hs, x, "left"), np.searchsorted(hs, x, "right")
            if 0 < hi - lo < thr:
                seeds.append(order[lo:hi])
        seeds = np.sort(np.concatenate(seeds)) if seeds else np.zeros(0, np.int64)
        full = _candidates(seeds, s, w, min_hits, flen)
        assert full == [tuple(c) for c in l1]                            # the restated rule IS the oracle's
        if min_hits < 2:
            continue                                                     # (one hit makes a candidate: nothing is dead)
        keep = _live(seeds, min_hits, flen)
        assert _candidates(seeds[keep], s, w, min_hits, flen) == full
        fragments += 1
        dropped += int((~keep).sum())
        with_regions += bool(full)
    print(f"{fragments} fragments with minHits >= 2, {with_regions} with candidate regions, {dropped} dead seeds dropped")
    assert fragments >= 20 and with_regions >= 10 and dropped >= 200     # the filter had something to do, and regions to keep

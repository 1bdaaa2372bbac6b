"""The oracle's sliding map against the DEFINITION of what it maintains (CPU; no GPU, no HIP library).

`oracle/fastani_oracle.hpp: SlideMapper` restates skch::SlideMapper (slidingMap.hpp): an ordered map over the union of the query
sketch and the reference minimizers inside the super-window, a pivot at the s-th smallest key, and a counter of keys up to the
pivot that carry both a query and a reference position, kept up to date through insertions and deletions with five status
cases.  What that counter must equal at every window position is a statement about SETS:

    shared(p) = | { h among the s smallest of (Q u W_p) : h in Q and h in W_p } |,   W_p = hashes of the records in the window,

and this test recomputes it that way -- sorted(set | set)[:s] -- for every candidate region of every fragment of a query, walks the
window positions exactly as `compute_l2` does (same admissions and drops: that rule is NOT what is tested here), and compares the
optimum (count, first and last optimal position -> mapped position) with the mappings the oracle reports.  It pins the incremental
bookkeeping -- the part of the slide that the HIP kernels replace by a rank-space event stream -- to its definition, independently
of either implementation; the window-advance rule itself stays pinned by the self-query invariant (tests/test_gpu_fullsize.py)."""
import bisect

import numpy as np

from oracle.oracle import OracleSketch
from pyfastani_amd import synthetic as syn


def _loci_by_definition(osk, frag_bytes, h, s, w, params):
    """[(seq, mean optimal position, shared)] of one fragment: candidate regions from the oracle's L1 stage, the slide by sets."""
    k, window, flen = params
    qh, _ = osk.sketch_sequence(frag_bytes)
    q = sorted(set(int(x) for x in qh))
    ssize, min_hits, l1 = osk.l1_fragment(frag_bytes)
    assert ssize == len(q)
    qset = set(q)
    cmw = flen - (window - 1) - (k - 1)
    keys = list(zip(s.tolist(), w.tolist()))                             # records are sorted by (contig, window position)

    def search_index(seq, pos):
        return bisect.bisect_left(keys, (seq, pos))

    def shared_of(beg, end):
        wset = set(int(x) for x in h[beg:end])
        bottom = sorted(qset | wset)[:ssize]
        return sum(1 for x in bottom if x in qset and x in wset)

    out = []
    for seq, rs, re_ in l1:
        beg = search_index(seq, rs)
        p = int(w[beg])
        end = search_index(seq, p + cmw)
        last = search_index(seq, re_ + cmw)
        best, o_start, o_end, first = 0, beg, beg, True
        cur = shared_of(beg, end)
        while True:
            if first or cur > best:
                best, o_start, o_end, first = cur, beg, beg, False
            elif cur == best:
                o_end = beg
            if end >= last:
                break
            p += 1
            changed = False
            if beg + 1 < len(keys) and s[beg + 1] == seq and w[beg + 1] <= p:
                beg += 1
                changed = True
            if end < last and w[end] <= p + cmw - 1:
                end += 1
                changed = True
            if changed:
                cur = shared_of(beg, end)
        out.append((int(seq), (int(w[o_start]) + int(w[o_end])) // 2, best))
    return ssize, out


def _check(params, seed, n_frag):
    k, flen, pid = params
    g = syn.rng(seed)
    anc = syn.random_codes(g, 60_000)
    anc[20_000:23_000] = anc[5_000:8_000]                                # a repeat: the same hashes at two places of a contig
    anc[30_000:30_600] = anc[29_400:30_000]                              # a tandem duplication: the same hash TWICE inside one window (the
    anc[41_000:41_250] = anc[40_750:41_000]                              # REV / NOOP cases of the sliding map), at two distances
    anc[50_000:50_400] = 0                                               # a low-complexity run: one hash in many consecutive windows
    refs = [[syn.to_ascii(syn.mutate_codes(g, anc, d))] for d in (0.0, 0.03, 0.08, 0.14)]
    refs.append([syn.to_ascii(anc[:25_000]), syn.to_ascii(anc[25_000:])])   # a draft: regions end at contig ends
    osk = OracleSketch(k=k, fragment_length=flen, percentage_identity=pid)
    for i, r in enumerate(refs):
        osk.add_draft(f"r{i}", r)
    osk.index()
    h, s, w = osk.minimizers()
    query = syn.mutate_codes(g, anc, 0.05)
    qbytes = syn.to_ascii(query)
    _, det = osk.query_draft([qbytes], details=True)
    m = det["mappings"]
    by_frag = {}
    for i in range(len(m["qseq"])):
        by_frag.setdefault(int(m["qseq"][i]), []).append((int(m["rseq"][i]), int(m["rstart"][i]), int(m["shared"][i]), int(m["sketch"][i])))
    assert len(by_frag) >= n_frag
    checked = 0
    for f in sorted(by_frag)[:n_frag]:
        frag = qbytes[f * flen:(f + 1) * flen]
        ssize, loci = _loci_by_definition(osk, frag, h, s, w, (k, osk.window_size, flen))
        assert len(loci) >= len(by_frag[f])
        for rseq, rstart, shared, sketch in by_frag[f]:
            assert sketch == ssize
            assert (rseq, rstart, shared) in loci, (f, (rseq, rstart, shared), loci)
            checked += 1
    return checked


def test_slide_counter_equals_its_set_definition_default_cell():
    assert _check((16, 3000, 80.0), 7001, 19) >= 70


def test_slide_counter_equals_its_set_definition_small_window():
    # w = 13 at fragment length 1000: three times the minimizers per base, sketches of ~140
    assert _check((16, 1000, 80.0), 7002, 30) >= 90


def test_slide_counter_equals_its_set_definition_k21():
    # k = 21, w = 15: the slowest live cell of config 5
    assert _check((21, 3000, 80.0), 7003, 12) >= 40

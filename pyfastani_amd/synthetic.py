"""Seeded synthetic genomes for tests and benchmarks.

``numpy.random.Generator(PCG64(seed))``; uniform ACGT ancestors; family members by independent
substitution at a per-genome divergence; optional draft assemblies (log-normal contig lengths).
Genomes are returned as ``numpy.uint8`` arrays of ASCII bytes.
"""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
DIVERGENCES = (0.01, 0.03, 0.05, 0.10, 0.15, 0.20)


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def random_codes(g, n):
    return g.integers(0, 4, n, dtype=np.uint8)


def mutate_codes(g, codes, d):
    """Substitute each base independently with probability d (uniform over the three other bases)."""
    out = codes.copy()
    mask = g.random(len(codes)) < d
    out[mask] = (out[mask] + g.integers(1, 4, int(mask.sum()), dtype=np.uint8)) % 4
    return out


def to_ascii(codes):
    return ACGT[codes]


def reverse_complement_codes(codes):
    return (3 - codes)[::-1].copy()


def family(seed, n_members, length, divergences=DIVERGENCES):
    """An ancestor and n_members substitution-only descendants (ASCII arrays) plus their divergences."""
    g = rng(seed)
    anc = random_codes(g, length)
    ds = [divergences[i % len(divergences)] for i in range(n_members)]
    return anc, [to_ascii(mutate_codes(g, anc, d)) for d in ds], ds


def split_contigs(g, seq, n_contigs):
    """Cut a genome into n_contigs pieces with log-normal lengths (draft assembly)."""
    w = g.lognormal(0.0, 0.6, n_contigs)
    cuts = np.floor(np.cumsum(w) / w.sum() * len(seq)).astype(np.int64)
    cuts[-1] = len(seq)
    out, a = [], 0
    for b in cuts:
        if b > a:
            out.append(seq[a:b])
        a = b
    return out


def config2(seed=1000, length=5_000_000, n_related=60, n_unrelated=40):
    """BASELINE config 2: one query (d=0.05 from ancestor A) x (n_related family-A + n_unrelated) references."""
    g = rng(seed)
    anc = random_codes(g, length)
    refs, names = [], []
    for i in range(n_related):
        d = DIVERGENCES[i % len(DIVERGENCES)]
        refs.append(to_ascii(mutate_codes(g, anc, d)))
        names.append(f"A{i:03d}_d{d:.2f}")
    for i in range(n_unrelated):
        refs.append(to_ascii(random_codes(g, length)))
        names.append(f"U{i:03d}")
    query = to_ascii(mutate_codes(g, anc, 0.05))
    return query, refs, names

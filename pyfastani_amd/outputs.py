"""FastANI-style outputs built from the hit table (upstream `outputCGI` / `outputPhylip`,
include/fastani/cgi/compute_core_identity.pxd:39-51 -- not exposed by pyfastani, provided here for all-vs-all runs).

The rows are ``cgi::CGI_Results`` records (``pyfastani_amd._batch.ROW_DTYPE``): query_id, ref_genome_id, count_seq,
total_query_fragments, identity.
"""
import numpy as np


def filter_rows(rows, query_lengths, reference_lengths, fragment_length, minimum_fraction=0.2):
    """The reference's hit filter (_fastani.pyx:1121-1132) applied to a whole hit table: keep a row when the mapped
    fragments cover at least ``minimum_fraction`` of the shorter genome (float32 arithmetic, like the reference)."""
    q = np.asarray(query_lengths, dtype=np.float64)[rows["query_id"]]
    r = np.asarray(reference_lengths, dtype=np.float64)[rows["ref_genome_id"]]
    min_len = np.minimum(q, r).astype(np.float32)
    shared = rows["count_seq"].astype(np.float32) * np.float32(fragment_length)
    return rows[shared >= min_len * np.float32(minimum_fraction)]


def identity_matrix(rows, n_queries, n_references, symmetric=False):
    """Dense identity matrix (NaN where there is no hit).  With ``symmetric`` (all-vs-all over one genome set) a cell is
    the mean of the two directions when both exist, as FastANI's matrix output does."""
    m = np.full((n_queries, n_references), np.nan, dtype=np.float64)
    m[rows["query_id"], rows["ref_genome_id"]] = rows["identity"]
    if symmetric:
        if n_queries != n_references:
            raise ValueError("a symmetric matrix needs the same genomes as queries and references")
        both = ~np.isnan(m) & ~np.isnan(m.T)
        either = np.where(np.isnan(m), m.T, m)
        m = np.where(both, (m + m.T) / 2.0, either)
    return m


def write_matrix(path, names, matrix):
    """FastANI's ``--matrix`` layout: the number of genomes, then one line per genome with its name and the identities
    to all earlier genomes (lower triangle), ``NA`` where no hit passed the filters."""
    n = len(names)
    with open(path, "w") as f:
        f.write(f"{n}\n")
        for i in range(n):
            cells = ["NA" if np.isnan(matrix[i, j]) else f"{matrix[i, j]:.6f}" for j in range(i)]
            f.write("\t".join([str(names[i])] + cells) + "\n")


def write_hits(path, query_names, reference_names, rows):
    """FastANI's tabular output: query, reference, ANI, mapped fragments, total query fragments (one line per hit,
    queries in order, hits of a query by decreasing identity)."""
    order = np.lexsort((-rows["identity"], rows["query_id"]))
    with open(path, "w") as f:
        for r in rows[order]:
            f.write(f"{query_names[r['query_id']]}\t{reference_names[r['ref_genome_id']]}\t{r['identity']:.6g}\t"
                    f"{r['count_seq']}\t{r['total_query_fragments']}\n")

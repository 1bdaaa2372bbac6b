"""Row layout of the hit table (``cgi::CGI_Results``, include/fastani/cgi/cgid_types.pxd:19-27 of the reference, plus the
query index of a batch) as a numpy structured dtype, and the resident-batch class (implemented in the compiled binding)."""
import numpy as np

from ._fastani import GenomeBatch  # noqa: F401  (re-export)

ROW_DTYPE = np.dtype(
    [("query_id", "<i4"), ("ref_genome_id", "<i4"), ("count_seq", "<i4"), ("total_query_fragments", "<i4"),
     ("identity", "<f4")]
)
assert ROW_DTYPE.itemsize == 20

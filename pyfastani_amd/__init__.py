"""MI355X-native FastANI fragment-mapping engine with pyfastani's Python surface.

Drop-in for the path ``Sketch`` -> ``add_draft``/``add_genome`` -> ``index`` -> ``Mapper.query_draft``/
``query_genome`` of `pyfastani <https://github.com/althonos/pyfastani>`_ (reference:
``src/pyfastani/__init__.py:1-27``).  All compute runs in hand-written HIP kernels for gfx950 behind the C ABI of
``include/fastani_hip.h``; there is no CPU fallback.
"""
from ._api import (
    MAX_KMER_SIZE,
    Hit,
    Mapper,
    MinimizerIndex,
    MinimizerInfo,
    Minimizers,
    Position,
    Sketch,
)
from ._batch import GenomeBatch

__all__ = [
    "MAX_KMER_SIZE", "Hit", "Mapper", "MinimizerIndex", "MinimizerInfo", "Minimizers", "Position", "Sketch",
    "GenomeBatch",
]
__version__ = "0.1.0"

"""MI355X-native FastANI fragment-mapping engine with pyfastani's Python surface.

Drop-in for the path ``Sketch`` -> ``add_draft``/``add_genome`` -> ``index`` -> ``Mapper.query_draft``/
``query_genome`` of `pyfastani <https://github.com/althonos/pyfastani>`_ (reference:
``src/pyfastani/__init__.py:1-27``).  The classes live in the Cython module ``pyfastani_amd._fastani`` (the reference's
``_fastani.pyx`` re-based on ``fastani_hip.pxd``); all compute runs in hand-written HIP kernels for
gfx950 behind the C ABI of ``include/fastani_hip.h``.  There is no CPU fallback, and importing the package needs neither
PyTorch nor numpy (``pyfastani_amd.sharding``, the multi-GPU layer, imports torch when it is used).
"""
try:
    from ._fastani import (
        MAX_KMER_SIZE,
        GenomeBatch,
        Hit,
        Mapper,
        MinimizerIndex,
        MinimizerInfo,
        PackedGenomes,
        Minimizers,
        Position,
        Sketch,
        device_count,
        device_trim,
        set_device,
    )
except ImportError as exc:  # the compiled binding or libfastani_hip.so is missing: there is nothing to fall back to
    raise ImportError(
        "pyfastani_amd needs its compiled binding (pyfastani_amd/_fastani*.so) and pyfastani_amd/lib/libfastani_hip.so: "
        "build them with `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950 + cython). "
        f"There is no CPU fallback.  ({exc})"
    ) from exc

__all__ = [
    "MAX_KMER_SIZE", "Hit", "Mapper", "MinimizerIndex", "MinimizerInfo", "Minimizers", "Position", "Sketch",
    "GenomeBatch", "PackedGenomes", "device_count", "device_trim", "set_device",
]
__version__ = "0.2.0"

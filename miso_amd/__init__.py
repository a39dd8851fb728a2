"""miso_amd -- MI355X-native MISO posterior sampler (the `pysplicing` MISO / MISOPaired hot path).

Everything numeric runs in libmiso_amd.so (HIP, gfx950); this package is the thin host mirror of
the reference's Python-facing interface (pysplicing module, misopy/miso_sampler.py).
"""
from . import capi  # noqa: F401
from .capi import Batch, Gene, InternalError  # noqa: F401

__all__ = ["capi", "Batch", "Gene", "InternalError"]

"""miso_amd -- MI355X-native MISO posterior sampler (the `pysplicing` MISO / MISOPaired hot path).

Everything numeric runs in libmiso_amd.so (HIP, gfx950); this package is the thin host mirror of
the reference's Python-facing interface (pysplicing module, misopy/miso_sampler.py).
"""
import os as _os

# A batch of whole genes is many kernels side by side; the HIP runtime's default of 4 hardware queues serialises them
# (docs/history.md 4.3 (iv)).  The library's constructor sets this too, but the runtime reads it at the process's FIRST HIP
# call: say it as early as the package is imported, unless the host chose a value.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import capi  # noqa: F401,E402
from .capi import Batch, Gene, InternalError  # noqa: F401,E402

__all__ = ["capi", "Batch", "Gene", "InternalError"]

"""`pysplicing` -- drop-in for the reference's package of the same name
(/root/reference/pysplicing/pysplicing/__init__.py), backed by the MI355X sampler.

    sys.path.insert(0, "<repo>/miso_amd")   # then `import pysplicing` resolves here
"""
from .pysplicing import *  # noqa: F401,F403
from .pysplicing import InternalError  # noqa: F401

# pysplicing/pysplicing/__init__.py:2-13
MISO_START_AUTO = 0
MISO_START_UNIFORM = 1
MISO_START_RANDOM = 2
MISO_START_GIVEN = 3
MISO_START_LINEAR = 4

MISO_STOP_FIXEDNO = 0
MISO_STOP_CONVERGENT_MEAN = 1

MISO_ALGO_REASSIGN = 0
MISO_ALGO_MARGINAL = 1
MISO_ALGO_CLASSES = 2

"""chromosome3d_amd — MI355X (gfx950) hot path of Chromosome3D behind a C ABI.

The product is chromosome3d_amd/csrc (hand-written HIP kernels + C++ host, built into
_lib/libc3d.so and _lib/c3d_solve).  This Python package is the host-side mirror of the
reference driver's interface used by tests, bench.py and the examples; it never computes
anything itself and has no CPU fallback.
"""
from .lib import C3DError, LIB_PATH  # noqa: F401
from .solver import Solver, default_fire, default_model, default_schedule, make_stages  # noqa: F401
from . import pipeline  # noqa: F401

__all__ = ["Solver", "C3DError", "default_model", "default_fire", "default_schedule", "make_stages", "pipeline", "LIB_PATH"]

"""openmm_amd -- MI355X-native HE-linear + modulus-conversion hot path of M&M.

The compute lives in ``libmm_hip.so`` (hand-written HIP for gfx950, C ABI declared in
``include/mm_hip.h``).  This package is the thin host-side binding: it loads the
library, moves buffers with torch, and mirrors the reference's operator names.
There is no CPU fallback: importing :mod:`openmm_amd.lib` without the built library
raises.
"""
from .lib import MMHError, Context, Sampler, load_library  # noqa: F401

"""MI355X-native implementation of Lattigo v1.3.1's ``ring`` hot path.

``ring``      host-side mirror of the reference's ring.Context / FastBasisExtender / Decomposer
              API over the C ABI of include/lattigo_ring.h (hand-written gfx950 HIP kernels).
``params``    the reference's parameter sets, regenerated.
``sampling``  reproducible synthetic operands.
"""
from . import _native, params, ring, sampling  # noqa: F401

__all__ = ["ring", "params", "sampling", "_native"]

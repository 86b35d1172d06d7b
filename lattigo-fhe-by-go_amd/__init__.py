"""MI355X-native implementation of Lattigo v1.3.1's ``ring`` hot path.

``ring``      host-side mirror of the reference's ring.Context / FastBasisExtender / Decomposer
              API over the C ABI of include/lattigo_ring.h (hand-written gfx950 HIP kernels).
``params``    the reference's parameter sets, regenerated.
``sampling``  reproducible synthetic operands.
``sharding``  batch partition across one process per GPU and the gather of results (the path's only collective).
"""
from . import _native, params, ring, sampling, sharding  # noqa: F401

__all__ = ["ring", "params", "sampling", "sharding", "_native"]

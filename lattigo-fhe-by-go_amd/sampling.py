"""Reproducible synthetic operands.

The reference fills benchmark operands with ``Context.NewUniformPoly`` (ring/sampler.go:11-64):
draw 8 random bytes, mask to the modulus' bit length, reject values >= q.  Its source is
crypto/rand; ours is splitmix64 with a fixed seed so that the CPU oracle and the GPU see the
same inputs.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

DEFAULT_SEED = 0x4C415454  # "LATT"


def splitmix64(seed, count):
    """`count` outputs of splitmix64 started at `seed` (vectorised: state_i = seed + (i+1)*golden)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform_limb(q, n, seed):
    """n values uniform in [0, q): mask-and-reject like UniformPoly (ring/sampler.go:50-56)."""
    mask = np.uint64((1 << int(q).bit_length()) - 1)
    out = np.empty(n, dtype=np.uint64)
    have, s = 0, np.uint64(seed)
    while have < n:
        draw = splitmix64(s, 2 * (n - have) + 64) & mask
        with np.errstate(over="ignore"):
            s = s + np.uint64(0xA5A5A5A5) * np.uint64(2 * (n - have) + 64)
        good = draw[draw < np.uint64(q)]
        take = min(len(good), n - have)
        out[have:have + take] = good[:take]
        have += take
    return out


def uniform_poly(moduli, N, batch=1, seed=DEFAULT_SEED):
    """[batch, limbs, N] uint64, limb i uniform in [0, moduli[i])."""
    out = np.empty((batch, len(moduli), N), dtype=np.uint64)
    for b in range(batch):
        for i, q in enumerate(moduli):
            out[b, i] = uniform_limb(q, N, (seed * 0x100000001B3 + b * 1000003 + i * 7919 + 1) & 0xFFFFFFFFFFFFFFFF)
    return out


def random_u64(shape, seed=DEFAULT_SEED):
    """Full-range 64-bit values (NewPolyUniform, ring/ring_object.go:26-47)."""
    n = int(np.prod(shape))
    return splitmix64(seed, n).reshape(shape)

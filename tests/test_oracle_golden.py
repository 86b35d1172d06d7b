"""Pins the CPU oracle against the reference's golden vectors (ring/ntt_test.go:101-142,
ring/test_data/*): all N coefficients of all 14 limb vectors, plus the InvNTT round trip."""
import numpy as np
import pytest

from conftest import GOLDEN_SIZES, golden_pair


@pytest.mark.parametrize("n", GOLDEN_SIZES)
def test_ntt_matches_reference_vectors(oracle, n):
    N, moduli, x, want = golden_pair(n)
    ctx = oracle.Context(N, moduli)
    got = ctx.ntt(x)
    assert np.array_equal(got, want)          # every coefficient (the Go test only looks at two)
    assert np.array_equal(ctx.intt(got), x)   # InvNTT(NTT(x)) == x, ring/ntt_test.go:129-139


def test_golden_primitive_roots(oracle):
    # the two moduli of the fixtures; g = 15 and g = 3 were observed for them
    assert oracle.lib().oc_primitive_root(576460752303439873) == 15
    assert oracle.lib().oc_primitive_root(576460752303702017) == 3


def test_ntt_tolerates_unreduced_input(oracle):
    """NTT returns the canonical transform of (input mod q) for inputs >= q,
    which ring/ring_scaling.go:19,102 relies on."""
    N, moduli, x, want = golden_pair(64)
    ctx = oracle.Context(N, moduli)
    shifted = x.copy()
    for i, q in enumerate(moduli):
        shifted[i] = x[i] + np.uint64(3 * q)
    assert np.array_equal(ctx.ntt(shifted), want)

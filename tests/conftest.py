import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference (test infrastructure)."""
    o = graft.load_oracle()
    o.build()
    return o


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    """The package with the HIP library loaded and a device present; fails loudly otherwise."""
    pkg._native.lib()
    if pkg._native.device_count() < 1:
        pytest.fail("gpu test started without a HIP device: the product has no CPU fallback")
    return pkg


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden", "ring_test_data")


def load_golden(name):
    """ring/test_data format (ring/ntt_test.go:38-99): line 1 N, line 2 moduli, then one line per limb."""
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        lines = f.read().split("\n")
    N = int(lines[0])
    moduli = [int(x) for x in lines[1].split()]
    coeffs = np.array([[int(x) for x in lines[2 + i].split()] for i in range(len(moduli))], dtype=np.uint64)
    assert coeffs.shape == (len(moduli), N)
    return N, moduli, coeffs


GOLDEN_SIZES = [8, 16, 32, 64, 128, 256, 512]


def golden_pair(n):
    tag = str(n).rjust(4, "_")
    N, mod, x = load_golden("test_pol_60_%s_2" % tag)
    N2, mod2, y = load_golden("test_pol_NTT_60_%s_2" % tag)
    assert (N, mod) == (N2, mod2)
    return N, mod, x, y


def crt_reconstruct(limbs, moduli):
    """[limbs, N] residues -> list of N python ints in [0, prod moduli)."""
    Q = 1
    for m in moduli:
        Q *= m
    out = []
    n = limbs.shape[1]
    coefs = [(Q // m) * pow(Q // m, -1, m) for m in moduli]
    for j in range(n):
        v = 0
        for i, _ in enumerate(moduli):
            v += int(limbs[i, j]) * coefs[i]
        out.append(v % Q)
    return out

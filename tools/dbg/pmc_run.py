"""One launch pattern for counter collection: forward NTT R15 (batch 256), a few launches."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 15
N, moduli = params.DefaultParamsQi(logn)
L = len(moduli)
if len(sys.argv) > 2:                      # pmc_run.py 15 40: moduli of that many bits (below 2^46: the FP64 body)
    moduli = params.GenerateNTTPrimes(int(sys.argv[2]), logn, L)
B = (1 << 30) // (8 * N * L)
ctx = ring.NewContextWithParams(N, moduli)
base = sampling.uniform_poly(moduli, N, 2, seed=1)
src, dst = ctx.NewPoly(B).set(np.concatenate([base] * (B // 2))), ctx.NewPoly(B)
for _ in range(5):
    ctx.NTT(src, dst)
ctx.Sync()

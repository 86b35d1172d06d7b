"""One launch pattern for counter collection: K calls of lr_ckks_mulrelin (PN15QP880 or PN16QP1761), nothing else after set-up.
    python tools/dbg/mulrelin_pmc.py PN15QP880 64 4
Prints the number of ciphertext products executed, so that the collector can divide the summed counters by it."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
name = sys.argv[1] if len(sys.argv) > 1 else "PN15QP880"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
N, Q, P = params.ckks_moduli(name)
cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
plan = ring.CkksPlan(cQ, cP, B)
level = len(Q) - 1
beta = -(-len(Q) // len(P))
evk = plan.NewSwitchingKey().set(sampling.uniform_poly(Q + P, N, 2 * beta, seed=9))
base = sampling.uniform_poly(Q, N, 2, seed=3)
host = np.concatenate([base] * (B // 2)) if B >= 2 else base[:1]
mk = lambda: cQ.NewPoly(B).set(host)
ct0, ct1, out = (mk(), mk()), (mk(), mk()), (cQ.NewPoly(B), cQ.NewPoly(B))
for _ in range(K):
    plan.MulRelin(level, ct0, ct1, evk, out)
cQ.Sync()
print("PRODUCTS %d" % (K * B))

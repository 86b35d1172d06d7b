"""Launch pattern for a kernel trace: K x (CKKS RotateColumns of one ciphertext) at PN15QP880 level 17, batch B.   python tools/dbg/rotate_b1.py [B] [K]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N, Q, P = params.ckks_moduli("PN15QP880")
cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
plan = ring.CkksPlan(cQ, cP, B)
level, beta = len(Q) - 1, -(-len(Q) // len(P))
key = plan.NewSwitchingKey().set(sampling.uniform_poly(Q + P, N, 2 * beta, seed=9))
host = sampling.uniform_poly(Q, N, B, seed=3).reshape(B, len(Q), N)
ct = (cQ.NewPoly(B).set(host), cQ.NewPoly(B).set(host))
out = (cQ.NewPoly(B), cQ.NewPoly(B))
gal = pow(5, 3, 2 * N)
for it in range(K + 5):
    if it == 5:
        cQ.Sync()
        t0 = time.perf_counter()
    plan.PermuteNTT(level, ct, gal, key, out)
cQ.Sync()
print("ROTATE us per call: %.1f" % ((time.perf_counter() - t0) / K * 1e6))

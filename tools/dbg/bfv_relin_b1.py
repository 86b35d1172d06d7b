"""Launch pattern for a kernel trace: K x (BFV Relinearize of one degree-2 ciphertext), batch B.   python tools/dbg/bfv_relin_b1.py [PN14QP438] [B] [K]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
name = sys.argv[1] if len(sys.argv) > 1 else "PN14QP438"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = int(sys.argv[3]) if len(sys.argv) > 3 else 30
N, Q, P, QMul = params.bfv_moduli(name)
cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
plan = ring.CkksPlan(cQ, cP, B)
beta = -(-len(Q) // len(P))
key = plan.NewSwitchingKey().set(sampling.uniform_poly(list(Q) + list(P), N, 2 * beta, seed=9))
host = [sampling.uniform_poly(Q, N, B, seed=5 + k).reshape(B, len(Q), N) for k in range(3)]
ct = tuple(cQ.NewPoly(B).set(h) for h in host)
out = (cQ.NewPoly(B), cQ.NewPoly(B))
for it in range(K + 5):
    if it == 5:
        cQ.Sync()
        t0 = time.perf_counter()
    plan.BfvRelinearize(ct, key, out)
cQ.Sync()
print("BFV RELINEARIZE us per call: %.1f" % ((time.perf_counter() - t0) / K * 1e6))

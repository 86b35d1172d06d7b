"""Launch pattern for a kernel trace: K x (BFV Mul of one ciphertext pair), batch B.   python tools/dbg/bfv_b1.py [PN14QP438] [B] [K]"""
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
cQ, cM = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, QMul)
plan = ring.BfvPlan(cQ, cM, 65537, B)
host = [sampling.uniform_poly(Q, N, B, seed=5 + k).reshape(B, len(Q), N) for k in range(4)]
mk = lambda k: cQ.NewPoly(B).set(host[k])
c0, c1 = (mk(0), mk(1)), (mk(2), mk(3))
out = (cQ.NewPoly(B), cQ.NewPoly(B), cQ.NewPoly(B))
for it in range(K + 5):
    if it == 5:
        cQ.Sync()
        t0 = time.perf_counter()
    plan.Mul(c0, c1, out)
cQ.Sync()
print("BFV MUL us per call: %.1f" % ((time.perf_counter() - t0) / K * 1e6))

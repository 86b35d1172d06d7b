"""Launch pattern for a kernel trace: K x (CKKS Rescale of one ciphertext) at PN15QP880 level 17, batch B.   python tools/dbg/rescale_b1.py [B] [K]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling, nat = pkg.ring, pkg.params, pkg.sampling, pkg._native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N, Q, P = params.ckks_moduli("PN15QP880")
cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
plan = ring.CkksPlan(cQ, cP, B)
host = sampling.uniform_poly(Q, N, B, seed=3).reshape(B, len(Q), N)
ct = (cQ.NewPoly(B).set(host), cQ.NewPoly(B).set(host))
import time
for it in range(K + 5):
    if it == 5:
        cQ.Sync()
        t0 = time.perf_counter()
    for p in ct:
        nat.check(nat.lib().lr_poly_set_limbs(p.h, len(Q)))
    plan.Rescale(ct)
cQ.Sync()
print("RESCALE us per call: %.1f" % ((time.perf_counter() - t0) / K * 1e6))

"""MulRelin products per second against the batch of one call (one plan, one stream).   python tools/dbg/mulrelin_sweep.py [PN15QP880] [batches]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
name = sys.argv[1] if len(sys.argv) > 1 else "PN15QP880"
Bs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4,6,8,12,16,24,32,64,128").split(",")]
N, Q, P = params.ckks_moduli(name)
level, beta = len(Q) - 1, -(-len(Q) // len(P))
key_h = sampling.uniform_poly(Q + P, N, 2 * beta, seed=9)
base = sampling.uniform_poly(Q, N, 2, seed=3)
for B in Bs:
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, B)
    key = plan.NewSwitchingKey().set(key_h)
    host = np.concatenate([base] * (-(-B // 2)))[:B]
    mk = lambda: cQ.NewPoly(B).set(host)
    ct0, ct1, out = (mk(), mk()), (mk(), mk()), (cQ.NewPoly(B), cQ.NewPoly(B))
    K = max(10, 200 // B)
    for it in range(K + 3):
        if it == 3:
            cQ.Sync()
            t0 = time.perf_counter()
        plan.MulRelin(level, ct0, ct1, key, out)
    cQ.Sync()
    dt = (time.perf_counter() - t0) / K
    print("%s B=%3d  %8.1f us per call  %7.1f us per product  %8.0f products/s" % (name, B, dt * 1e6, dt * 1e6 / B, B / dt), flush=True)
    del plan, key, ct0, ct1, out, cQ, cP

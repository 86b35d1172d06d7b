"""Launch pattern for a kernel trace: K x (CKKS pk-encrypt tail + decrypt of one ciphertext) at PN15QP880, batch B.   python tools/dbg/encrypt_b1.py [B] [K]"""
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
QP = list(Q) + list(P)
cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
plan = ring.CkksPlan(cQ, cP, B)
level = len(Q) - 1
mkqp = lambda s: ring.Poly(cQ, len(QP), B).set(sampling.uniform_poly(QP, N, B, seed=s))
u, pk0, pk1, e0, e1 = mkqp(1), mkqp(2), mkqp(3), mkqp(4), mkqp(5)
pt = cQ.NewPoly(B).set(sampling.uniform_poly(Q, N, B, seed=6))
sk = cQ.NewPoly(B).set(sampling.uniform_poly(Q, N, B, seed=7))
ct = (cQ.NewPoly(B), cQ.NewPoly(B))
out = cQ.NewPoly(B)
for name, fn in (("ENCRYPT", lambda: plan.EncryptPk(level, u, (pk0, pk1), (e0, e1), pt, ct)), ("DECRYPT", lambda: plan.Decrypt(level, ct, sk, out))):
    for it in range(K + 5):
        if it == 5:
            cQ.Sync()
            t0 = time.perf_counter()
        fn()
    cQ.Sync()
    print("%s us per call: %.1f" % (name, (time.perf_counter() - t0) / K * 1e6))

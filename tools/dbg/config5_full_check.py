"""One-off: every one of the 128 products of the config-5 leg's shape (PN16QP1761, level 33, chunks of 32) against the restatement -- the
leg itself checks two units per run.  Operands: four distinct ciphertext pairs tiled over the batch, so four oracle products suffice."""
import sys
import time
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg, oracle = g.load_package(), g.load_oracle()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
N, Q, P = params.ckks_moduli("PN16QP1761")
nq, np_ = len(Q), len(P)
level, beta = nq - 1, -(-nq // np_)
cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
chunk, total, distinct = 32, 128, 4
plan = ring.CkksPlan(cQ, cP, chunk)
evk_h = sampling.uniform_poly(Q + P, N, 2 * beta, seed=9)
evk = plan.NewSwitchingKey().set(evk_h)
ops = [sampling.uniform_poly(Q, N, 4, seed=700 + j).reshape(4, nq, N) for j in range(distinct)]
oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
t0 = time.time()
wants = [oplan.mulrelin(level, o[0:2], o[2:4], evk_h.reshape(beta, 2, nq + np_, N)) for o in ops]
print("oracle: %d products in %.1f s" % (distinct, time.time() - t0), flush=True)
bad = 0
for rep in range(3):
    for c0 in range(0, total, chunk):
        tile = lambda k: np.stack([ops[(c0 + u) % distinct][k] for u in range(chunk)])
        mk = lambda k: cQ.NewPoly(chunk).set(tile(k))
        out = (cQ.NewPoly(chunk), cQ.NewPoly(chunk))
        plan.MulRelin(level, (mk(0), mk(1)), (mk(2), mk(3)), evk, out)
        g0, g1 = out[0].get(), out[1].get()
        for u in range(chunk):
            w = wants[(c0 + u) % distinct]
            if not (np.array_equal(g0[u], w[0]) and np.array_equal(g1[u], w[1])):
                bad += 1
                print("MISMATCH rep", rep, "unit", c0 + u)
print("3 x %d products checked, mismatches: %d" % (total, bad))
sys.exit(1 if bad else 0)

"""Times lr_ckks_mulrelin / rescale on CKKS DefaultParams[PN15QP880] (synthetic operands)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
name = sys.argv[1] if len(sys.argv) > 1 else "PN15QP880"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
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
for _ in range(2): plan.MulRelin(level, ct0, ct1, evk, out)
cQ.Sync()
best = 1e9
for rep in range(3):
    cQ.TimerStart()
    for _ in range(3): plan.MulRelin(level, ct0, ct1, evk, out)
    best = min(best, cQ.TimerStop() / 3)
print(name, "N=%d |Q|=%d |P|=%d batch=%d: MulRelin %.3f ms/batch = %.1f mul/s; algorithmic %.1f GB/s" % (
    N, len(Q), len(P), B, best, B / (best * 1e-3), B * 8 * N * 360 / (best * 1e-3) / 1e9))
warm = (mk(), mk())
plan.Rescale(warm)            # first use at a level builds its constant table
cQ.Sync()
cQ.TimerStart()
plan.Rescale(out)
print("Rescale (both components): %.3f ms/batch" % cQ.TimerStop())

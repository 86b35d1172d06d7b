"""Times lr_bfv_mul on bfv DefaultParams (synthetic operands)."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
name = sys.argv[1] if len(sys.argv) > 1 else "PN14QP438"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N, Q, _, QMul = params.bfv_moduli(name)
cQ, cM = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, QMul)
plan = ring.BfvPlan(cQ, cM, 65537, B)
base = sampling.uniform_poly(Q, N, 2, seed=5)
host = np.concatenate([base] * (B // 2))
mk = lambda: cQ.NewPoly(B).set(host)
a, b, o = (mk(), mk()), (mk(), mk()), (cQ.NewPoly(B), cQ.NewPoly(B), cQ.NewPoly(B))
for _ in range(3): plan.Mul(a, b, o)
cQ.Sync()
cQ.TimerStart()
for _ in range(10): plan.Mul(a, b, o)
ms = cQ.TimerStop() / 10
print(name, "N=%d |Q|=%d |QMul|=%d batch=%d: Mul %.3f ms/batch = %.1f Mul/s" % (N, len(Q), len(QMul), B, ms, B / (ms * 1e-3)))

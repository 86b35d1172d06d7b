"""MulRelin + Rescale on every CKKS default parameter set (ckks/params.go:36-87) and BFV Mul on every BFV default set
(bfv/params.go:47-88): throughput table for DESIGN.md section 6.  Run on the GPU box."""
import json
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
rows = []
for name, B in (("PN12QP109", 1024), ("PN13QP218", 512), ("PN14QP438", 256), ("PN15QP880", 128), ("PN16QP1761", 32)):
    N, Q, P = params.ckks_moduli(name)
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, B)
    level = len(Q) - 1
    beta = -(-len(Q) // len(P))
    evk = plan.NewSwitchingKey().set(sampling.uniform_poly(Q + P, N, 2 * beta, seed=9))
    base = sampling.uniform_poly(Q, N, 2, seed=3)
    host = np.concatenate([base] * (B // 2))
    mk = lambda: cQ.NewPoly(B).set(host)
    ct0, ct1, out = (mk(), mk()), (mk(), mk()), (cQ.NewPoly(B), cQ.NewPoly(B))
    for _ in range(2):
        plan.MulRelin(level, ct0, ct1, evk, out)
    cQ.Sync()
    best = 1e9
    for rep in range(3):
        cQ.TimerStart()
        for _ in range(3):
            plan.MulRelin(level, ct0, ct1, evk, out)
        best = min(best, cQ.TimerStop() / 3)
    warm = (mk(), mk())
    plan.Rescale(warm)
    cQ.Sync()
    cQ.TimerStart()
    plan.Rescale(out)
    rs = cQ.TimerStop()
    r = {"params": name, "N": N, "limbs_Q": len(Q), "limbs_P": len(P), "beta": beta, "batch": B, "mulrelin_ms_per_batch": round(best, 4),
         "mulrelin_per_s": round(B / (best * 1e-3)), "rescale_ms_per_batch": round(rs, 4)}
    rows.append(r)
    print(json.dumps(r), flush=True)
    del plan, ct0, ct1, out, evk, warm, cQ, cP
json.dump({"device": "MI355X", "rows": rows}, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ckks_sets.json", "w"), indent=1)

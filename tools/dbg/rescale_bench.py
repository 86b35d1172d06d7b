"""DivRoundByLastModulusNTT on the reference's rings (60-bit limbs) and on the CKKS moduli: ms per 1 GiB batch, fraction of the roofline on
the algorithmic 8 N (2L - 1) bytes per poly, kernel of the transform.   python tools/dbg/rescale_bench.py [logn ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling, nat = pkg.ring, pkg.params, pkg.sampling, pkg._native
for logn in [int(a) for a in sys.argv[1:]] or [13, 14, 15, 16]:
    for name in ("qi60", "ckks"):
        N, Q = params.DefaultParamsQi(logn)
        Q = list(Q)
        if name == "ckks":
            if logn != 15:
                continue
            Q = list(params.ckks_moduli("PN15QP880")[1][:16])
        L = len(Q)
        B = (1 << 30) // (8 * N * L)
        ctx = ring.NewContextWithParams(N, Q)
        base = sampling.uniform_poly(Q, N, 2, seed=1)
        p = ctx.NewPoly(B).set(np.concatenate([base] * (B // 2)))

        def once():
            nat.check(nat.lib().lr_poly_set_limbs(p.h, L))
            ctx.DivRoundByLastModulusNTT(p)
        for _ in range(30):
            once()
        ctx.Sync()
        best = 1e9
        for _ in range(3):
            ctx.TimerStart()
            for _ in range(20):
                once()
            best = min(best, ctx.TimerStop() / 20)
        print("R%d %s: %.4f ms per %d polys, %.3f of the roofline, kernel %s" % (logn, name, best, B, 8 * N * (2 * L - 1) * B / (best * 1e-3) / 8e12, ctx.last_ntt_kernel()), flush=True)

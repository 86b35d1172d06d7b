"""NTT / InvNTT timing (HIP events) on the contexts of a BFV parameter set: Q, P and QMul as the evaluator uses them (in place, one operand of
`batch` polys per launch).      ntt_sets.py [PN14QP438] [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
name = sys.argv[1] if len(sys.argv) > 1 else "PN14QP438"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N, Q, P, M = params.bfv_moduli(name)
logn = N.bit_length() - 1
for label, moduli in (("Q", Q), ("QMul", M), ("Qi60", params.DefaultParamsQi(logn)[1][:len(Q)])):
    moduli = list(moduli)
    L = len(moduli)
    ctx = ring.NewContextWithParams(N, moduli)
    base = sampling.uniform_poly(moduli, N, 2, seed=1)
    bufs = [ctx.NewPoly(B).set(np.concatenate([base] * (B // 2))) for _ in range(6)]      # six operands in turn: above the Infinity Cache
    res = {}
    for op, fn in (("ntt", ctx.NTT), ("intt", ctx.InvNTT)):
        for b in bufs: fn(b, b)
        ctx.Sync()
        best = 1e9
        for rep in range(3):
            ctx.TimerStart()
            for _ in range(2):
                for b in bufs: fn(b, b)
            best = min(best, ctx.TimerStop() / 12)
        res[op] = "%.1f us  %.3f of 8 TB/s" % (best * 1e3, 16 * N * L * B / (best * 1e-3) / 8e12)
    print(label, [int(q).bit_length() for q in moduli], ctx.ntt_variants(), res, flush=True)

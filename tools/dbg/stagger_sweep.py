"""Sweep of the start-up stagger (LR_NTT_STAGGER, kilo-clocks per step) on the forward / inverse NTT at N = 2^15:
60-bit ring (integer bodies) and the CKKS moduli (FP64 bodies).  One process per setting (the switch is read at context creation)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package(); ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
N, q60 = params.DefaultParamsQi(15)
res = []
for name, moduli in (("q60", list(q60)), ("ckks", list(params.ckks_moduli("PN15QP880")[1][:16]))):
    ctx = ring.NewContextWithParams(N, moduli)
    B = 256
    base = sampling.uniform_poly(moduli, N, 2, seed=1)
    src, dst = ctx.NewPoly(B).set(np.concatenate([base] * (B // 2))), ctx.NewPoly(B)
    for fn, tag in ((ctx.NTT, "fwd"), (ctx.InvNTT, "inv")):
        for _ in range(30): fn(src, dst)
        ctx.Sync()
        best = 1e9
        for rep in range(3):
            ctx.TimerStart()
            for _ in range(50): fn(src, dst)
            best = min(best, ctx.TimerStop() / 50)
        res.append("%%s_%%s %%.4f ms (%%.3f)" %% (name, tag, best, 16 * N * 16 * B / (best * 1e-3) / 8e12))
print(" | ".join(res))
''' % ROOT
for st in sys.argv[1:] or ["0", "2", "4", "5", "6", "8"]:
    env = dict(os.environ, LR_NTT_STAGGER=st)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print("stagger", st, "->", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)

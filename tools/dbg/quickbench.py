"""Quick per-size NTT timing (HIP events), optional env LR_NTT_MODE."""
import sys, json
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
sizes = [int(a) for a in sys.argv[1:]] or [12, 13, 14, 15]
for logn in sizes:
    N, moduli = params.DefaultParamsQi(logn)
    bits = int(__import__('os').environ.get('QB_BITS', '0'))      # QB_BITS=40: moduli of that size instead (FP64 body below 2^46)
    if bits:
        moduli = params.GenerateNTTPrimes(bits, logn, len(moduli))
    if __import__('os').environ.get('QB_CKKS') and logn == 15:       # PN15QP880's first limbs: one of 51 bits, the rest 41
        moduli = list(params.ckks_moduli("PN15QP880")[1][:len(moduli)])
    L = len(moduli)
    B = (1 << 30) // (8 * N * L)      # 1 GiB per buffer
    ctx = ring.NewContextWithParams(N, moduli)
    nb = int(__import__('os').environ.get('QB_BASE', '2'))
    base = sampling.uniform_poly(moduli, N, nb, seed=1)
    host = np.concatenate([base] * (B // nb))
    src, dst = ctx.NewPoly(B).set(host), ctx.NewPoly(B)
    res = {}
    for name, fn in (("ntt", lambda: ctx.NTT(src, dst)), ("intt", lambda: ctx.InvNTT(src, dst))):
        for _ in range(3): fn()
        ctx.Sync()
        best = 1e9
        for rep in range(3):
            ctx.TimerStart()
            for _ in range(10): fn()
            best = min(best, ctx.TimerStop() / 10)
        gbs = 16 * N * L * B / (best * 1e-3) / 1e9
        res[name] = (round(best, 4), round(gbs), round(gbs / 8000, 4), "%.3g bfly/s" % (B * L * (N // 2) * logn / (best * 1e-3)))
    print(logn, L, B, res, flush=True)

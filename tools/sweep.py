"""Throughput sweep over the reference's benchmark rings R12..R16 (ring/params.go:10-25): NTT, InvNTT,
MulCoeffsMontgomery and ModUpSplitQP as absolute numbers and as fraction of the 8 TB/s HBM roofline
(algorithmic bytes of SURVEY.md 8(d)).  Writes one JSON document; run on the GPU box."""
import json
import sys

sys.path.insert(0, '/root/repo')
import numpy as np

import __graft_entry__ as g

pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
PEAK = 8000.0


def timeit(ctx, fn, reps=10):
    for _ in range(2):
        fn()
    ctx.Sync()
    best = 1e9
    for _ in range(3):
        ctx.TimerStart()
        for _ in range(reps):
            fn()
        best = min(best, ctx.TimerStop() / reps)
    return best


rows = []
for logn in (12, 13, 14, 15, 16):
    N, Q = params.DefaultParamsQi(logn)
    _, P = params.DefaultParamsPi(logn)
    L = len(Q)
    B = max(2, (1 << 30) // (8 * N * L))          # 1 GiB per buffer (R15: 256 polys, the batch bench.py uses)
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    base = sampling.uniform_poly(Q, N, 2, seed=logn)
    host = np.concatenate([base] * (B // 2))
    a, b, c = cQ.NewPoly(B).set(host), cQ.NewPoly(B).set(host), cQ.NewPoly(B)
    pp = cP.NewPoly(B)
    be = ring.NewFastBasisExtender(cQ, cP)
    r = {"logN": logn, "limbs": L, "batch": B}
    for name, fn, bytes_ in (
        ("ntt", lambda: cQ.NTT(a, c), 16 * N * L * B),
        ("intt", lambda: cQ.InvNTT(a, c), 16 * N * L * B),
        ("mulcoeffs_montgomery", lambda: cQ.MulCoeffsMontgomery(a, b, c), 24 * N * L * B),
        ("modup_split_qp", lambda: be.ModUpSplitQP(L - 1, a, pp), 8 * N * (L + L) * B),
    ):
        ms = timeit(cQ, fn, reps=10 if name != "modup_split_qp" else 3)
        gbs = bytes_ / (ms * 1e-3) / 1e9
        r[name] = {"ms": round(ms, 4), "poly_per_s": round(B / (ms * 1e-3)), "GBs": round(gbs), "frac_hbm": round(gbs / PEAK, 4)}
        if name in ("ntt", "intt"):
            r[name]["limb_ntt_per_s"] = round(B * L / (ms * 1e-3))
    rows.append(r)
    print(json.dumps(r), flush=True)
    del a, b, c, pp, be, cQ, cP
# the same transforms on the CKKS default moduli (ckks/params.go:36-87: 30..55-bit limbs): limbs below 2^46 run on the FP64
# bodies of the dual kernels
ckks_rows = []
for name in ("PN12QP109", "PN13QP218", "PN14QP438", "PN15QP880", "PN16QP1761"):
    N, Q, _ = params.ckks_moduli(name)
    L = len(Q)
    B = max(2, (1 << 30) // (8 * N * L)) & ~1
    cQ = ring.NewContextWithParams(N, Q)
    base = sampling.uniform_poly(Q, N, 2, seed=L)
    a, c = cQ.NewPoly(B).set(np.concatenate([base] * (B // 2))), cQ.NewPoly(B)
    r = {"params": name, "logN": N.bit_length() - 1, "limbs": L, "batch": B, "moduli_bits": [int(q).bit_length() for q in Q],
         "asm_variants": list(cQ.ntt_variants())}
    for op, fn in (("ntt", lambda: cQ.NTT(a, c)), ("intt", lambda: cQ.InvNTT(a, c))):
        ms = timeit(cQ, fn)
        gbs = 16 * N * L * B / (ms * 1e-3) / 1e9
        r[op] = {"ms": round(ms, 4), "limb_ntt_per_s": round(B * L / (ms * 1e-3)), "GBs": round(gbs), "frac_hbm": round(gbs / PEAK, 4)}
    ckks_rows.append(r)
    print(json.dumps(r), flush=True)
    del a, c, cQ
json.dump({"device": "MI355X", "peak_GBs": PEAK, "rows": rows, "ckks_moduli_rows": ckks_rows}, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sweep.json", "w"), indent=1)

"""Stress of the pair-flag inverse kernels at N = 2^16 (gen_intt.py: fused_last): 25 launches of 6144 / 9216 workgroups each, every
output word compared with the two-pass path (lazy sub-blocks + ntt_top_kernel, LR_NO_INVFUSE=1).  Run on the GPU box."""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
N = 1 << 16
for kind in ("qi60", "ckks"):
    Q = list(params.Qi60()[-4:]) if kind == "qi60" else list(params.ckks_moduli("PN16QP1761")[1][:6])
    L = len(Q)
    B = 768
    x = sampling.random_u64((4, L, N), seed=5)
    for i, q in enumerate(Q):
        x[:, i] %= np.uint64(4 * q)
    tiled = np.concatenate([x] * (B // 4))
    os.environ["LR_NO_INVFUSE"] = "1"
    ref_ctx = ring.NewContextWithParams(N, Q)
    p, r = ref_ctx.NewPoly(B).set(tiled), ref_ctx.NewPoly(B)
    ref_ctx.InvNTT(p, r)
    want = r.get()
    k_ref = ref_ctx.last_ntt_kernel()
    del os.environ["LR_NO_INVFUSE"]
    ctx = ring.NewContextWithParams(N, Q)
    p2, r2 = ctx.NewPoly(B).set(tiled), ctx.NewPoly(B)
    bad = 0
    for rep in range(25):
        ctx.InvNTT(p2, r2)
        got = r2.get()
        if not np.array_equal(got, want):
            bad += 1
            print(kind, "MISMATCH at rep", rep, int((got != want).sum()))
    print(kind, k_ref, ctx.last_ntt_kernel(), "launches 25 x", B * L * 2, "workgroups, mismatching launches:", bad)

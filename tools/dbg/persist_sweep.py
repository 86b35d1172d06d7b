"""Sweep of the polys-per-workgroup count of the persistent forward 2^15 kernels (LR_NTT_PERSIST; 0 = one-poly workgroups) on the
bench shape (256 polys x 16 limbs: 60-bit ring = integer body, CKKS moduli = FP64 body), with the inverse transforms and the PN15QP880
MulRelin beside them.  One process per setting (the switch is read at context creation); every output is compared with the setting-0
output of the same process tree through a checksum."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, zlib, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package(); ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
N, q60 = params.DefaultParamsQi(15)
res = []
for name, moduli in (("q60", list(q60)), ("ckks", list(params.ckks_moduli("PN15QP880")[1][:16])), ("fp40", params.GenerateNTTPrimes(40, 15, 16))):
    ctx = ring.NewContextWithParams(N, moduli)
    B = 256
    base = sampling.uniform_poly(moduli, N, 7, seed=1)
    src, dst = ctx.NewPoly(B).set(np.concatenate([base] * 37)[:B]), ctx.NewPoly(B)
    for fn, tag in ((ctx.NTT, "fwd"), (ctx.InvNTT, "inv")):
        for _ in range(30): fn(src, dst)
        ctx.Sync()
        best = 1e9
        for rep in range(3):
            ctx.TimerStart()
            for _ in range(50): fn(src, dst)
            best = min(best, ctx.TimerStop() / 50)
        crc = zlib.crc32(dst.get().tobytes())
        res.append("%%s_%%s %%.4f ms (%%.3f) %%s crc %%08x" %% (name, tag, best, 16 * N * 16 * B / (best * 1e-3) / 8e12, ctx.last_ntt_kernel(), crc))
    del src, dst, ctx
mN, mQ, mP = params.ckks_moduli("PN15QP880")
cQ, cP = ring.NewContextWithParams(mN, mQ), ring.NewContextWithParams(mN, mP)
mb = 128
plan = ring.CkksPlan(cQ, cP, mb)
beta = -(-len(mQ) // len(mP))
key = plan.NewSwitchingKey().set(sampling.uniform_poly(mQ + mP, mN, 2 * beta, seed=9))
ops = [sampling.uniform_poly(mQ, mN, 2, seed=40 + k) for k in range(4)]
tile = lambda x: np.concatenate([x] * (mb // 2))
c0 = (cQ.NewPoly(mb).set(tile(ops[0])), cQ.NewPoly(mb).set(tile(ops[1])))
c1 = (cQ.NewPoly(mb).set(tile(ops[2])), cQ.NewPoly(mb).set(tile(ops[3])))
co = (cQ.NewPoly(mb), cQ.NewPoly(mb))
for _ in range(5): plan.MulRelin(len(mQ) - 1, c0, c1, key, co)
cQ.Sync()
best = 1e9
for rep in range(3):
    cQ.TimerStart()
    for _ in range(5): plan.MulRelin(len(mQ) - 1, c0, c1, key, co)
    best = min(best, cQ.TimerStop() / 5)
crc = zlib.crc32(co[0].get().tobytes()) ^ zlib.crc32(co[1].get().tobytes())
res.append("mulrelin15 %%.3f ms = %%.1f k/s crc %%08x" %% (best, mb / best, crc))
print(" | ".join(res))
''' % ROOT
for st in sys.argv[1:] or ["0", "2", "4", "8", "16"]:
    env = dict(os.environ, LR_NTT_PERSIST=st)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print("persist", st, "->", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-800:], flush=True)

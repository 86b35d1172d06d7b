"""Is the NTT launch loop power-limited?  Samples rocm-smi (power, shader clock) while a kernel loop runs.
    python tools/dbg/power_probe.py [fp|int|mul|modup|mulrelin|idle]"""
import subprocess
import sys
import threading
import time
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
kind = sys.argv[1] if len(sys.argv) > 1 else "int"
N, Q = params.DefaultParamsQi(15)
if kind == "fp":
    Q = list(params.ckks_moduli("PN15QP880")[1][1:17])
ctx = ring.NewContextWithParams(N, Q)
B = 256
base = sampling.uniform_poly(Q, N, 2, seed=1)
src, dst = ctx.NewPoly(B).set(np.concatenate([base] * (B // 2))), ctx.NewPoly(B)
step = lambda: ctx.NTT(src, dst)
if kind == "mul":
    step = lambda: ctx.MulCoeffsMontgomery(src, src, dst)
elif kind == "modup":
    ctxP = ring.NewContextWithParams(N, params.DefaultParamsPi(15)[1])
    be, outP = ring.NewFastBasisExtender(ctx, ctxP), ctxP.NewPoly(B)
    step = lambda: be.ModUpSplitQP(len(Q) - 1, src, outP)
elif kind == "mulrelin":
    _, Qc, Pc = params.ckks_moduli("PN15QP880")
    cQ, cP = ring.NewContextWithParams(N, Qc), ring.NewContextWithParams(N, Pc)
    plan = ring.CkksPlan(cQ, cP, 128)
    beta = -(-len(Qc) // len(Pc))
    evk = plan.NewSwitchingKey().set(sampling.uniform_poly(Qc + Pc, N, 2 * beta, seed=9))
    hb = np.concatenate([sampling.uniform_poly(Qc, N, 2, seed=3)] * 64)
    mk = lambda: cQ.NewPoly(128).set(hb)
    ct0, ct1, out = (mk(), mk()), (mk(), mk()), (cQ.NewPoly(128), cQ.NewPoly(128))
    ctx = cQ
    step = lambda: plan.MulRelin(len(Qc) - 1, ct0, ct1, evk, out)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.time(), out.strip()[:1500]))
        except Exception as ex:
            samples.append((time.time(), "ERR %s" % ex))
        time.sleep(0.3)
t = threading.Thread(target=sampler)
t.start()
t0 = time.time()
n = 0
while time.time() - t0 < 6.0:
    if kind != "idle":
        for _ in range(20):
            step()
        ctx.Sync()
        n += 20
    else:
        time.sleep(0.1)
el = time.time() - t0
stop = True
t.join()
print(kind, ctx.last_ntt_kernel() if kind in ("int", "fp") else "", "launches", n, "ms per launch", el / max(n, 1) * 1e3)
for ts, s in samples[::3]:
    print(round(ts - t0, 1), s[:700])

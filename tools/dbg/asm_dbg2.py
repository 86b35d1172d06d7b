import sys, os
import numpy as np
from isa import Machine, Program, s, v
import gen_ntt as G
q = 1152921504100384769
gen = G.Gen(14)
m = Machine(1024, 1024, 1024)
rng = np.random.default_rng(1)
def setv(reg, vals):
    m.vgpr[reg.idx] = (vals & 0xFFFFFFFF).astype(np.uint32); m.vgpr[reg.idx+1] = (vals >> 32).astype(np.uint32); m.vdef[reg.idx:reg.idx+2] = True
def getv(reg):
    return m.vgpr[reg.idx].astype(np.uint64) | (m.vgpr[reg.idx+1].astype(np.uint64) << np.uint64(32))
def sets(reg, val):
    m.sgpr[reg.idx] = val & 0xFFFFFFFF; m.sdef[reg.idx] = True
    if reg.n == 2: m.sgpr[reg.idx+1] = (val >> 32) & 0xFFFFFFFF; m.sdef[reg.idx+1] = True
sets(gen.Qm, q); sets(gen.NQ, (1<<64) - q); sets(gen.Q4, 4*q); sets(gen.NQ8, (1<<64) - 8*q)
qh = (q >> 32) + 1; g = qh.bit_length() - 1
sets(gen.REDM, min((1 << (32 + g)) // qh, 0xFFFFFFFF)); sets(gen.REDG, g)
m.vgpr[gen.Z1.idx] = 0; m.vgpr[gen.Z3.idx] = 0; m.vdef[gen.Z1.idx] = m.vdef[gen.Z3.idx] = True
w = 123456789123456789 % q; ws = (w << 64) // q
tw = (s(36), s(37), s(38), s(39))
sets(tw[0], w & 0xFFFFFFFF); sets(tw[1], w >> 32); sets(tw[2], ws & 0xFFFFFFFF); sets(tw[3], ws >> 32)
U = rng.integers(0, 1 << 63, 1024, dtype=np.uint64) % np.uint64(16*q - 1)
V = rng.integers(0, 1 << 63, 1024, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
for correct in (False, True):
    setv(gen.X[0], U); setv(gen.X[1], V)
    gen.p = Program()
    gen.butterfly(gen.X[0], gen.X[1], tw, correct)
    m.run(gen.p)
    X, Y = getv(gen.X[0]), getv(gen.X[1])
    ok = True
    for i in range(1024):
        u, vv = int(U[i]), int(V[i])
        r = vv * w % q
        if int(X[i]) % q != (u + r) % q or int(Y[i]) % q != (u - r) % q: ok = False; print("bad", i, correct); break
        if correct and u >= 8*q and int(X[i]) > 12*q: print("range", i)
    print("butterfly correct=%s ok=%s maxX/q=%.2f" % (correct, ok, max(int(a) for a in X)/q))
xs = rng.integers(0, 1 << 63, 1024, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
setv(gen.X[2], xs); gen.p = Program(); gen.reduce_2q(gen.X[2]); m.run(gen.p)
r = getv(gen.X[2]); print("reduce_2q ok", all(int(r[i]) % q == int(xs[i]) % q and int(r[i]) < 2*q for i in range(1024)))
xs = rng.integers(0, 1 << 63, 1024, dtype=np.uint64) % np.uint64(16*q-1)
setv(gen.X[2], xs); gen.p = Program(); gen.canon(gen.X[2]); m.run(gen.p)
r = getv(gen.X[2]); print("canon ok", all(int(r[i]) == int(xs[i]) % q for i in range(1024)))
# trace reduce_2q for lane 0
xs = np.full(1024, 0xFEDCBA9876543211, dtype=np.uint64)
setv(gen.X[2], xs); gen.p = Program(); gen.reduce_2q(gen.X[2])
for ins in gen.p.ins:
    p1 = Program(); p1.ins=[ins]; m.run(p1)
    print(ins[0], [repr(a) for a in ins[1]], "T0=%x T2=%x R=%x X=%x" % (m.vgpr[gen.T0.idx][0], m.vgpr[gen.T2.idx][0], int(getv(gen.R)[0]), int(getv(gen.X[2])[0])))
x0 = 0xFEDCBA9876543211
k = ((x0 >> 32) * int(m.sgpr[gen.REDM.idx][0]) >> 32) >> int(m.sgpr[gen.REDG.idx][0])
print("expect k", k, "x-kq", hex(x0 - k*q), x0//q)

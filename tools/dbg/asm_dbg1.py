import sys, os
import numpy as np
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "lattigo-fhe-by-go_amd", "csrc", "asmgen"))
import __graft_entry__ as graft
from isa import Machine, Program
import gen_ntt as G

logn = int(sys.argv[1])
oracle = graft.load_oracle(); pkg = graft.load_package()
N = 1 << logn
q = pkg.params.Qi60()[-3]
oc = oracle.Context(N, [q])
x = pkg.sampling.random_u64((N,), seed=5)
x[:4] = np.uint64(0xFFFFFFFFFFFFFFFF)
psi = [int(oracle.inv_mform(int(w), q)) for w in oc.ntt_psi[0]]

def partial_ntt(vals, nstages):
    a = [int(v) % q for v in vals]
    t = N
    m = 1
    for s in range(nstages):
        t >>= 1
        for i in range(m):
            w = psi[m + i]
            j1 = 2 * i * t
            for j in range(j1, j1 + t):
                u, v_ = a[j], a[j + t] * w % q
                a[j], a[j + t] = (u + v_) % q, (u - v_) % q
        m <<= 1
    return a

# rebuild machine like selftest
tw = np.zeros((N, 2), dtype=np.uint64)
for i, w in enumerate(psi):
    tw[i, 0] = w; tw[i, 1] = (w << 64) // q
blocks = N // 16
twf = np.zeros((15, blocks, 2), dtype=np.uint64)
for cc in range(4):
    for j in range(1 << cc):
        for bk in range(blocks):
            twf[(1 << cc) - 1 + j, bk] = tw[((blocks + bk) << cc) + j]
qh = (q >> 32) + 1; g = qh.bit_length() - 1
red_m = min((1 << (32 + g)) // qh, 0xFFFFFFFF)
lp = np.zeros(8, dtype=np.uint64); lp[0] = q; lp[7] = red_m | (g << 32)
A_IN, A_OUT, A_LP, A_TW, A_TWF, A_KARG = 0x1000, 0x1000 + 8 * N, 0x200000, 0x300000, 0x300000 + 16 * N + 0x1000, 0x800
mem = np.zeros((A_TWF + 16 * 15 * blocks + 0x1000) // 4, dtype=np.uint32)
def place(arr, addr):
    words = np.ascontiguousarray(arr).view(np.uint32).ravel(); mem[addr // 4: addr // 4 + words.size] = words
place(x, A_IN); place(lp, A_LP); place(tw, A_TW); place(twf, A_TWF)
karg = np.zeros(11, dtype=np.uint64)
karg[0], karg[1], karg[2], karg[3] = A_IN, A_OUT, N, N
karg[4] = karg[5] = karg[6] = (1 << 32); karg[7] = 1 | (1 << 32)
karg[8], karg[9], karg[10] = A_LP, A_TW, A_TWF
place(karg, A_KARG)
gen = G.Gen(logn)
gen.prologue(); gen.pass_a()
m = Machine(1024, 160 * 1024, mem.size); m.mem = mem
m.vgpr[0] = np.arange(1024, dtype=np.uint32); m.vdef[0] = True
m.sgpr[0], m.sgpr[1] = A_KARG, 0; m.sdef[0:4] = True
m.run(gen.p)
print("q", hex(q), "s14/15", hex(int(m.sgpr[14][0]) | int(m.sgpr[15][0]) << 32), "redm", hex(int(m.sgpr[22][0])), int(m.sgpr[23][0]))
for ns in range(0, gen.A + 1):
    ref = partial_ntt(x, ns)
    bad = 0
    for k in range(gen.RA):
        val = m.vgpr[2 * k].astype(np.uint64) | (m.vgpr[2 * k + 1].astype(np.uint64) << np.uint64(32))
        want = np.array([ref[k * 1024 + t] for t in range(1024)], dtype=object)
        got = np.array([int(vv) % q for vv in val], dtype=object)
        bad += int((got != want).sum())
    print("after", ns, "stages: mismatches", bad)

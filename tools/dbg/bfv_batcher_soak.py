"""Soak of the BFV batcher on the GPU box: T host threads, each R rounds of Mul then Relinearize of one ciphertext pair through
lr_bfv_batcher_*, the operands drawn from K fixed pairs whose results were computed beforehand by the plain batched entry points
(lr_bfv_mul / lr_bfv_relinearize, themselves parity-tested against the oracle).  Every returned poly is compared bit for bit.
    bfv_batcher_soak.py [PN14QP438] [threads] [rounds] [lanes] [max_batch]"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
name = sys.argv[1] if len(sys.argv) > 1 else "PN14QP438"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
R = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 2
max_batch = int(sys.argv[5]) if len(sys.argv) > 5 else 32
K = 8
N, Q, P, QM = params.bfv_moduli(name)
Q, P, QM = list(Q), list(P), list(QM)
nq, np_ = len(Q), len(P)
beta = -(-nq // np_)
t = 65537
cq, cp, cm = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P), ring.NewContextWithParams(N, QM)
mul, ks = ring.BfvPlan(cq, cm, t, K), ring.CkksPlan(cq, cp, K)
evk = sampling.uniform_poly(Q + P, N, 2 * beta, seed=17)
ops = [sampling.uniform_poly(Q, N, K, seed=100 + k) for k in range(4)]
P_ = lambda x: cq.NewPoly(K).set(x)
deg2 = (cq.NewPoly(K), cq.NewPoly(K), cq.NewPoly(K))
mul.Mul((P_(ops[0]), P_(ops[1])), (P_(ops[2]), P_(ops[3])), deg2)
lin = (cq.NewPoly(K), cq.NewPoly(K))
ks.BfvRelinearize(deg2, ks.NewSwitchingKey().set(evk), lin)
want2 = [p.get().reshape(K, nq, N) for p in deg2]
want1 = [p.get().reshape(K, nq, N) for p in lin]

bat = ring.BfvBatcher(N, Q, P, QM, t, max_batch=max_batch, lanes=lanes)
key = bat.NewSwitchingKey().set(evk)
errors, done = [], [0] * T


def evaluator(i):
    try:
        rng = np.random.default_rng(i)
        c = ring.NewContextWithParams(N, Q)
        ins = [[c.NewPoly(1).set(ops[k][j:j + 1]) for k in range(4)] for j in range(K)]
        d2 = (c.NewPoly(1), c.NewPoly(1), c.NewPoly(1))
        l1 = (c.NewPoly(1), c.NewPoly(1))
        for r in range(R):
            j = int(rng.integers(0, K))
            a0, a1, b0, b1 = ins[j]
            bat.Mul((a0, a1), (b0, b1), d2)
            bat.Relinearize(d2, key, l1)
            if r % 4 == 0 or r == R - 1:          # (the downloads dominate the loop otherwise)
                for k in range(3):
                    if not np.array_equal(d2[k].get().reshape(nq, N), want2[k][j]):
                        raise AssertionError("Mul: thread %d round %d pair %d component %d" % (i, r, j, k))
                for k in range(2):
                    if not np.array_equal(l1[k].get().reshape(nq, N), want1[k][j]):
                        raise AssertionError("Relinearize: thread %d round %d pair %d component %d" % (i, r, j, k))
            done[i] += 1
    except Exception as e:  # noqa: BLE001
        errors.append(repr(e))


t0 = time.time()
ths = [threading.Thread(target=evaluator, args=(i,)) for i in range(T)]
for th in ths:
    th.start()
last = t0
while any(th.is_alive() for th in ths):
    time.sleep(0.5)
    if time.time() - last < 20:
        continue
    last = time.time()
    print("[%.0f s] rounds done: %d of %d, errors %d" % (time.time() - t0, sum(done), T * R, len(errors)), flush=True)
for th in ths:
    th.join()
elapsed = time.time() - t0
st = bat.Stats()
print("%s: %d threads x %d rounds (Mul + Relinearize), lanes %d, max_batch %d: %.1f s, %.0f pairs/s; batcher %s; errors: %s"
      % (name, T, R, lanes, max_batch, elapsed, sum(done) / elapsed, st, errors[:3] or "none"))
sys.exit(1 if errors or sum(done) != T * R else 0)

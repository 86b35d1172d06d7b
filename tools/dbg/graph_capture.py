"""Can the MulRelin pipeline be captured into a HIP graph (torch.cuda.CUDAGraph on the stream the contexts are set to) and replayed?
Small batches are launch-bound (about 20 launches per product): eager vs replay, outputs compared with the oracle."""
import sys
import time
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
import __graft_entry__ as g
pkg, oracle = g.load_package(), g.load_oracle()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
name = sys.argv[1] if len(sys.argv) > 1 else "PN13QP218"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N, Q, P = params.ckks_moduli(name)
nq, np_ = len(Q), len(P)
level, beta = nq - 1, -(-nq // np_)
cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
plan = ring.CkksPlan(cQ, cP, B)
evk_h = sampling.uniform_poly(Q + P, N, 2 * beta, seed=9)
evk = plan.NewSwitchingKey().set(evk_h)
ops = [sampling.uniform_poly(Q, N, B, seed=50 + k).reshape(B, nq, N) for k in range(4)]
mk = lambda k: cQ.NewPoly(B).set(ops[k])
ct0, ct1, out = (mk(0), mk(1)), (mk(2), mk(3)), (cQ.NewPoly(B), cQ.NewPoly(B))
side = torch.cuda.Stream()
def set_stream(s):
    cQ.SetStream(s.cuda_stream)
    cP.SetStream(s.cuda_stream)
with torch.cuda.stream(side):
    set_stream(side)
    for _ in range(3):
        plan.MulRelin(level, ct0, ct1, evk, out)          # warm-up: every pool reaches its size, no allocation afterwards
    side.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        plan.MulRelin(level, ct0, ct1, evk, out)
    side.synchronize()
    eager = (time.perf_counter() - t0) / 200
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        plan.MulRelin(level, ct0, ct1, evk, out)
    out[0].set(np.zeros_like(ops[0]))
    graph.replay()
    side.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        graph.replay()
    side.synchronize()
    replay = (time.perf_counter() - t0) / 200
oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
want = oplan.mulrelin(level, np.stack([ops[0][0], ops[1][0]]), np.stack([ops[2][0], ops[3][0]]), evk_h.reshape(beta, 2, nq + np_, N))
ok = np.array_equal(out[0].get().reshape(B, nq, N)[0], want[0]) and np.array_equal(out[1].get().reshape(B, nq, N)[0], want[1])
print("%s batch %d: eager %.1f us per MulRelin call, graph replay %.1f us, replayed output bit-exact: %s" % (name, B, eager * 1e6, replay * 1e6, ok))

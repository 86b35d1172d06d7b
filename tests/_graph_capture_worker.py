"""Worker of test_gpu_graph_capture.py (own process: torch brings its own HIP runtime and has to initialise it before the library's)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    torch.cuda.init()
    pkg, oracle = graft.load_package(), graft.load_oracle()
    ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
    N, Qf, Pf = params.ckks_moduli("PN16QP1761")        # N = 2^16: sub-block kernels, pair flags (a memset node), extension with the top stage
    Q, P = list(Qf[:6]), list(Pf[:2])
    nq, np_ = len(Q), len(P)
    level, beta, B = nq - 1, -(-nq // np_), 2
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, B)
    evk_h = sampling.uniform_poly(Q + P, N, 2 * beta, seed=9)
    evk = plan.NewSwitchingKey().set(evk_h)
    ops = [sampling.uniform_poly(Q, N, B, seed=60 + k).reshape(B, nq, N) for k in range(4)]
    mk = lambda k: cQ.NewPoly(B).set(ops[k])
    ct0, ct1, out = (mk(0), mk(1)), (mk(2), mk(3)), (cQ.NewPoly(B), cQ.NewPoly(B))
    oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    wants = [oplan.mulrelin(level, np.stack([ops[0][b], ops[1][b]]), np.stack([ops[2][b], ops[3][b]]), evk_h.reshape(beta, 2, nq + np_, N))
             for b in range(B)]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        cQ.SetStream(side.cuda_stream)
        cP.SetStream(side.cuda_stream)
        plan.MulRelin(level, ct0, ct1, evk, out)          # warm-up outside the capture: pools at their size, nothing allocates afterwards
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            plan.MulRelin(level, ct0, ct1, evk, out)
        for rep in range(2):
            out[0].set(np.zeros_like(ops[0]))
            out[1].set(np.zeros_like(ops[0]))
            side.synchronize()
            graph.replay()
            side.synchronize()
            for b in range(B):
                assert np.array_equal(out[0].get().reshape(B, nq, N)[b], wants[b][0]), (rep, b, 0)
                assert np.array_equal(out[1].get().reshape(B, nq, N)[b], wants[b][1]), (rep, b, 1)
    print("graph replay ok")


if __name__ == "__main__":
    main()

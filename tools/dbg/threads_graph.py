"""Serving shape of MulRelin: T host threads, each with its own contexts / plan / stream (the reference's evaluator-per-goroutine model),
batch b per call, either calling the library directly or replaying one HIP graph of the call.  Prints products per second per (T, b, how).
usage: threads_graph.py [PN15QP880] [T,T,...] [b,b,...]"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "PN15QP880"
    Ts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,4,16").split(",")]
    Bs = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1").split(",")]
    torch.cuda.init()
    pkg, oracle = graft.load_package(), graft.load_oracle()
    ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
    N, Q, P = params.ckks_moduli(name)
    nq, np_ = len(Q), len(P)
    level, beta = nq - 1, -(-nq // np_)
    key_h = sampling.uniform_poly(Q + P, N, 2 * beta, seed=9)
    ops1 = [sampling.uniform_poly(Q, N, 1, seed=60 + k).reshape(1, nq, N) for k in range(4)]
    want = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P)).mulrelin(
        level, np.stack([ops1[0][0], ops1[1][0]]), np.stack([ops1[2][0], ops1[3][0]]), key_h.reshape(beta, 2, nq + np_, N))
    for b in Bs:
        ops = [np.concatenate([o] * b) for o in ops1]
        for T in Ts:
            workers = []
            for i in range(T):
                st = torch.cuda.Stream()
                cq, cp = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
                cq.SetStream(st.cuda_stream)
                cp.SetStream(st.cuda_stream)
                plan = ring.CkksPlan(cq, cp, b)
                key = plan.NewSwitchingKey().set(key_h)
                c0 = (cq.NewPoly(b).set(ops[0]), cq.NewPoly(b).set(ops[1]))
                c1 = (cq.NewPoly(b).set(ops[2]), cq.NewPoly(b).set(ops[3]))
                out = (cq.NewPoly(b), cq.NewPoly(b))
                with torch.cuda.stream(st):
                    plan.MulRelin(level, c0, c1, key, out)
                    st.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=st):
                        plan.MulRelin(level, c0, c1, key, out)
                    st.synchronize()
                workers.append((st, cq, cp, plan, key, c0, c1, out, g))
            bat = ring.CkksBatcher(N, Q, P, max_batch=max(64, T * b), lanes=int(os.environ.get("LANES", "2")))
            bkey = bat.NewSwitchingKey().set(key_h)
            for w in workers[:1]:
                bat.MulRelin(level, w[5], w[6], bkey, w[7])      # warm-up: pools at their size
            for how in ("direct", "graph", "batcher"):
                iters = max(20, 200 // b)
                gate = threading.Barrier(T + 1)

                def loop(w):
                    st, cq, _, plan, key, c0, c1, out, g = w
                    with torch.cuda.stream(st):
                        gate.wait()
                        for _ in range(iters):
                            if how == "direct":
                                plan.MulRelin(level, c0, c1, key, out)
                            elif how == "graph":
                                g.replay()
                            else:
                                bat.MulRelin(level, c0, c1, bkey, out)
                        st.synchronize()
                for w in workers:
                    w[7][0].set(np.zeros_like(ops[0]))
                    w[7][1].set(np.zeros_like(ops[0]))
                    w[1].Sync()
                ths = [threading.Thread(target=loop, args=(w,)) for w in workers]
                for th in ths:
                    th.start()
                gate.wait()
                t0 = time.perf_counter()
                for th in ths:
                    th.join()
                dt = time.perf_counter() - t0
                ok = all(np.array_equal(w[7][k].get().reshape(b, nq, N)[b - 1], want[k]) for w in workers for k in range(2))
                print("%s T=%d b=%d %s: %.0f products/s (%.1f us per call per thread) bit_exact=%s" %
                      (name, T, b, how, T * b * iters / dt, dt / iters * 1e6, ok), flush=True)
            print("   batcher:", bat.Stats(), flush=True)
            del bat
            del workers


if __name__ == "__main__":
    main()

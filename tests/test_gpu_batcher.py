"""The MulRelin batcher (lr_ckks_batcher_*): concurrent batch-1 calls from many host threads -- the reference's one evaluator per
goroutine (examples/dbfv/psi/psi.go:215-233) -- are merged into batched launches.  Every caller must get the product of ITS OWN
operands, bit-identical to the oracle's MulRelin, whatever batch its request ended up in."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(gpu_pkg, oracle, logn, nq, np_, max_batch, lanes):
    N = 1 << logn
    _, Qf, Pf = gpu_pkg.params.ckks_moduli("PN16QP1761" if logn == 16 or np_ > 3 else "PN15QP880")
    Q, P = list(Qf[:nq]), list(Pf[:np_])
    ring = gpu_pkg.ring
    bat = ring.CkksBatcher(N, Q, P, max_batch=max_batch, lanes=lanes)
    beta = -(-nq // np_)
    evk = gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=99)
    key = bat.NewSwitchingKey().set(evk)
    oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    return N, Q, P, bat, key, oplan, evk.reshape(beta, 2, nq + np_, N)


@pytest.mark.parametrize("logn,nq,np_,level,threads,lanes,max_batch", [(12, 6, 2, 5, 8, 2, 4), (12, 6, 2, 4, 6, 1, 8), (14, 7, 3, 6, 8, 2, 16),
                                                                       (16, 6, 2, 5, 4, 2, 4)])
def test_concurrent_callers_get_their_own_products(gpu_pkg, oracle, logn, nq, np_, level, threads, lanes, max_batch):
    N, Q, P, bat, key, oplan, evk = _setup(gpu_pkg, oracle, logn, nq, np_, max_batch, lanes)
    ring = gpu_pkg.ring
    rounds = 3
    errors, results = [], {}

    def evaluator(t):
        try:
            cq = ring.NewContextWithParams(N, Q)                      # the caller's own context, like every goroutine's evaluator
            for r in range(rounds):
                seed = 1000 * t + 10 * r
                ops = [gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 1, seed=seed + k) for k in range(4)]
                c0 = (cq.NewPolyLvl(level, 1).set(ops[0]), cq.NewPolyLvl(level, 1).set(ops[1]))
                c1 = (cq.NewPolyLvl(level, 1).set(ops[2]), cq.NewPolyLvl(level, 1).set(ops[3]))
                out = (cq.NewPolyLvl(level, 1), cq.NewPolyLvl(level, 1))
                bat.MulRelin(level, c0, c1, key, out)
                results[(t, r)] = (ops, out[0].get().reshape(level + 1, N), out[1].get().reshape(level + 1, N))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=evaluator, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    assert len(results) == threads * rounds
    for (t, r), (ops, g0, g1) in results.items():
        w0, w1 = oplan.mulrelin(level, np.stack([ops[0].reshape(level + 1, N), ops[1].reshape(level + 1, N)]),
                                np.stack([ops[2].reshape(level + 1, N), ops[3].reshape(level + 1, N)]), evk)
        assert np.array_equal(g0, w0), (t, r)
        assert np.array_equal(g1, w1), (t, r)
    st = bat.Stats()
    assert st["products"] == threads * rounds and 1 <= st["largest"] <= max_batch and st["batches"] <= st["products"]


def test_requests_with_several_polys_and_mixed_levels(gpu_pkg, oracle):
    """a request may carry a batch of its own; requests of different levels never share a launch"""
    logn, nq, np_ = 12, 6, 2
    N, Q, P, bat, key, oplan, evk = _setup(gpu_pkg, oracle, logn, nq, np_, 8, 2)
    ring = gpu_pkg.ring
    errors, results = [], {}

    def evaluator(t):
        try:
            level, b = (5, 3) if t % 2 == 0 else (3, 2)
            cq = ring.NewContextWithParams(N, Q)
            ops = [gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, b, seed=50 * t + k).reshape(b, level + 1, N) for k in range(4)]
            mk = lambda x: cq.NewPolyLvl(level, b).set(x)
            out = (cq.NewPolyLvl(level, b), cq.NewPolyLvl(level, b))
            bat.MulRelin(level, (mk(ops[0]), mk(ops[1])), (mk(ops[2]), mk(ops[3])), key, out)
            results[t] = (level, b, ops, out[0].get().reshape(b, level + 1, N), out[1].get().reshape(b, level + 1, N))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=evaluator, args=(t,)) for t in range(6)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    for t, (level, b, ops, g0, g1) in results.items():
        for i in range(b):
            w0, w1 = oplan.mulrelin(level, np.stack([ops[0][i], ops[1][i]]), np.stack([ops[2][i], ops[3][i]]), evk)
            assert np.array_equal(g0[i], w0) and np.array_equal(g1[i], w1), (t, i)


def test_batcher_argument_errors(gpu_pkg):
    ring = gpu_pkg.ring
    N = 1 << 12
    _, Qf, Pf = gpu_pkg.params.ckks_moduli("PN15QP880")
    Q, P = list(Qf[:4]), list(Pf[:2])
    bat = ring.CkksBatcher(N, Q, P, max_batch=2, lanes=1)
    key = bat.NewSwitchingKey()
    cq = ring.NewContextWithParams(N, Q)
    mk = lambda b: cq.NewPoly(b)
    with pytest.raises(ring.LatticeRingError):                        # request larger than max_batch
        bat.MulRelin(3, (mk(3), mk(3)), (mk(3), mk(3)), key, (mk(3), mk(3)))
    with pytest.raises(ring.LatticeRingError):                        # level beyond the chain
        bat.MulRelin(4, (mk(1), mk(1)), (mk(1), mk(1)), key, (mk(1), mk(1)))
    with pytest.raises(ring.LatticeRingError):                        # batch mismatch between operands
        bat.MulRelin(3, (mk(1), mk(2)), (mk(1), mk(1)), key, (mk(1), mk(1)))
    # one plan twice is not two lanes
    lib, C = gpu_pkg._native.lib(), __import__("ctypes")
    arr = (C.c_void_p * 2)(bat.lanes[0][2].h, bat.lanes[0][2].h)
    h = C.c_void_p()
    assert lib.lr_ckks_batcher_create(arr, 2, C.byref(h)) != 0
    # still usable afterwards
    bat.MulRelin(3, (mk(1), mk(1)), (mk(1), mk(1)), key, (mk(1), mk(1)))
    assert bat.Stats()["products"] == 1


@pytest.mark.parametrize("logn,nq,np_,level,threads", [(12, 6, 2, 5, 8), (15, 7, 3, 6, 6), (16, 6, 2, 5, 4)])
def test_concurrent_rotations_and_products_through_one_batcher(gpu_pkg, oracle, logn, nq, np_, level, threads):
    """lr_ckks_batcher_rotate: rotations of concurrent callers share a launch per (level, Galois element, key); MulRelin requests of other
    callers go through the same lanes in between.  Every caller gets the rotation / product of its own ciphertext; two Galois elements
    (a rotation and the conjugation), in place for half of the callers."""
    N, Q, P, bat, key, oplan, evk = _setup(gpu_pkg, oracle, logn, nq, np_, 8, 2)
    ring = gpu_pkg.ring
    rot_h = gpu_pkg.sampling.uniform_poly(Q + P, N, evk.shape[0] * 2, seed=123)
    rotkey = bat.NewSwitchingKey().set(rot_h)
    rot_evk = rot_h.reshape(evk.shape[0], 2, nq + np_, N)
    gens = (pow(5, 3, 2 * N), 2 * N - 1)
    errors, results = [], {}

    def evaluator(t):
        try:
            cq = ring.NewContextWithParams(N, Q)
            mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 1, seed=s).reshape(1, level + 1, N)
            for r in range(2):
                a0, a1 = mk(500 * t + 10 * r), mk(500 * t + 10 * r + 1)
                ct = (cq.NewPolyLvl(level, 1).set(a0), cq.NewPolyLvl(level, 1).set(a1))
                if t % 3 == 2:
                    b0, b1 = mk(500 * t + 10 * r + 2), mk(500 * t + 10 * r + 3)
                    out = (cq.NewPolyLvl(level, 1), cq.NewPolyLvl(level, 1))
                    bat.MulRelin(level, ct, (cq.NewPolyLvl(level, 1).set(b0), cq.NewPolyLvl(level, 1).set(b1)), key, out)
                    want = ("mul", a0, a1, b0, b1)
                else:
                    gen = gens[(t + r) % 2]
                    out = ct if t % 2 == 0 else (cq.NewPolyLvl(level, 1), cq.NewPolyLvl(level, 1))
                    bat.PermuteNTT(level, ct, gen, rotkey, out)
                    want = ("rot", a0, a1, gen)
                results[(t, r)] = (want, out[0].get().reshape(level + 1, N), out[1].get().reshape(level + 1, N))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=evaluator, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    assert len(results) == 2 * threads
    for k, (what, g0, g1) in results.items():                          # (the oracle's plan is a single-threaded object: expectations here)
        if what[0] == "mul":
            want = oplan.mulrelin(level, np.stack([what[1][0], what[2][0]]), np.stack([what[3][0], what[4][0]]), evk)
        else:
            want = oplan.permute_ntt(level, np.stack([what[1][0], what[2][0]]), what[3], rot_evk)
        assert np.array_equal(g0, want[0]) and np.array_equal(g1, want[1]), (k, what[0])


def test_lanes_cannot_be_destroyed_under_a_live_batcher(gpu_pkg, oracle):
    """a lane's plan and contexts belong to the batcher while it lives (its destroy dereferences them and the contexts run on the lane's
    stream): destroying one first is refused with LR_ERR_ARG and changes nothing; after the batcher is gone the same calls succeed"""
    N, Q, P, bat, key, oplan, evk = _setup(gpu_pkg, oracle, 12, 4, 2, 4, 2)
    lib = gpu_pkg._native.lib()
    cq, cp, plan = bat.lanes[0]
    assert lib.lr_ckks_plan_destroy(plan.h) == 4 and b"batcher" in lib.lr_last_error_string()
    assert lib.lr_context_destroy(cq.h) == 4 and lib.lr_context_destroy(cp.h) == 4
    ring = gpu_pkg.ring
    level = len(Q) - 1
    ctx = ring.NewContextWithParams(N, Q)
    ops = [gpu_pkg.sampling.uniform_poly(Q, N, 1, seed=70 + k) for k in range(4)]
    mk = lambda k: ctx.NewPoly(1).set(ops[k])
    out = (ctx.NewPoly(1), ctx.NewPoly(1))
    bat.MulRelin(level, (mk(0), mk(1)), (mk(2), mk(3)), key, out)          # the refused calls left the lanes intact
    want = oplan.mulrelin(level, np.stack([ops[0][0], ops[1][0]]), np.stack([ops[2][0], ops[3][0]]), evk)
    assert np.array_equal(out[0].get(), want[0]) and np.array_equal(out[1].get(), want[1])
    h = bat.h
    bat.h = None
    lib.lr_ckks_batcher_destroy(h)
    assert lib.lr_ckks_plan_destroy(plan.h) == 0
    plan.h = None
    # the contexts are back on the library's stream and usable
    p = cq.NewPoly(1).set(ops[0])
    cq.NTT(p, p)
    assert np.array_equal(p.get(), oracle.Context(N, Q).ntt(ops[0][0]))


# ---- the BFV batcher: the reference's own pooled workload (examples/dbfv/psi/psi.go:219-228: evaluator.Mul + evaluator.Relinearize per task) ----
@pytest.mark.parametrize("name,logn,threads,lanes,max_batch", [("PN13QP218", 12, 8, 2, 4), ("PN14QP438", 14, 6, 2, 8), ("PN12QP109", 12, 5, 1, 3)])
def test_bfv_batcher_serves_mul_and_relinearize_per_caller(gpu_pkg, oracle, name, logn, threads, lanes, max_batch):
    """concurrent one-ciphertext callers, each running Mul then Relinearize on ITS operands through lr_bfv_batcher_*: every result against
    the oracle's bfv Mul / Relinearize, whatever batch a request ended up in (PN14QP438 = BASELINE config 4's set at full size)"""
    _, Q, P, QM = gpu_pkg.params.bfv_moduli(name)
    Q, P, QM = list(Q), list(P), list(QM)
    N = 1 << logn
    ring = gpu_pkg.ring
    nq, np_ = len(Q), len(P)
    beta = -(-nq // np_)
    bat = ring.BfvBatcher(N, Q, P, QM, 65537, max_batch=max_batch, lanes=lanes)
    evk = gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=17)
    key = bat.NewSwitchingKey().set(evk)
    omul = oracle.BfvPlan(oracle.Context(N, Q), oracle.Context(N, QM), 65537)
    oks = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    rounds = 2
    errors, results = [], {}

    def evaluator(t):
        try:
            cq = ring.NewContextWithParams(N, Q)
            for r in range(rounds):
                seed = 500 * t + 10 * r
                ops = [gpu_pkg.sampling.uniform_poly(Q, N, 1, seed=seed + k) for k in range(4)]
                mk = lambda k: cq.NewPoly(1).set(ops[k])
                deg2 = (cq.NewPoly(1), cq.NewPoly(1), cq.NewPoly(1))
                bat.Mul((mk(0), mk(1)), (mk(2), mk(3)), deg2)
                lin = (cq.NewPoly(1), cq.NewPoly(1))
                bat.Relinearize(deg2, key, lin)
                results[(t, r)] = (ops, [p.get() for p in deg2], [p.get() for p in lin])
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=evaluator, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    assert len(results) == threads * rounds
    for (t, r), (ops, deg2, lin) in results.items():
        want2 = omul.mul(np.stack([ops[0][0], ops[1][0]]), np.stack([ops[2][0], ops[3][0]]))
        for k in range(3):
            assert np.array_equal(deg2[k], want2[k]), (t, r, "mul", k)
        want1 = oks.bfv_relinearize(want2, evk.reshape(beta, 2, nq + np_, N))
        for k in range(2):
            assert np.array_equal(lin[k], want1[k]), (t, r, "relin", k)
    st = bat.Stats()
    assert st["products"] == 2 * threads * rounds and 1 <= st["largest"] <= max_batch
    # refused requests, and lanes that cannot be destroyed under the live batcher
    cq = ring.NewContextWithParams(N, Q)
    a = cq.NewPoly(1)
    err = gpu_pkg._native.LatticeRingError
    with pytest.raises(err):
        bat.Mul((a, a), (a, a), (a, a, cq.NewPoly(1)))                       # result polys must be distinct
    with pytest.raises(err):
        big = cq.NewPoly(max_batch + 1)
        bat.Mul((big, big), (big, big), (cq.NewPoly(max_batch + 1), cq.NewPoly(max_batch + 1), cq.NewPoly(max_batch + 1)))
    lib = gpu_pkg._native.lib()
    assert lib.lr_bfv_plan_destroy(bat.lanes[0][3].h) == 4 and lib.lr_ckks_plan_destroy(bat.lanes[0][4].h) == 4 and lib.lr_context_destroy(bat.lanes[0][2].h) == 4

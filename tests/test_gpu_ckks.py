"""Parity of the device-resident ckks.Evaluator call sequences (switchKeysInPlace, MulRelin, Rescale)
against the CPU oracle's restatement of the same ring calls on the same synthetic operands."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ckks(gpu_pkg, oracle, logn, nq, np_, batch, max_batch=None):
    N = 1 << logn
    _, Qf, Pf = gpu_pkg.params.ckks_moduli("PN15QP880")
    Q, P = Qf[:nq], Pf[:np_]
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, max_batch or batch)
    oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    beta = -(-nq // np_)
    evk = gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=99)   # [2*beta, nQ+nP, N], any values < q
    pevk = plan.NewSwitchingKey().set(evk)
    return N, Q, P, cQ, cP, plan, oplan, evk.reshape(beta, 2, nq + np_, N), pevk


@pytest.mark.parametrize("logn,nq,np_,level", [(10, 6, 2, 5), (10, 6, 2, 4), (10, 6, 2, 2), (11, 7, 3, 6), (10, 18, 3, 17),
                                               (10, 18, 3, 12)])
def test_switch_keys(gpu_pkg, oracle, logn, nq, np_, level):
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, 2)
    cx = gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 2, seed=level)
    pcx = cQ.NewPolyLvl(level, 2).set(cx)
    p0, p1 = cQ.NewPolyLvl(level, 2), cQ.NewPolyLvl(level, 2)
    plan.SwitchKeysInPlace(level, pcx, pevk, p0, p1)
    for b in range(2):
        w0, w1 = oplan.switch_keys(level, cx[b], evk)
        assert np.array_equal(p0.get()[b], w0)
        assert np.array_equal(p1.get()[b], w1)


@pytest.mark.parametrize("logn,nq,np_,level", [(10, 6, 2, 5), (12, 18, 3, 17), (11, 18, 3, 9)])
def test_mulrelin_and_rescale(gpu_pkg, oracle, logn, nq, np_, level):
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, 2)
    mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 2, seed=s)
    a0, a1, b0, b1 = mk(1), mk(2), mk(3), mk(4)
    P_ = lambda x: cQ.NewPolyLvl(level, 2).set(x)
    ct0, ct1 = (P_(a0), P_(a1)), (P_(b0), P_(b1))
    out = (cQ.NewPolyLvl(level, 2), cQ.NewPolyLvl(level, 2))
    plan.MulRelin(level, ct0, ct1, pevk, out)
    wants = []
    for b in range(2):
        want = oplan.mulrelin(level, np.stack([a0[b], a1[b]]), np.stack([b0[b], b1[b]]), evk)
        wants.append(want)
        assert np.array_equal(out[0].get()[b], want[0])
        assert np.array_equal(out[1].get()[b], want[1])
    # Rescale, ckks/evaluator.go:958-960
    plan.Rescale(out)
    oc = oracle.Context(N, Q)
    for b in range(2):
        for k in range(2):
            assert np.array_equal(out[k].get()[b], oc.rescale_op("oc_div_round_by_last_modulus_ntt", wants[b][k]))

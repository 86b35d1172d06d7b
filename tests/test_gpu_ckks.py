"""Parity of the device-resident ckks.Evaluator call sequences (switchKeysInPlace, MulRelin, Rescale)
against the CPU oracle's restatement of the same ring calls on the same synthetic operands."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ckks(gpu_pkg, oracle, logn, nq, np_, batch, max_batch=None):
    N = 1 << logn
    # NTT-friendly up to 2^logN; PN15QP880 has three special primes, PN16QP1761 four
    _, Qf, Pf = gpu_pkg.params.ckks_moduli("PN16QP1761" if logn == 16 or np_ > 3 else "PN15QP880")
    assert len(Qf) >= nq and len(Pf) >= np_
    Q, P = Qf[:nq], Pf[:np_]
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, max_batch or batch)
    oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    beta = -(-nq // np_)
    evk = gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=99)   # [2*beta, nQ+nP, N], any values < q
    pevk = plan.NewSwitchingKey().set(evk)
    return N, Q, P, cQ, cP, plan, oplan, evk.reshape(beta, 2, nq + np_, N), pevk


# the N = 2^14 cases run on the assembly NTT kernels, whose grouped launch transforms every full digit at once
# (grid z = digit, own limbs skipped); levels 6 and 4 end in a partial digit that takes the per-digit path
@pytest.mark.parametrize("logn,nq,np_,level", [(10, 6, 2, 5), (10, 6, 2, 4), (10, 6, 2, 2), (11, 7, 3, 6), (10, 18, 3, 17),
                                               (10, 18, 3, 12), (14, 7, 3, 6), (14, 7, 3, 5), (14, 6, 2, 4), (14, 4, 2, 1),
                                               # alpha = 4 digits (DefaultParams[PN16QP1761]'s shape) at a small degree, full and partial last digit
                                               (10, 10, 4, 9), (10, 10, 4, 5), (13, 9, 4, 8),
                                               # BASELINE config 5 at full size: PN16QP1761, 34 Q + 4 P limbs, beta = 9 (ckks/params.go:78-86)
                                               (16, 34, 4, 33), (16, 34, 4, 20)])
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


# (16, 5, 2, 4): BASELINE config 5's degree (N = 2^16: two-pass NTT) on a short modulus chain
# (15, 18, 3, 17) is BASELINE config 3 at full size: DefaultParams[PN15QP880], level 17
# (16, 34, 4, 33) is BASELINE config 5's per-GPU workload at full size: DefaultParams[PN16QP1761] (ckks/params.go:78-86), level 33;
# level 20 ends in a partial digit (21 limbs = 5 digits of 4 + 1 limb) and takes the trivial-copy branch of the decomposer
@pytest.mark.parametrize("logn,nq,np_,level", [(10, 6, 2, 5), (12, 18, 3, 17), (11, 18, 3, 9), (15, 5, 2, 4), (16, 5, 2, 4),
                                               (15, 18, 3, 17), (12, 10, 4, 9), (16, 34, 4, 33), (16, 34, 4, 20), (16, 34, 4, 22)])
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


@pytest.mark.parametrize("name,logn", [("PN12QP109", 12), ("PN13QP218", 11), ("PN14QP438", 14)])
def test_bfv_mul(gpu_pkg, oracle, name, logn):
    """bfv.Evaluator.Mul = tensorAndRescale (bfv/evaluator.go:278-467) on the reference's parameter sets
    (PN14QP438 = BASELINE config 4: 6 Q primes, 6 61-bit QMul primes, t = 65537); degree-1 x degree-1."""
    _, Q, P, QMul = gpu_pkg.params.bfv_moduli(name)
    N, t = 1 << logn, 65537
    ring = gpu_pkg.ring
    cQ, cM = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, QMul)
    plan = ring.BfvPlan(cQ, cM, t, 2)
    oplan = oracle.BfvPlan(oracle.Context(N, Q), oracle.Context(N, QMul), t)
    mk = lambda s: gpu_pkg.sampling.uniform_poly(Q, N, 2, seed=s)
    a0, a1, b0, b1 = mk(11), mk(12), mk(13), mk(14)
    P_ = lambda x: cQ.NewPoly(2).set(x)
    out = (cQ.NewPoly(2), cQ.NewPoly(2), cQ.NewPoly(2))
    plan.Mul((P_(a0), P_(a1)), (P_(b0), P_(b1)), out)
    for b in range(2):
        want = oplan.mul(np.stack([a0[b], a1[b]]), np.stack([b0[b], b1[b]]))
        for k in range(3):
            assert np.array_equal(out[k].get()[b], want[k]), (b, k)


@pytest.mark.parametrize("env", [{}, {"LR_BFV_NO_GATHER": "1"}])
@pytest.mark.parametrize("name,logn,batch", [("PN12QP109", 12, 2), ("PN13QP218", 11, 3), ("PN14QP438", 14, 1), ("PN14QP438", 14, 80)])
def test_bfv_square(gpu_pkg, oracle, name, logn, batch, env, monkeypatch):
    """evaluator.Mul(ct, ct, .) (the reference's Square benchmark, bfv/bfv_benchmark_test.go:139): the operand is lifted and transformed
    once (bfv/evaluator.go:306) and the tensor kernel reads its slots for both factors; against the oracle's restatement of the squaring case, on the gathered
    small-batch path and on the per-operand one (batch 80 at N = 2^14 is above the gather threshold)"""
    monkeypatch.delenv("LR_BFV_NO_GATHER", raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _, Q, P, QMul = gpu_pkg.params.bfv_moduli(name)
    N, t = 1 << logn, 65537
    ring = gpu_pkg.ring
    cQ, cM = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, QMul)
    plan = ring.BfvPlan(cQ, cM, t, batch)
    oplan = oracle.BfvPlan(oracle.Context(N, Q), oracle.Context(N, QMul), t)
    mk = lambda s: gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=s).reshape(batch, len(Q), N)
    a0, a1 = mk(31), mk(32)
    ct = (cQ.NewPoly(batch).set(a0), cQ.NewPoly(batch).set(a1))
    out = (cQ.NewPoly(batch), cQ.NewPoly(batch), cQ.NewPoly(batch))
    plan.Mul(ct, ct, out)
    got = [o.get().reshape(batch, len(Q), N) for o in out]
    for b in sorted({0, batch - 1}):
        want = oplan.square(np.stack([a0[b], a1[b]]))
        for k in range(3):
            assert np.array_equal(got[k][b], want[k]), (b, k)
    if batch > 2:                                                   # every unit: against the regular path on two copies of the operand
        ct2 = (cQ.NewPoly(batch).set(a0), cQ.NewPoly(batch).set(a1))
        out2 = (cQ.NewPoly(batch), cQ.NewPoly(batch), cQ.NewPoly(batch))
        plan.Mul(ct, ct2, out2)
        for k in range(3):
            assert np.array_equal(out2[k].get(), out[k].get()), k
    plan.Mul(ct, ct, (ct[0], ct[1], out[2]))                        # the result over the operand
    assert np.array_equal(ct[0].get().reshape(batch, len(Q), N), got[0]) and np.array_equal(ct[1].get().reshape(batch, len(Q), N), got[1])


@pytest.mark.parametrize("name,logn", [("PN13QP218", 11), ("PN14QP438", 14)])
def test_bfv_mul_without_extension_epilogues(gpu_pkg, oracle, name, logn, monkeypatch):
    """the same with the subtract-multiply of the ModDown and the SubScalar / MulScalar tail as separate passes instead of in the
    extension kernels' stores (the switch is read when the plan is created)"""
    monkeypatch.setenv("LR_BFV_NO_EXT_EPILOGUE", "1")
    test_bfv_mul(gpu_pkg, oracle, name, logn)


@pytest.mark.parametrize("name,logn", [("PN13QP218", 11), ("PN14QP438", 14), ("PN15QP880", 15)])
def test_bfv_mul_small_batch_and_per_operand_paths(gpu_pkg, oracle, name, logn, monkeypatch):
    """at a small batch the four operand polys are gathered into one batch of 4 B and the three products leave as one batch of 3 B
    (every step one launch); LR_BFV_NO_GATHER keeps one set of launches per operand / product as large batches have it.  One ciphertext
    pair and three, both ways, with and without the extension epilogues, against the oracle; the output may be an operand."""
    _, Q, P, QMul = gpu_pkg.params.bfv_moduli(name)
    N, t = 1 << logn, 65537
    ring = gpu_pkg.ring
    oplan = oracle.BfvPlan(oracle.Context(N, Q), oracle.Context(N, QMul), t)
    for batch in (1, 3):
        mk = lambda s: gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=s).reshape(batch, len(Q), N)
        a0, a1, b0, b1 = mk(21), mk(22), mk(23), mk(24)
        wants = [oplan.mul(np.stack([a0[b], a1[b]]), np.stack([b0[b], b1[b]])) for b in range(batch)]
        # (LR_NO_EXT_CHUNKS: the extensions as one launch over all target columns instead of column ranges on grid z)
        for env in ({}, {"LR_BFV_NO_GATHER": "1"}, {"LR_BFV_NO_EXT_EPILOGUE": "1"}, {"LR_NO_EXT_CHUNKS": "1"}, {"LR_BFV_NO_GATHER": "1", "LR_NO_EXT_CHUNKS": "1"}):
            for k in ("LR_BFV_NO_GATHER", "LR_BFV_NO_EXT_EPILOGUE", "LR_NO_EXT_CHUNKS"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            cQ, cM = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, QMul)
            plan = ring.BfvPlan(cQ, cM, t, batch)
            P_ = lambda x: cQ.NewPoly(batch).set(x)
            ct0, ct1 = (P_(a0), P_(a1)), (P_(b0), P_(b1))
            out = (ct0[0], ct1[1], cQ.NewPoly(batch))            # two of the outputs are operands
            plan.Mul(ct0, ct1, out)
            for b in range(batch):
                for k in range(3):
                    assert np.array_equal(out[k].get().reshape(batch, len(Q), N)[b], wants[b][k]), (batch, env, b, k)


def _galois(gpu_pkg, N, k):
    """Galois element of a left rotation by k: 5^k mod 2N (ckks/keygen.go:281-286 GenRot... -> ring.PermuteNTTIndex(GaloisGen, k, N))"""
    return pow(5, k, 2 * N)


@pytest.mark.parametrize("logn,nq,np_,level,k", [(10, 6, 2, 5, 1), (11, 7, 3, 6, 5), (10, 6, 2, 3, 17), (14, 6, 2, 5, 3), (12, 18, 3, 17, 2), (16, 6, 2, 5, 7)])
def test_rotate_columns(gpu_pkg, oracle, logn, nq, np_, level, k):
    """evaluator.permuteNTT (ckks/evaluator.go:1448) = RotateColumns with the key of that rotation: both components
    permuted, the second key-switched, against the oracle's restatement; also in place and the conjugation element"""
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, 2)
    mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 2, seed=s)
    a0, a1 = mk(31), mk(32)
    P_ = lambda x: cQ.NewPolyLvl(level, 2).set(x)
    for gen in (_galois(gpu_pkg, N, k), 2 * N - 1):
        ct = (P_(a0), P_(a1))
        out = (cQ.NewPolyLvl(level, 2), cQ.NewPolyLvl(level, 2))
        plan.PermuteNTT(level, ct, gen, pevk, out)
        for b in range(2):
            want = oplan.permute_ntt(level, np.stack([a0[b], a1[b]]), gen, evk)
            assert np.array_equal(out[0].get()[b], want[0]), (gen, b)
            assert np.array_equal(out[1].get()[b], want[1]), (gen, b)
        got = (out[0].get(), out[1].get())
        plan.PermuteNTT(level, ct, gen, pevk, ct)       # ctOut == ct0
        assert np.array_equal(ct[0].get(), got[0]) and np.array_equal(ct[1].get(), got[1])


# (16, 6, 2, 5): N = 2^16 with the top stage inside the extension kernel and the own limbs copied into the digits
@pytest.mark.parametrize("logn,nq,np_,level", [(10, 6, 2, 5), (11, 7, 3, 6), (10, 6, 2, 2), (14, 6, 2, 4), (16, 6, 2, 5), (16, 5, 2, 4)])
def test_rotate_hoisted(gpu_pkg, oracle, logn, nq, np_, level):
    """RotateHoisted (ckks/evaluator.go:1252): several rotations share one digit decomposition; every output equals
    the oracle's restatement of switchKeyHoisted.  (Hoisted and plain rotations are not bit-identical: permuting the
    extended digits differs from extending the permuted digit by multiples of the digit modulus, which the reference
    only compares after decryption.)"""
    N, Q, P, cQ, cP, plan, oplan, evk0, pevk0 = _ckks(gpu_pkg, oracle, logn, nq, np_, 2)
    beta = -(-nq // np_)
    rots = [1, 3, 4]
    gens = [_galois(gpu_pkg, N, k) for k in rots]
    evks = [gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=200 + k) for k in rots]
    pevks = [plan.NewSwitchingKey().set(e) for e in evks]
    mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 2, seed=s)
    a0, a1 = mk(41), mk(42)
    ct = (cQ.NewPolyLvl(level, 2).set(a0), cQ.NewPolyLvl(level, 2).set(a1))
    outs = [(cQ.NewPolyLvl(level, 2), cQ.NewPolyLvl(level, 2)) for _ in rots]
    plan.RotateHoisted(level, ct, gens, pevks, outs)
    for b in range(2):
        want = oplan.rotate_hoisted(level, np.stack([a0[b], a1[b]]), gens, [e.reshape(beta, 2, nq + np_, N) for e in evks])
        for r in range(len(rots)):
            assert np.array_equal(outs[r][0].get()[b], want[r][0]), (r, b)
            assert np.array_equal(outs[r][1].get()[b], want[r][1]), (r, b)
    # the first component does not go through the key switch: plain and hoisted agree on everything but the noise term
    assert not np.array_equal(outs[0][0].get(), outs[1][0].get())


def test_rotate_columns_pow2_chain(gpu_pkg, oracle):
    """rotateColumnsPow2 (ckks/evaluator.go:1408): rotation by k = 5 as the chain of the rotations by 1 and 4, each a
    permuteNTT with its own key; equals the oracle's permute_ntt applied in the same order"""
    logn, nq, np_, level = 11, 6, 2, 5
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, 2)
    beta = -(-nq // np_)
    keys, okeys = {}, {}
    for i in (1, 2, 4):
        e = gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=300 + i)
        keys[i] = (_galois(gpu_pkg, N, i), plan.NewSwitchingKey().set(e))
        okeys[i] = e.reshape(beta, 2, nq + np_, N)
    mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 2, seed=s)
    a0, a1 = mk(51), mk(52)
    ct = (cQ.NewPolyLvl(level, 2).set(a0), cQ.NewPolyLvl(level, 2).set(a1))
    out = (cQ.NewPolyLvl(level, 2), cQ.NewPolyLvl(level, 2))
    plan.RotateColumnsPow2(level, ct, 5, keys, out)
    for b in range(2):
        want = np.stack([a0[b], a1[b]])
        for i in (1, 4):
            want = oplan.permute_ntt(level, want, keys[i][0], okeys[i])
        assert np.array_equal(out[0].get()[b], want[0]) and np.array_equal(out[1].get()[b], want[1])


@pytest.mark.parametrize("nq,np_,level", [(5, 2, 4), (6, 2, 5), (7, 3, 5), (10, 4, 9)])
def test_mulrelin_2p16_paths(gpu_pkg, oracle, nq, np_, level, monkeypatch):
    """N = 2^16, the three ways the key switch feeds its forward transforms: the top stage applied by the extension kernel + plain
    sub-block kernels in place (default where every digit is extended: (6,2,5), (7,3,5), (10,4,9); (5,2,4) ends in a one-limb
    digit and falls back), staged extensions + fused-top kernels (LR_NO_EXTTOP), in-place transforms behind the separate top-stage
    pass (LR_NO_EXTTOP + LR_NO_STAGING)"""
    for env in ({}, {"LR_NO_EXTTOP": "1"}, {"LR_NO_EXTTOP": "1", "LR_NO_STAGING": "1"}, {"LR_NO_INVTOP": "1"}):
        for k in ("LR_NO_EXTTOP", "LR_NO_STAGING", "LR_NO_INVTOP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        test_mulrelin_and_rescale(gpu_pkg, oracle, 16, nq, np_, level)
        test_switch_keys(gpu_pkg, oracle, 16, nq, np_, level)


def test_2p16_epilogue_kernels_are_the_ones_that_run(gpu_pkg, oracle, monkeypatch):
    """N = 2^16: ModDown's subtract-multiply-add rides in the copy-out of the plain sub-block kernels (input rows carry the top
    stage from the extension kernel) and Rescale's in the fused-top ones (the last limb's row is read by every other row's
    transform, batch > 1 included); parity of both is test_mulrelin_and_rescale / test_mulrelin_2p16_paths"""
    for k in ("LR_NO_EXTTOP", "LR_NO_STAGING", "LR_NO_EPILOGUE"):
        monkeypatch.delenv(k, raising=False)
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, 16, 6, 2, 2)
    level = 5
    mk = lambda s: cQ.NewPolyLvl(level, 2).set(gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 2, seed=s))
    out = (cQ.NewPolyLvl(level, 2), cQ.NewPolyLvl(level, 2))
    plan.MulRelin(level, (mk(1), mk(2)), (mk(3), mk(4)), pevk, out)
    assert cQ.last_ntt_kernel() == "lr_ntt_fwd16p_m4"
    plan.Rescale(out)
    assert cQ.last_ntt_kernel() == "lr_ntt_fwd16s_m4"


@pytest.mark.parametrize("logn,nq,np_,level", [(12, 18, 3, 17), (15, 5, 2, 4), (16, 6, 2, 5)])
def test_mulrelin_and_rescale_without_epilogue(gpu_pkg, oracle, logn, nq, np_, level, monkeypatch):
    """ModDown and the rounding rescale with the separate subtract-multiply pass instead of the forward kernels' epilogue"""
    monkeypatch.setenv("LR_NO_EPILOGUE", "1")
    test_mulrelin_and_rescale(gpu_pkg, oracle, logn, nq, np_, level)


# ---- MulRelin's other branches (ckks/evaluator.go:1038-1131) ---------------------------------------------------------------
@pytest.mark.parametrize("logn,nq,np_,level", [(10, 6, 2, 5), (13, 6, 2, 3), (15, 18, 3, 17)])
def test_mul_without_relinearisation_key(gpu_pkg, oracle, logn, nq, np_, level):
    """evakey == nil: the degree-2 tensor (:1061-1066, :1105-1111); regular case, squaring case (el0 == el1, :1083-1088) and the
    receiver being one of the inputs"""
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, 2)
    mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 2, seed=s)
    a0, a1, b0, b1 = mk(1), mk(2), mk(3), mk(4)
    P_ = lambda x: cQ.NewPolyLvl(level, 2).set(x)
    ct0, ct1 = (P_(a0), P_(a1)), (P_(b0), P_(b1))
    out = tuple(cQ.NewPolyLvl(level, 2) for _ in range(3))
    plan.MulRelin(level, ct0, ct1, None, out)
    for b in range(2):
        want = oplan.mul_norelin(level, np.stack([a0[b], a1[b]]), np.stack([b0[b], b1[b]]))
        for k in range(3):
            assert np.array_equal(out[k].get()[b], want[k]), (b, k)
    # squaring: the same operand handles on both sides, against the oracle's restatement of the el0 == el1 branch
    plan.MulRelin(level, ct0, ct0, None, out)
    for b in range(2):
        ct = np.stack([a0[b], a1[b]])
        want = oplan.mul_norelin(level, ct, ct, squaring=True)
        assert np.array_equal(want, oplan.mul_norelin(level, ct, ct, squaring=False))     # the two branches agree
        for k in range(3):
            assert np.array_equal(out[k].get()[b], want[k]), (b, k)
    # ctOut == ct0: components 0 and 1 of the result overwrite the first operand (:1105-1111)
    c2 = cQ.NewPolyLvl(level, 2)
    plan.MulRelin(level, ct0, ct1, None, (ct0[0], ct0[1], c2))
    for b in range(2):
        want = oplan.mul_norelin(level, np.stack([a0[b], a1[b]]), np.stack([b0[b], b1[b]]))
        assert np.array_equal(ct0[0].get()[b], want[0]) and np.array_equal(ct0[1].get()[b], want[1]) and np.array_equal(c2.get()[b], want[2])


@pytest.mark.parametrize("logn,nq,np_,level,pt_batch", [(10, 6, 2, 5, 1), (13, 6, 2, 3, 2), (15, 18, 3, 17, 1)])
def test_mul_plaintext_ciphertext(gpu_pkg, oracle, logn, nq, np_, level, pt_batch):
    """plaintext x ciphertext and ciphertext x plaintext (:1113-1131), a shared plaintext (batch 1) and one per ciphertext"""
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, 2)
    mk = lambda s, n=2: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, n, seed=s).reshape(n, level + 1, N)
    a0, a1, m = mk(1), mk(2), mk(7, pt_batch)
    ct = (cQ.NewPolyLvl(level, 2).set(a0), cQ.NewPolyLvl(level, 2).set(a1))
    pt = (cQ.NewPolyLvl(level, pt_batch).set(m),)
    for ops in ((pt, ct), (ct, pt)):
        out = (cQ.NewPolyLvl(level, 2), cQ.NewPolyLvl(level, 2))
        plan.MulRelin(level, ops[0], ops[1], None, out)
        for b in range(2):
            want = oplan.mul_plain(level, m[b % pt_batch], np.stack([a0[b], a1[b]]))
            assert np.array_equal(out[0].get()[b], want[0]) and np.array_equal(out[1].get()[b], want[1]), b


def test_switch_keys_outputs_with_different_strides(gpu_pkg, oracle):
    """p0 and p1 allocated with different limb counts (strides): the inner product carries one stride per output"""
    logn, nq, np_, level = 12, 6, 2, 4
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, 2)
    cx = gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 2, seed=3)
    pcx = cQ.NewPolyLvl(level, 2).set(cx)
    p0, p1 = cQ.NewPolyLvl(level, 2), cQ.NewPoly(2)
    plan.SwitchKeysInPlace(level, pcx, pevk, p0, p1)
    for b in range(2):
        w0, w1 = oplan.switch_keys(level, cx[b], evk)
        assert np.array_equal(p0.get()[b], w0)
        assert np.array_equal(p1.get()[b][:level + 1], w1)


# ---- encrypt / decrypt tails as C-ABI entry points ------------------------------------------------------------
@pytest.mark.parametrize("logn,nq,np_,batch", [(10, 4, 2, 2), (13, 6, 3, 3), (15, 18, 3, 2)])
def test_encrypt_pk_and_decrypt_against_oracle(gpu_pkg, oracle, logn, nq, np_, batch):
    """lr_ckks_encrypt_pk (ckks/encryptor.go:205-234, after the sampling) and lr_ckks_decrypt (ckks/decryptor.go:53-78) against
    the oracle's restatement on the same synthetic operands, bit for bit; (15, 18, 3) is DefaultParams[PN15QP880] at full size"""
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, batch)
    QP = Q + P
    level = nq - 1
    ocQP = oracle.Context(N, QP)
    uni = lambda s, n: gpu_pkg.sampling.uniform_poly(QP, N, n, seed=s).reshape(n, nq + np_, N)
    u, e0, e1 = uni(1, batch), uni(2, batch), uni(3, batch)
    for i, q in enumerate(QP):
        e0[0, i, 0] = q                                   # SampleAndAdd's residue of -0 (ring/gaussianSampler.go:268)
    pk0, pk1 = uni(4, 1), uni(5, 1)
    pt = gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=6).reshape(batch, nq, N)
    QPpoly = lambda x: gpu_pkg.ring.Poly(cQ, nq + np_, x.shape[0]).set(x)
    ct = (cQ.NewPoly(batch), cQ.NewPoly(batch))
    plan.EncryptPk(level, QPpoly(u), (QPpoly(pk0), QPpoly(pk1)), (QPpoly(e0), QPpoly(e1)), cQ.NewPoly(batch).set(pt), ct)
    wants = []
    for b in range(batch):
        want = oplan.encrypt_pk(ocQP, level, u[b], pk0[0], pk1[0], e0[b], e1[b], pt[b])
        wants.append(want)
        assert np.array_equal(ct[0].get().reshape(batch, nq, N)[b], want[0]), b
        assert np.array_equal(ct[1].get().reshape(batch, nq, N)[b], want[1]), b
    sk = gpu_pkg.sampling.uniform_poly(Q, N, 1, seed=8).reshape(1, nq, N)
    psk = cQ.NewPoly(1).set(sk)
    # degree 1 (fresh ciphertext) and degree 2 (a product before relinearisation): Horner with the reference's cadence
    extra = gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=9).reshape(batch, nq, N)
    pextra = cQ.NewPoly(batch).set(extra)
    for cts, stack in (((ct[0], ct[1]), lambda b: np.stack([wants[b][0], wants[b][1]])),
                       ((ct[0], ct[1], pextra), lambda b: np.stack([wants[b][0], wants[b][1], extra[b]]))):
        out = cQ.NewPoly(batch)
        plan.Decrypt(level, cts, psk, out)
        for b in range(batch):
            assert np.array_equal(out.get().reshape(batch, nq, N)[b], oplan.decrypt(level, stack(b), sk[0])), (len(cts), b)


def test_decrypt_degree_seven_reduction_cadence(gpu_pkg, oracle):
    """degree 7 and 8: the `i&7 == 7` lazy reduction inside the loop and the skipped final one (ckks/decryptor.go:70-77)"""
    logn, nq, np_ = 8, 3, 1
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, 1)
    level = nq - 1
    sk = gpu_pkg.sampling.uniform_poly(Q, N, 1, seed=1).reshape(1, nq, N)
    psk = cQ.NewPoly(1).set(sk)
    for degree in (7, 8):
        ct = gpu_pkg.sampling.uniform_poly(Q, N, degree + 1, seed=degree).reshape(degree + 1, nq, N)
        polys = [cQ.NewPoly(1).set(ct[i][None]) for i in range(degree + 1)]
        out = cQ.NewPoly(1)
        plan.Decrypt(level, polys, psk, out)
        assert np.array_equal(out.get().reshape(nq, N), oplan.decrypt(level, ct, sk[0])), degree


@pytest.mark.parametrize("logn,nq,np_,level", [(12, 18, 3, 17), (14, 7, 3, 6), (16, 10, 4, 9)])
def test_key_inner_product_per_term_fallback(gpu_pkg, oracle, logn, nq, np_, level, monkeypatch):
    """the key inner product with one Montgomery product per term (LR_KEYMAC_NARROW) instead of the exact 128-bit sums"""
    monkeypatch.setenv("LR_KEYMAC_NARROW", "1")
    test_switch_keys(gpu_pkg, oracle, logn, nq, np_, level)


def test_switch_keys_accepts_non_canonical_own_limbs(gpu_pkg, oracle):
    """cx values in [q, 2q) -- what the reference's lazy operations leave and its InvNTT still takes -- reach the
    inner product unreduced in the limbs a digit owns (ckks/evaluator.go:1579-1584): MRed accepts them, and so does the
    128-bit sum (its final BRedAdd takes the whole 64-bit range)"""
    logn, nq, np_, level = 12, 6, 2, 5
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, 2)
    cx = gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 2, seed=5)
    for i, q in enumerate(Q[:level + 1]):
        cx[0, i, :8] = np.uint64(2 * q - 1)
        cx[1, i, 1] = np.uint64(2 * q - 1)
        cx[1, i, 2] = np.uint64(q)
    pcx = cQ.NewPolyLvl(level, 2).set(cx)
    p0, p1 = cQ.NewPolyLvl(level, 2), cQ.NewPolyLvl(level, 2)
    plan.SwitchKeysInPlace(level, pcx, pevk, p0, p1)
    for b in range(2):
        w0, w1 = oplan.switch_keys(level, cx[b], evk)
        assert np.array_equal(p0.get()[b], w0) and np.array_equal(p1.get()[b], w1)


def test_key_inner_product_wide_and_per_term_agree_on_lazy_own_limbs(gpu_pkg, monkeypatch):
    """60-bit moduli, beta = 4 digits, own limbs anywhere below 4q (the InvNTT's input range): the 128-bit-sum inner product reduces a
    digit's own operand before it enters the carry-less middle column (lr_ewise.hip: own_operand), so it agrees with the per-term
    kernel (LR_KEYMAC_NARROW) wherever the caller's values lie"""
    ring, params, sampling = gpu_pkg.ring, gpu_pkg.params, gpu_pkg.sampling
    N = 1 << 12
    Q, P = list(params.Qi60()[:8]), list(params.Pi60()[:2])
    level, beta = 7, 4
    evk = sampling.uniform_poly(Q + P, N, 2 * beta, seed=77)
    cx = sampling.uniform_poly(Q, N, 2, seed=78)
    rng = np.random.default_rng(3)
    for i, q in enumerate(Q):
        k = rng.integers(0, 4, size=(2, N), dtype=np.uint64)
        cx[:, i, :] += k * np.uint64(q)                     # anywhere in [0, 4q)
        cx[0, i, :4] = np.uint64(4 * q - 1)
    got = []
    for narrow in (False, True):
        if narrow:
            monkeypatch.setenv("LR_KEYMAC_NARROW", "1")
        else:
            monkeypatch.delenv("LR_KEYMAC_NARROW", raising=False)
        cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
        plan = ring.CkksPlan(cQ, cP, 2)
        pevk = plan.NewSwitchingKey().set(evk)
        p0, p1 = cQ.NewPolyLvl(level, 2), cQ.NewPolyLvl(level, 2)
        plan.SwitchKeysInPlace(level, cQ.NewPolyLvl(level, 2).set(cx), pevk, p0, p1)
        got.append((p0.get(), p1.get()))
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    for i, q in enumerate(Q):
        assert int(got[0][0][:, i].max()) < q and int(got[0][1][:, i].max()) < q


@pytest.mark.parametrize("name,logn", [("PN12QP109", 12), ("PN13QP218", 11), ("PN14QP438", 14), ("PN15QP880", 12)])
def test_bfv_relinearize(gpu_pkg, oracle, name, logn, monkeypatch):
    """bfv.evaluator.Relinearize / switchKeys (bfv/evaluator.go:480-501, 736-812) on the reference's BFV parameter sets (PN14QP438 =
    BASELINE config 4 at full size), degree 2 -> 1, coefficient domain in and out, against the oracle's restatement; in place; and
    with the subtract-multiply of the ModDown as a separate pass (LR_NO_EPILOGUE)"""
    _, Q, P, _ = gpu_pkg.params.bfv_moduli(name)
    N = 1 << logn
    ring = gpu_pkg.ring
    nq, np_ = len(Q), len(P)
    beta = -(-nq // np_)
    evk = gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=41)
    ct = [gpu_pkg.sampling.uniform_poly(Q, N, 2, seed=50 + k) for k in range(3)]
    oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    wants = [oplan.bfv_relinearize(np.stack([ct[0][b], ct[1][b], ct[2][b]]), evk.reshape(beta, 2, nq + np_, N)) for b in range(2)]
    for no_epi in (False, True):
        if no_epi:
            monkeypatch.setenv("LR_NO_EPILOGUE", "1")
        cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
        plan = ring.CkksPlan(cQ, cP, 2)
        pevk = plan.NewSwitchingKey().set(evk)
        c = [cQ.NewPoly(2).set(x) for x in ct]
        out = (cQ.NewPoly(2), cQ.NewPoly(2))
        plan.BfvRelinearize(c, pevk, out)
        for b in range(2):
            assert np.array_equal(out[0].get()[b], wants[b][0]), (no_epi, b)
            assert np.array_equal(out[1].get()[b], wants[b][1]), (no_epi, b)
        # switchKeys alone
        p0, p1 = cQ.NewPoly(2), cQ.NewPoly(2)
        plan.BfvSwitchKeys(c[2], pevk, p0, p1)
        w0, w1 = oplan.bfv_switch_keys(ct[2][1], evk.reshape(beta, 2, nq + np_, N))
        assert np.array_equal(p0.get()[1], w0) and np.array_equal(p1.get()[1], w1)
        # ctOut == ct0 (the reference relinearizes in place when the evaluator is handed the same ciphertext)
        plan.BfvRelinearize(c, pevk, (c[0], c[1]))
        assert np.array_equal(c[0].get(), out[0].get()) and np.array_equal(c[1].get(), out[1].get())


def test_bfv_relinearize_refuses_aliased_operands(gpu_pkg):
    _, Q, P, _ = gpu_pkg.params.bfv_moduli("PN12QP109")
    N = 1 << 10
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, 1)
    key = plan.NewSwitchingKey()
    a, b, c = cQ.NewPoly(1), cQ.NewPoly(1), cQ.NewPoly(1)
    with pytest.raises(gpu_pkg._native.LatticeRingError):
        plan.BfvRelinearize((a, b, c), key, (a, c))
    with pytest.raises(gpu_pkg._native.LatticeRingError):
        plan.BfvSwitchKeys(c, key, c, a)


# ---- small batches: grouped digit extensions, independent launches side by side (PlanFork in lr_abi_ckks.cpp) -------------------------
@pytest.mark.parametrize("logn,nq,np_,level,batch", [(14, 7, 3, 6, 1), (14, 7, 3, 5, 3), (15, 18, 3, 17, 1), (16, 6, 2, 5, 1), (16, 10, 4, 9, 2), (12, 10, 4, 9, 2)])
def test_small_batch_paths_give_the_same_bits(gpu_pkg, oracle, logn, nq, np_, level, batch, monkeypatch):
    """a lone plan at a small batch sends the digits' extensions out as one launch and, at N = 2^16, forks the digits' P-row transforms
    and ModDown's second component onto its auxiliary stream; LR_NO_FORK / LR_NO_EXT_GROUP keep everything in order, one launch per digit:
    all four combinations against the oracle, and the counters say which path ran"""
    import gc
    for env in ({}, {"LR_NO_FORK": "1"}, {"LR_NO_EXT_GROUP": "1"}, {"LR_NO_FORK": "1", "LR_NO_EXT_GROUP": "1"}):
        for k in ("LR_NO_FORK", "LR_NO_EXT_GROUP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        gc.collect()                                                    # plans of earlier tests are gone: this one is alone on the device
        N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, batch)
        mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, batch, seed=s).reshape(batch, level + 1, N)
        ops = [mk(31), mk(32), mk(33), mk(34)]
        P_ = lambda x: cQ.NewPolyLvl(level, batch).set(x)
        out = (cQ.NewPolyLvl(level, batch), cQ.NewPolyLvl(level, batch))
        for rep in range(2):                                            # the second call runs on the streams and events the first created
            plan.MulRelin(level, (P_(ops[0]), P_(ops[1])), (P_(ops[2]), P_(ops[3])), pevk, out)
        for b in range(batch):
            w0, w1 = oplan.mulrelin(level, np.stack([ops[0][b], ops[1][b]]), np.stack([ops[2][b], ops[3][b]]), evk)
            assert np.array_equal(out[0].get().reshape(batch, level + 1, N)[b], w0), (env, b)
            assert np.array_equal(out[1].get().reshape(batch, level + 1, N)[b], w1), (env, b)
        st = plan.Stats()
        assert (st["forks"] > 0) == ("LR_NO_FORK" not in env and logn == 16), (env, st)       # (forks pay at N = 2^16 only: lr_abi_ckks.cpp, PlanFork)
        full_digits = (level + 1) // np_
        assert (st["grouped_extensions"] > 0) == ("LR_NO_EXT_GROUP" not in env and full_digits > 1), (env, st)   # full digits share a shape
        del plan, pevk, out


def test_two_plans_alive_do_not_fork(gpu_pkg, oracle):
    """two evaluators on one device: their small launches already share the chip, neither forks (and a batcher's lanes follow their own count)"""
    import gc
    gc.collect()
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, 16, 6, 2, 1)
    other = gpu_pkg.ring.CkksPlan(cQ, cP, 1)
    level = 5
    mk = lambda s: cQ.NewPolyLvl(level, 1).set(gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 1, seed=s))
    out = (cQ.NewPolyLvl(level, 1), cQ.NewPolyLvl(level, 1))
    plan.MulRelin(level, (mk(1), mk(2)), (mk(3), mk(4)), pevk, out)
    assert plan.Stats()["forks"] == 0
    del other
    gc.collect()
    plan.MulRelin(level, (mk(1), mk(2)), (mk(3), mk(4)), pevk, out)
    assert plan.Stats()["forks"] > 0


@pytest.mark.parametrize("nq,np_,level,batch", [(18, 3, 17, 1), (18, 3, 12, 1), (7, 3, 6, 3), (5, 2, 4, 2)])
def test_mulrelin_2p15_split_paths(gpu_pkg, oracle, nq, np_, level, batch, monkeypatch):
    """N = 2^15 at a small batch: the key switch's transforms run as two 2^14 sub-blocks per limb -- forward ones behind the top stage the
    basis extensions apply (digits, ModDown), inverse ones in front of ntt_top_kernel; LR_NTT_SPLIT15=0 keeps one workgroup per transform,
    =1 splits whatever the size; LR_NO_EXTTOP leaves the forward top stage to ntt_top_kernel.  All against the oracle."""
    # (LR_NO_PAIR: ModDown's two components of a single ciphertext as two launches; by default one launch whose strides are the distances
    # between the components' operands -- negative when the second output was allocated first, as in the last round)
    # (LR_NO_INVTOP: the inverse transforms finish with their own last-stage pass instead of leaving it to the extension behind them)
    for env in ({}, {"LR_NTT_SPLIT15": "0"}, {"LR_NTT_SPLIT15": "1"}, {"LR_NO_EXTTOP": "1"}, {"LR_NTT_SPLIT15": "1", "LR_NO_FORK": "1", "LR_NO_EXT_GROUP": "1"},
                {"LR_NO_PAIR": "1"}, {"swap_outputs": "1"}, {"LR_NO_INVTOP": "1"}, {"LR_NO_INVTOP": "1", "LR_NO_EXT_CHUNKS": "1"}):
        for k in ("LR_NTT_SPLIT15", "LR_NO_EXTTOP", "LR_NO_FORK", "LR_NO_EXT_GROUP", "LR_NO_PAIR", "LR_NO_INVTOP", "LR_NO_EXT_CHUNKS"):
            monkeypatch.delenv(k, raising=False)
        swap = env.pop("swap_outputs", None) if isinstance(env, dict) else None
        env = dict(env)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, 15, nq, np_, batch)
        mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, batch, seed=s).reshape(batch, level + 1, N)
        ops = [mk(41), mk(42), mk(43), mk(44)]
        P_ = lambda x: cQ.NewPolyLvl(level, batch).set(x)
        out = (cQ.NewPolyLvl(level, batch), cQ.NewPolyLvl(level, batch))
        if swap:
            out = (out[1], out[0])
        plan.MulRelin(level, (P_(ops[0]), P_(ops[1])), (P_(ops[2]), P_(ops[3])), pevk, out)
        # the last launch is ModDown's transform with the epilogue: split only behind an extension that applied the top stage
        assert ("15h" in cQ.last_ntt_kernel()) == (env.get("LR_NTT_SPLIT15") != "0" and "LR_NO_EXTTOP" not in env), (env, cQ.last_ntt_kernel())
        for b in range(batch):
            w0, w1 = oplan.mulrelin(level, np.stack([ops[0][b], ops[1][b]]), np.stack([ops[2][b], ops[3][b]]), evk)
            assert np.array_equal(out[0].get().reshape(batch, level + 1, N)[b], w0), (env, b)
            assert np.array_equal(out[1].get().reshape(batch, level + 1, N)[b], w1), (env, b)
        plan.Rescale(out)                                  # (the rescale's fused-top form exists at 2^16 only: one workgroup per transform here)
        oc = oracle.Context(N, Q[:level + 1])
        for b in range(batch):
            w = oplan.mulrelin(level, np.stack([ops[0][b], ops[1][b]]), np.stack([ops[2][b], ops[3][b]]), evk)
            for k in range(2):
                assert np.array_equal(out[k].get().reshape(batch, level, N)[b], oc.rescale_op("oc_div_round_by_last_modulus_ntt", w[k])), (env, b, k)
        del plan, pevk, out


@pytest.mark.parametrize("logn,nq,np_,batch", [(15, 18, 3, 1), (14, 7, 3, 1), (16, 6, 2, 1), (12, 6, 2, 3)])
def test_rescale_of_both_components_in_one_set_of_launches(gpu_pkg, oracle, logn, nq, np_, batch, monkeypatch):
    """lr_ckks_rescale addresses the two components as one batch where base + p * stride reaches both (one poly each: the stride is
    their distance, whichever lies first; batches laid out back to back): same bits as one after the other (LR_RESCALE_UNPAIRED) and
    as the oracle, twice in a row (two levels)"""
    N = 1 << logn
    _, Qf, Pf = gpu_pkg.params.ckks_moduli("PN16QP1761" if logn == 16 else "PN15QP880")
    Q = list(Qf[:nq])
    oc = oracle.Context(N, Q)
    x = [gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=70 + k).reshape(batch, nq, N) for k in range(2)]
    want = [[oc.rescale_op("oc_div_round_by_last_modulus_ntt", x[k][b]) for b in range(batch)] for k in range(2)]
    oc2 = oracle.Context(N, Q[:-1])
    want2 = [[oc2.rescale_op("oc_div_round_by_last_modulus_ntt", want[k][b]) for b in range(batch)] for k in range(2)]
    for env, order in (({}, (0, 1)), ({}, (1, 0)), ({"LR_RESCALE_UNPAIRED": "1"}, (0, 1))):
        monkeypatch.delenv("LR_RESCALE_UNPAIRED", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cQ = gpu_pkg.ring.NewContextWithParams(N, Q)
        cP = gpu_pkg.ring.NewContextWithParams(N, list(Pf[:np_]))
        plan = gpu_pkg.ring.CkksPlan(cQ, cP, batch)
        polys = [None, None]
        for k in order:                                     # allocation order decides which component lies first in memory
            polys[k] = cQ.NewPoly(batch).set(x[k])
        whole = None
        if batch > 1 and not env:
            # back to back: the two components are the halves of one allocation (wrapped without a copy)
            whole = cQ.NewPoly(2 * batch).set(np.concatenate([x[order[0]], x[order[1]]]))
            base = whole.device_ptr
            polys[order[0]] = gpu_pkg.ring.Poly.wrap(cQ, base, nq, batch)
            polys[order[1]] = gpu_pkg.ring.Poly.wrap(cQ, base + 8 * batch * nq * N, nq, batch)
        ct = (polys[0], polys[1])
        plan.Rescale(ct)
        for k in range(2):
            got = ct[k].get().reshape(batch, nq - 1, N)
            for b in range(batch):
                assert np.array_equal(got[b], want[k][b]), (env, order, k, b)
        plan.Rescale(ct)
        for k in range(2):
            got = ct[k].get().reshape(batch, nq - 2, N)
            for b in range(batch):
                assert np.array_equal(got[b], want2[k][b]), (env, order, k, b)


@pytest.mark.parametrize("logn,nq,np_,level,k", [(14, 7, 3, 6, 3), (15, 18, 3, 17, 2), (15, 7, 3, 4, 9), (16, 6, 2, 5, 7), (12, 6, 2, 5, 1)])
def test_rotate_columns_of_a_single_ciphertext(gpu_pkg, oracle, logn, nq, np_, level, k, monkeypatch):
    """one ciphertext per call: the two permutations and ModDown's two transforms each go out as one launch whose strides are the
    distances between the components (the second component's addend is the row of zeros); LR_NO_PAIR keeps separate launches; both
    against the oracle, also with the outputs allocated in the other order and in place"""
    for env, swap in (({}, False), ({}, True), ({"LR_NO_PAIR": "1"}, False)):
        monkeypatch.delenv("LR_NO_PAIR", raising=False)
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, 1)
        mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 1, seed=s).reshape(1, level + 1, N)
        a0, a1 = mk(31), mk(32)
        P_ = lambda x: cQ.NewPolyLvl(level, 1).set(x)
        for gen in (_galois(gpu_pkg, N, k), 2 * N - 1):
            ct = (P_(a1), P_(a0))[::-1] if swap else (P_(a0), P_(a1))
            out = (cQ.NewPolyLvl(level, 1), cQ.NewPolyLvl(level, 1))
            if swap:
                out = out[::-1]
            plan.PermuteNTT(level, ct, gen, pevk, out)
            want = oplan.permute_ntt(level, np.stack([a0[0], a1[0]]), gen, evk)
            for c in range(2):
                assert np.array_equal(out[c].get().reshape(level + 1, N), want[c]), (env, swap, gen, c)
            plan.PermuteNTT(level, ct, gen, pevk, ct)       # ctOut == ct0
            for c in range(2):
                assert np.array_equal(ct[c].get().reshape(level + 1, N), want[c]), (env, swap, gen, c)
        del plan, pevk


@pytest.mark.parametrize("logn,nq,np_,level,batch", [(15, 18, 3, 17, 1), (16, 10, 4, 9, 1), (14, 7, 3, 5, 1), (13, 9, 4, 8, 2)])
def test_small_batch_extensions_in_column_ranges(gpu_pkg, oracle, logn, nq, np_, level, batch, monkeypatch):
    """the basis extensions of a small batch go out as one grouped launch over disjoint column ranges (grid z = range; the digits
    of a key switch that do not share a shape with others and ModDown's extension); LR_NO_EXT_CHUNKS keeps one launch over all columns"""
    for env in ({}, {"LR_NO_EXT_CHUNKS": "1"}):
        monkeypatch.delenv("LR_NO_EXT_CHUNKS", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, batch)
        mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, batch, seed=s).reshape(batch, level + 1, N)
        ops = [mk(51), mk(52), mk(53), mk(54)]
        P_ = lambda x: cQ.NewPolyLvl(level, batch).set(x)
        out = (cQ.NewPolyLvl(level, batch), cQ.NewPolyLvl(level, batch))
        plan.MulRelin(level, (P_(ops[0]), P_(ops[1])), (P_(ops[2]), P_(ops[3])), pevk, out)
        for b in range(batch):
            w0, w1 = oplan.mulrelin(level, np.stack([ops[0][b], ops[1][b]]), np.stack([ops[2][b], ops[3][b]]), evk)
            assert np.array_equal(out[0].get().reshape(batch, level + 1, N)[b], w0), (env, b)
            assert np.array_equal(out[1].get().reshape(batch, level + 1, N)[b], w1), (env, b)
        del plan, pevk, out


@pytest.mark.parametrize("name,logn", [("PN13QP218", 11), ("PN14QP438", 14), ("PN15QP880", 15)])
def test_bfv_relinearize_of_a_single_ciphertext(gpu_pkg, oracle, name, logn, monkeypatch):
    """one degree-2 ciphertext: the coefficient-domain tail runs its two accumulators as one batch (inverse transforms, ModDown's extension
    in place) and the two additions go out as one launch; LR_NO_PAIR keeps them apart; switchKeys alone with user polys at any distance;
    in place; against the oracle"""
    _, Q, P, _ = gpu_pkg.params.bfv_moduli(name)
    N = 1 << logn
    ring = gpu_pkg.ring
    nq, np_ = len(Q), len(P)
    beta = -(-nq // np_)
    evk = gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=41)
    ct = [gpu_pkg.sampling.uniform_poly(Q, N, 1, seed=60 + k).reshape(1, nq, N) for k in range(3)]
    oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    want = oplan.bfv_relinearize(np.stack([ct[0][0], ct[1][0], ct[2][0]]), evk.reshape(beta, 2, nq + np_, N))
    w0, w1 = oplan.bfv_switch_keys(ct[2][0], evk.reshape(beta, 2, nq + np_, N))
    for env, swap in (({}, False), ({}, True), ({"LR_NO_PAIR": "1"}, False), ({"LR_NO_EPILOGUE": "1"}, False)):
        for k in ("LR_NO_PAIR", "LR_NO_EPILOGUE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
        plan = ring.CkksPlan(cQ, cP, 1)
        pevk = plan.NewSwitchingKey().set(evk)
        c = [cQ.NewPoly(1).set(x) for x in ct]
        out = (cQ.NewPoly(1), cQ.NewPoly(1))
        if swap:
            out = out[::-1]
        plan.BfvRelinearize(c, pevk, out)
        for k in range(2):
            assert np.array_equal(out[k].get().reshape(nq, N), want[k]), (env, swap, k)
        p = (cQ.NewPoly(1), cQ.NewPoly(1))
        if swap:
            p = p[::-1]
        plan.BfvSwitchKeys(c[2], pevk, p[0], p[1])
        assert np.array_equal(p[0].get().reshape(nq, N), w0) and np.array_equal(p[1].get().reshape(nq, N), w1), (env, swap)
        plan.BfvRelinearize(c, pevk, (c[0], c[1]))
        for k in range(2):
            assert np.array_equal(c[k].get().reshape(nq, N), want[k]), (env, swap, "in place", k)
        del plan, pevk


@pytest.mark.parametrize("name", ["PN12QP109", "PN13QP218", "PN14QP438", "PN15QP880", "PN16QP1761"])
def test_one_ciphertext_on_every_default_parameter_set(gpu_pkg, oracle, name):
    """ckks.DefaultParams at full size, ONE ciphertext per call -- the reference's calling shape, which takes the small-batch paths (grouped
    and column-chunked extensions, sub-block transforms, lazy inverse outputs, the two components in one launch): MulRelin at the top
    level and two levels down (a partial last digit where the set has one), Rescale, RotateColumns and Conjugate, each against the oracle"""
    N, Q, P = gpu_pkg.params.ckks_moduli(name)
    Q, P = list(Q), list(P)
    nq, np_ = len(Q), len(P)
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, 1)
    oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    beta = -(-nq // np_)
    evk = gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=77)
    pevk = plan.NewSwitchingKey().set(evk)
    evk = evk.reshape(beta, 2, nq + np_, N)
    for level in sorted({nq - 1, max(1, nq - 3)}, reverse=True):
        mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 1, seed=s).reshape(1, level + 1, N)
        ops = [mk(81), mk(82), mk(83), mk(84)]
        P_ = lambda x: cQ.NewPolyLvl(level, 1).set(x)
        out = (cQ.NewPolyLvl(level, 1), cQ.NewPolyLvl(level, 1))
        plan.MulRelin(level, (P_(ops[0]), P_(ops[1])), (P_(ops[2]), P_(ops[3])), pevk, out)
        want = oplan.mulrelin(level, np.stack([ops[0][0], ops[1][0]]), np.stack([ops[2][0], ops[3][0]]), evk)
        for k in range(2):
            assert np.array_equal(out[k].get().reshape(level + 1, N), want[k]), (name, level, "mulrelin", k)
        for gen in (pow(5, 1, 2 * N), 2 * N - 1):
            rot = (cQ.NewPolyLvl(level, 1), cQ.NewPolyLvl(level, 1))
            plan.PermuteNTT(level, (P_(ops[0]), P_(ops[1])), gen, pevk, rot)
            wr = oplan.permute_ntt(level, np.stack([ops[0][0], ops[1][0]]), gen, evk)
            for k in range(2):
                assert np.array_equal(rot[k].get().reshape(level + 1, N), wr[k]), (name, level, "rotate", gen, k)
        plan.Rescale(out)
        oc = oracle.Context(N, Q[:level + 1])
        for k in range(2):
            assert np.array_equal(out[k].get().reshape(level, N), oc.rescale_op("oc_div_round_by_last_modulus_ntt", want[k])), (name, level, "rescale", k)


# ---- bfv rotations end to end (bfv/evaluator.go:579-735) ------------------------------------------------------------------------
@pytest.mark.parametrize("name,logn,batch", [("PN12QP109", 12, 2), ("PN13QP218", 11, 3), ("PN14QP438", 14, 2), ("PN14QP438", 14, 1), ("PN15QP880", 13, 1)])
def test_bfv_rotation_sequence(gpu_pkg, oracle, name, logn, batch, monkeypatch):
    """bfv.evaluator.permute (bfv/evaluator.go:711-735) = the body of RotateRows and RotateColumns: Context.Permute of both components,
    switchKeys of the second, Add, Copy -- the composed sequence through ONE C-ABI call (lr_bfv_rotate), against the oracle's restatement
    (oc_bfv_permute) at PN14QP438 full size (BASELINE config 4's set) and the other BFV sets; column generator 5^k, row generator 2N - 1;
    in place (the reference's polypool branch, :717-721); a batch of one with its two components in one launch and with LR_NO_PAIR"""
    _, Q, P, _ = gpu_pkg.params.bfv_moduli(name)
    N = 1 << logn
    ring = gpu_pkg.ring
    nq, np_ = len(Q), len(P)
    beta = -(-nq // np_)
    evk = gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=43)
    ct = [gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=70 + k).reshape(batch, nq, N) for k in range(2)]
    ct[0][0, :, 3] = 0                                      # a zero coefficient whose sign flips becomes q (ring_galois.go:121)
    ct[1][0, :, 5] = 0
    oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    gens = [pow(5, 1, 2 * N), pow(5, 37, 2 * N), 2 * N - 1]        # GaloisGen^k (bfv/bfv.go galElRotColLeft), galElRotRow
    for env in ({}, {"LR_NO_PAIR": "1"}) if batch == 1 else ({},):
        monkeypatch.delenv("LR_NO_PAIR", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
        plan = ring.CkksPlan(cQ, cP, batch)
        pevk = plan.NewSwitchingKey().set(evk)
        for gen in gens:
            want = [oplan.bfv_permute(np.stack([ct[0][b], ct[1][b]]), gen, evk.reshape(beta, 2, nq + np_, N)) for b in range(batch)]
            c = (cQ.NewPoly(batch).set(ct[0]), cQ.NewPoly(batch).set(ct[1]))
            out = (cQ.NewPoly(batch), cQ.NewPoly(batch))
            plan.BfvPermute(c, gen, pevk, out)
            for b in range(batch):
                for k in range(2):
                    assert np.array_equal(out[k].get().reshape(batch, nq, N)[b], want[b][k]), (env, gen, b, k)
            assert np.array_equal(c[0].get().reshape(batch, nq, N), ct[0])          # the input is left alone
            plan.BfvPermute(c, gen, pevk, c)                                          # ct0 == ctOut
            for k in range(2):
                assert np.array_equal(c[k].get(), out[k].get()), (env, gen, "in place", k)
        del plan, pevk


def test_bfv_rotate_columns_pow2_chain_and_argument_errors(gpu_pkg, oracle):
    """rotateColumnsPow2 (bfv/evaluator.go:636-662): k = 5 = 1 + 4 as the chain of the power-of-two rotations with the generator squared at
    every step, through the host mirror, against the oracle applying bfv_permute step by step; and the entry point's argument checks"""
    _, Q, P, _ = gpu_pkg.params.bfv_moduli("PN13QP218")
    N = 1 << 12
    ring = gpu_pkg.ring
    nq, np_ = len(Q), len(P)
    beta = -(-nq // np_)
    keys_h = {1 << i: gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=90 + i) for i in range(3)}
    ct = [gpu_pkg.sampling.uniform_poly(Q, N, 1, seed=95 + k).reshape(nq, N) for k in range(2)]
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, 1)
    keys = {i: plan.NewSwitchingKey().set(k) for i, k in keys_h.items()}
    c = (cQ.NewPoly(1).set(ct[0]), cQ.NewPoly(1).set(ct[1]))
    out = (cQ.NewPoly(1), cQ.NewPoly(1))
    plan.BfvRotateColumnsPow2(c, 5, 5, keys, out)
    oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    want = np.stack(ct)
    gen, idx, k = 5, 1, 5
    while k:
        if k & 1:
            want = oplan.bfv_permute(want, gen, keys_h[idx].reshape(beta, 2, nq + np_, N))
        gen = (gen * gen) & (2 * N - 1)
        idx <<= 1
        k >>= 1
    for j in range(2):
        assert np.array_equal(out[j].get().reshape(nq, N), want[j]), j
    err = gpu_pkg._native.LatticeRingError
    with pytest.raises(err):
        plan.BfvPermute(c, 5, keys[1], (out[0], out[0]))                              # one output poly twice
    with pytest.raises(err):
        plan.BfvPermute((cQ.NewPoly(2), cQ.NewPoly(2)), 5, keys[1], (cQ.NewPoly(2), cQ.NewPoly(2)))   # batch beyond the plan's max_batch
    with pytest.raises(err):
        plan.BfvPermute((cQ.NewPolyLvl(nq - 2, 1), c[1]), 5, keys[1], out)            # a component with fewer limbs than Q


@pytest.mark.parametrize("degree", [1, 2, 7, 8, 9])
def test_decrypt_fused_pass_against_the_call_by_call_form(gpu_pkg, oracle, degree, monkeypatch):
    """lr_ckks_decrypt as ONE pass (horner_kernel: every component and the key read once) against the oracle and against the call-by-call
    form (Copy, MulCoeffsMontgomery, Add, Reduce launches: LR_NO_EPILOGUE, and degree 9 which the fused kernel does not take); operands
    anywhere in [0, 2^64) (the reference's Reduce cadence decides what comes out); the plaintext in place of the top component (a plaintext that IS a
    lower component keeps the reference's call-by-call order, whose Copy overwrites that component first)"""
    logn, nq, np_, batch = 11, 4, 2, 3
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, batch)
    level = nq - 1
    sk = gpu_pkg.sampling.uniform_poly(Q, N, 1, seed=1).reshape(1, nq, N)
    ct = gpu_pkg.sampling.random_u64((degree + 1, batch, nq, N), seed=degree)        # full 64-bit words, as NewCiphertextRandom fills them
    ct[0] = gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=50).reshape(batch, nq, N)
    want = [oplan.decrypt(level, ct[:, b], sk[0]) for b in range(batch)]
    got = []
    for env in ({}, {"LR_NO_EPILOGUE": "1"}):
        monkeypatch.delenv("LR_NO_EPILOGUE", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cq = gpu_pkg.ring.NewContextWithParams(N, Q)
        pl = gpu_pkg.ring.CkksPlan(cq, gpu_pkg.ring.NewContextWithParams(N, P), batch)
        psk = cq.NewPoly(1).set(sk)
        polys = [cq.NewPoly(batch).set(ct[i]) for i in range(degree + 1)]
        out = cq.NewPoly(batch)
        pl.Decrypt(level, polys, psk, out)
        got.append(out.get().reshape(batch, nq, N))
        for b in range(batch):
            assert np.array_equal(got[-1][b], want[b]), (env, b)
        pl.Decrypt(level, polys, psk, polys[degree])                                    # plaintext.value is the top component
        assert np.array_equal(polys[degree].get().reshape(batch, nq, N), got[-1]), env
    assert np.array_equal(got[0], got[1])


@pytest.mark.parametrize("env", [{}, {"LR_NO_EPILOGUE": "1"}, {"LR_KEYMAC_NARROW": "1"}, {"LR_NO_PAIR": "1", "LR_KEYMAC_NARROW": "1"}])
@pytest.mark.parametrize("logn,nq,np_,level,batch", [(12, 6, 2, 5, 3), (15, 18, 3, 17, 1), (13, 7, 3, 3, 2)])
def test_rotate_hoisted_reads_the_digits_through_the_permutation(gpu_pkg, oracle, logn, nq, np_, level, batch, env, monkeypatch):
    """round 4: the Galois automorphism of the digits (ckks/evaluator.go:1346-1347) rides on the key inner product's loads (KeyMacLaunch::
    perm_gen) instead of a pass that writes permuted copies of every digit; LR_NO_EPILOGUE keeps the copies, LR_KEYMAC_NARROW takes the
    per-term kernel through the same loads; every way against the oracle's switchKeyHoisted, PN15QP880 at full size with one ciphertext
    (the paired Q + P launch) included"""
    for k in ("LR_NO_EPILOGUE", "LR_KEYMAC_NARROW", "LR_NO_PAIR"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    N, Q, P, cQ, cP, plan, oplan, evk0, pevk0 = _ckks(gpu_pkg, oracle, logn, nq, np_, batch)
    beta = -(-nq // np_)
    gens = [_galois(gpu_pkg, N, 1), _galois(gpu_pkg, N, 7), 2 * N - 1]                 # two column rotations and the conjugation
    evks = [gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=400 + k) for k in range(3)]
    pevks = [plan.NewSwitchingKey().set(e) for e in evks]
    mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, batch, seed=s).reshape(batch, level + 1, N)
    a0, a1 = mk(43), mk(44)
    ct = (cQ.NewPolyLvl(level, batch).set(a0), cQ.NewPolyLvl(level, batch).set(a1))
    outs = [(cQ.NewPolyLvl(level, batch), cQ.NewPolyLvl(level, batch)) for _ in gens]
    plan.RotateHoisted(level, ct, gens, pevks, outs)
    for b in range(batch):
        want = oplan.rotate_hoisted(level, np.stack([a0[b], a1[b]]), gens, [e.reshape(beta, 2, nq + np_, N) for e in evks])
        for r in range(len(gens)):
            for k in range(2):
                assert np.array_equal(outs[r][k].get().reshape(batch, level + 1, N)[b], want[r][k]), (env, r, b, k)


@pytest.mark.parametrize("env", [{}, {"LR_NO_EPILOGUE": "1"}, {"LR_EXT_NARROW": "1"}])
@pytest.mark.parametrize("logn,nq,np_,level,batch", [(12, 4, 2, 3, 3), (15, 18, 3, 17, 2), (13, 6, 3, 3, 2), (14, 7, 3, 6, 1)])
def test_encrypt_pk_fused_steps_against_the_call_by_call_form(gpu_pkg, oracle, logn, nq, np_, level, batch, env, monkeypatch):
    """round 4: pkEncryptor.encrypt's two products with the public key in one pass over u (mul2_kernel) and the Q rows' share of SampleAndAdd
    inside the ModDown's extension epilogue (ExtSegment::epi_x2), at the top level; LR_NO_EPILOGUE keeps one launch per Context call, a
    level below the top keeps the separate additions (the reference's ModDownPQ reads rows of Q as its P part there), LR_EXT_NARROW takes
    an extension kernel without the epilogue; every way against the oracle, residues equal to q included"""
    for k in ("LR_NO_EPILOGUE", "LR_EXT_NARROW"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    N, Q, P, cQ, cP, plan, oplan, evk, pevk = _ckks(gpu_pkg, oracle, logn, nq, np_, batch)
    QP = Q + P
    ocQP = oracle.Context(N, QP)
    uni = lambda s, n: gpu_pkg.sampling.uniform_poly(QP, N, n, seed=s).reshape(n, nq + np_, N)
    u, e0, e1 = uni(11, batch), uni(12, batch), uni(13, batch)
    for i, q in enumerate(QP):
        e0[0, i, 0] = q
        e1[batch - 1, i, N - 1] = q
    pk0, pk1 = uni(14, 1), uni(15, 1)
    pt = gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, batch, seed=16).reshape(batch, level + 1, N)
    QPpoly = lambda x: gpu_pkg.ring.Poly(cQ, nq + np_, x.shape[0]).set(x)
    ct = (cQ.NewPolyLvl(level, batch), cQ.NewPolyLvl(level, batch))
    plan.EncryptPk(level, QPpoly(u), (QPpoly(pk0), QPpoly(pk1)), (QPpoly(e0), QPpoly(e1)), cQ.NewPolyLvl(level, batch).set(pt), ct)
    for b in range(batch):
        want = oplan.encrypt_pk(ocQP, level, u[b], pk0[0], pk1[0], e0[b], e1[b], pt[b])
        for k in range(2):
            assert np.array_equal(ct[k].get().reshape(batch, level + 1, N)[b], want[k]), (env, b, k)

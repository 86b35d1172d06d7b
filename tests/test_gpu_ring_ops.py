"""Parity of the coefficient-wise family, basis extension, decomposer and RNS rescale (HIP path through
the C ABI) against the CPU oracle on the same seeded inputs.  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THREE_OPERAND = ["ADD", "ADD_NOMOD", "SUB", "SUB_NOMOD", "MUL_COEFFS", "MUL_COEFFS_AND_ADD",
                 "MUL_COEFFS_AND_ADD_NOMOD", "MUL_COEFFS_CONSTANT", "MUL_MONT", "MUL_MONT_AND_ADD",
                 "MUL_MONT_AND_ADD_NOMOD", "MUL_MONT_CONSTANT_AND_ADD_NOMOD", "MUL_MONT_AND_SUB",
                 "MUL_MONT_AND_SUB_NOMOD", "MUL_MONT_CONSTANT"]
TWO_OPERAND = ["NEG", "REDUCE", "MFORM", "INV_MFORM", "COPY"]


def _setup(gpu_pkg, oracle, logn=12, limbs=3, batch=2, seed=1):
    N = 1 << logn
    moduli = list(gpu_pkg.params.Qi60()[-limbs:])
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    oc = oracle.Context(N, moduli)
    mk = lambda s: gpu_pkg.sampling.uniform_poly(moduli, N, batch, seed=seed * 10 + s)
    return N, moduli, ctx, oc, mk


@pytest.mark.parametrize("op", THREE_OPERAND)
def test_three_operand_ops(gpu_pkg, oracle, op):
    N, moduli, ctx, oc, mk = _setup(gpu_pkg, oracle)
    a, b, c = mk(1), mk(2), mk(3)
    pa, pb, pc = ctx.NewPoly(2).set(a), ctx.NewPoly(2).set(b), ctx.NewPoly(2).set(c)
    ctx._ew(op, len(moduli) - 1, pa, pb, pc)
    got = pc.get()
    for i in range(2):
        assert np.array_equal(got[i], oc.ewise(op, a[i], b[i], out=c[i])), op


@pytest.mark.parametrize("op", TWO_OPERAND)
def test_two_operand_ops(gpu_pkg, oracle, op):
    N, moduli, ctx, oc, mk = _setup(gpu_pkg, oracle)
    a = gpu_pkg.sampling.random_u64((2, len(moduli), N), seed=5) if op == "REDUCE" else mk(1)
    pa, pc = ctx.NewPoly(2).set(a), ctx.NewPoly(2)
    ctx._ew(op, len(moduli) - 1, pa, None, pc)
    got = pc.get()
    for i in range(2):
        assert np.array_equal(got[i], oc.ewise(op, a[i])), op


def test_level_and_broadcast(gpu_pkg, oracle):
    """...Lvl variants leave the upper limbs alone; a batch-1 operand is broadcast."""
    N, moduli, ctx, oc, mk = _setup(gpu_pkg, oracle, limbs=4, batch=3)
    a, c = mk(1), mk(3)
    key = mk(2)[:1]
    pa, pk, pc = ctx.NewPoly(3).set(a), ctx.NewPoly(1).set(key[0]), ctx.NewPoly(3).set(c)
    ctx.MulCoeffsMontgomeryAndAddNoModLvl(1, pk, pa, pc)
    got = pc.get()
    for i in range(3):
        want = oc.ewise("MUL_MONT_AND_ADD_NOMOD", key[0], a[i], out=c[i], level=1)
        assert np.array_equal(got[i, :2], want[:2])
        assert np.array_equal(got[i, 2:], c[i, 2:])


def test_scalar_ops(gpu_pkg, oracle):
    N, moduli, ctx, oc, mk = _setup(gpu_pkg, oracle)
    a = mk(1)
    pa, pc = ctx.NewPoly(2).set(a), ctx.NewPoly(2)
    ctx.MulScalar(pa, 0xFFFFFFFFFFFFFFF1, pc)
    assert np.array_equal(pc.get()[1], oc.ewise("MUL_SCALAR", a[1], scalars=[0xFFFFFFFFFFFFFFF1]))
    big = (1 << 150) + 977
    ctx.MulScalarBigint(pa, big, pc)
    want = np.array([[int(v) * big % m for v in a[0, i]] for i, m in enumerate(moduli)], dtype=np.uint64)
    assert np.array_equal(pc.get()[0], want)
    # AddScalarBigint / SubScalarBigint write into their FIRST argument (ring/ring.go:482,505)
    ctx.AddScalarBigint(pa, big, pc)
    assert np.array_equal(pa.get()[0], oc.ewise("ADD_SCALAR_LIMBS", a[0], scalars=[big % m for m in moduli]))
    ctx.SubScalarBigint(pa, big, pc)
    assert np.array_equal(pa.get(), a)
    ctx.MulByPow2(pa, 13, pc)
    assert np.array_equal(pc.get()[0], oc.ewise("MUL_BY_POW2", a[0], scalars=[13]))


def test_mulpoly_vs_schoolbook_small(gpu_pkg):
    # testMulPoly, ring/ring_test.go:503-548
    N, moduli = 64, [576460752303439873, 576460752303702017]
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    a = gpu_pkg.sampling.uniform_poly(moduli, N, 1, seed=5)[0]
    b = gpu_pkg.sampling.uniform_poly(moduli, N, 1, seed=6)[0]
    pa, pb, pc = ctx.NewPoly().set(a), ctx.NewPoly().set(b), ctx.NewPoly()
    ctx.MForm(pa, pa)
    ctx.MulPolyMontgomery(pa, pb, pc)
    got = pc.get()
    for i, q in enumerate(moduli):
        want = [0] * N
        for x in range(N):
            for y in range(N):
                k = x + y
                t = int(a[i, x]) * int(b[i, y])
                if k < N:
                    want[k] = (want[k] + t) % q
                else:
                    want[k - N] = (want[k - N] - t) % q
        assert [int(v) for v in got[i]] == want


@pytest.mark.parametrize("nq,np_,logn", [(2, 2, 12), (16, 16, 12), (3, 18, 10), (6, 6, 13), (18, 3, 11)])
def test_modup(gpu_pkg, oracle, nq, np_, logn):
    N = 1 << logn
    Q, P = list(gpu_pkg.params.Qi60()[-nq:]), list(gpu_pkg.params.Pi60()[-np_:])
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    be = ring.NewFastBasisExtender(cQ, cP)
    obe = oracle.BasisExtender(oracle.Context(N, Q), oracle.Context(N, P))
    assert np.array_equal(be.ModDownParamsPQ(), obe.moddown_params_pq)
    assert np.array_equal(be.ModDownParamsQP(), obe.moddown_params_qp)
    x = gpu_pkg.sampling.uniform_poly(Q, N, 2, seed=nq)
    px, pp = cQ.NewPoly(2).set(x), cP.NewPoly(2)
    be.ModUpSplitQP(nq - 1, px, pp)
    got = pp.get()
    for b in range(2):
        assert np.array_equal(got[b], obe.modup_split_qp(nq - 1, x[b]))
    y = gpu_pkg.sampling.uniform_poly(P, N, 2, seed=np_ + 50)
    py, pq = cP.NewPoly(2).set(y), cQ.NewPoly(2)
    be.ModUpSplitPQ(np_ - 1, py, pq)
    got = pq.get()
    for b in range(2):
        assert np.array_equal(got[b], obe.modup_split_pq(np_ - 1, y[b]))
    if nq > 2:   # a lower level uses fewer input limbs with the full-basis tables (reference behaviour)
        be.ModUpSplitQP(nq - 2, px, pp)
        assert np.array_equal(pp.get()[0], obe.modup_split_qp(nq - 2, x[0]))


@pytest.mark.parametrize("nq,np_,logn,above", [(17, 5, 6, False), (21, 2, 6, False), (32, 4, 8, False), (33, 2, 6, False), (40, 3, 5, False),
                                                (9, 4, 7, True), (20, 3, 6, True)])
def test_modup_long_sums(gpu_pkg, oracle, nq, np_, logn, above):
    """the 128-bit column sums of ext_wide_kernel at every group structure: one group (n * q < 2^64), groups of 16 (60-bit inputs,
    more than 16 limbs; one coefficient per thread beyond 20 limbs), groups of 8 (inputs just above 2^60: bfv's QMul primes)"""
    N = 1 << logn
    P_ = gpu_pkg.params
    if above:
        mods = P_.GenerateNTTPrimes(60, max(logn, 4), nq + np_)
        assert min(mods) > (1 << 60)
        Q, P = mods[:nq], mods[nq:]
    else:
        Q, P = list(P_.Qi60()[-nq:]), list(P_.Pi60()[-np_:])
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    be = ring.NewFastBasisExtender(cQ, cP)
    obe = oracle.BasisExtender(oracle.Context(N, Q), oracle.Context(N, P))
    x = gpu_pkg.sampling.uniform_poly(Q, N, 2, seed=nq)
    for i, q in enumerate(Q):
        x[0, i, 0] = q - 1                      # the largest residues in every limb at once: the sums at their maximum
        x[1, i, 1] = 0
    px, pp = cQ.NewPoly(2).set(x), cP.NewPoly(2)
    for level in (nq - 1, nq // 2):
        be.ModUpSplitQP(level, px, pp)
        got = pp.get()
        for b in range(2):
            assert np.array_equal(got[b], obe.modup_split_qp(level, x[b])), (level, b)


@pytest.mark.parametrize("nq,np_,level", [(4, 2, 3), (4, 2, 1), (18, 3, 17)])
def test_moddown_variants(gpu_pkg, oracle, nq, np_, level):
    N = 1 << 11
    Q, P = list(gpu_pkg.params.Qi60()[-nq:]), list(gpu_pkg.params.Pi60()[-np_:])
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    be = ring.NewFastBasisExtender(cQ, cP)
    ocQ, ocP = oracle.Context(N, Q), oracle.Context(N, P)
    obe = oracle.BasisExtender(ocQ, ocP)
    xq = gpu_pkg.sampling.uniform_poly(Q, N, 2, seed=1)
    xp = gpu_pkg.sampling.uniform_poly(P, N, 2, seed=2)
    pq, pp, out = cQ.NewPoly(2).set(xq), cP.NewPoly(2).set(xp), cQ.NewPolyLvl(level, 2)
    be.ModDownSplitedPQ(level, pq, pp, out)
    for b in range(2):
        assert np.array_equal(out.get()[b], obe.moddown_split_pq(level, xq[b], xp[b]))
    be.ModDownSplitedNTTPQ(level, pq, pp, out)
    for b in range(2):
        assert np.array_equal(out.get()[b], obe.moddown_split_ntt_pq(level, xq[b], xp[b]))
        assert np.array_equal(pp.get()[b], ocP.intt(xp[b]))       # p1P is left in the coefficient domain
    # joined forms
    joined = np.concatenate([xq, xp], axis=1)
    pj = ring.Poly(cQ, nq + np_, 2).set(joined)
    be.ModDownNTTPQ(level, pj, out)
    for b in range(2):
        assert np.array_equal(out.get()[b], obe.moddown_ntt_pq(level, joined[b]))
    joined_lvl = np.concatenate([xq[:, :level + 1], xp], axis=1)
    pjl = ring.Poly(cQ, level + 1 + np_, 2).set(joined_lvl)
    be.ModDownPQ(level, pjl, out)
    for b in range(2):
        assert np.array_equal(out.get()[b], obe.moddown_pq(level, joined_lvl[b]))
    # divide by Q, result over P (bfv/evaluator.go:450)
    outp = cP.NewPoly(2)
    be.ModDownSplitedQP(level, np_ - 1, pq, cP.NewPoly(2).set(xp), outp)
    for b in range(2):
        assert np.array_equal(outp.get()[b], obe.moddown_split_qp(level, np_ - 1, xq[b], xp[b]))


@pytest.mark.parametrize("nq,np_,level", [(4, 2, 3), (18, 3, 17)])
def test_moddown_variants_with_separate_passes(gpu_pkg, oracle, nq, np_, level, monkeypatch):
    """the coefficient-domain ModDowns write MRed(x + (q - ext), c) from the extension kernel's stores by default; LR_NO_EPILOGUE (read
    when the contexts are created) keeps the extension -> pool -> subtract-multiply passes"""
    monkeypatch.setenv("LR_NO_EPILOGUE", "1")
    test_moddown_variants(gpu_pkg, oracle, nq, np_, level)



@pytest.mark.parametrize("nq,np_,level", [(6, 2, 5), (6, 2, 4), (6, 2, 2), (7, 3, 6), (7, 3, 3), (5, 1, 4), (18, 3, 17),
                                          (18, 3, 9)])
def test_decomposer(gpu_pkg, oracle, nq, np_, level):
    N = 1 << 10
    Q, P = list(gpu_pkg.params.Qi60()[-nq:]), list(gpu_pkg.params.Pi60()[-np_:])
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    dec = ring.NewDecomposer(cQ, cP)
    odec = oracle.Decomposer(Q, P)
    x = gpu_pkg.sampling.uniform_poly(Q, N, 2, seed=level)
    px = cQ.NewPoly(2).set(x)
    beta = -(-(level + 1) // np_)
    for crt in range(beta):
        oq, op = cQ.NewPolyLvl(level, 2), cP.NewPoly(2)
        dec.DecomposeAndSplit(level, crt, px, oq, op)
        oj = ring.Poly(cQ, level + 1 + np_, 2)
        dec.Decompose(level, crt, px, oj)
        for b in range(2):
            wq, wp = odec.decompose_and_split(level, crt, x[b])
            assert np.array_equal(oq.get()[b], wq), (crt, b)
            assert np.array_equal(op.get()[b], wp), (crt, b)
            assert np.array_equal(oj.get()[b], odec.decompose(level, crt, x[b])), (crt, b)


@pytest.mark.parametrize("name", ["DivFloorByLastModulusNTT", "DivFloorByLastModulus", "DivRoundByLastModulusNTT",
                                  "DivRoundByLastModulus"])
@pytest.mark.parametrize("logn", [10, 12])
def test_rescale(gpu_pkg, oracle, name, logn):
    N = 1 << logn
    _, Q, _ = gpu_pkg.params.ckks_moduli("PN15QP880")      # 50-bit q0, 40-bit q1..: mixed sizes (values >= q_i occur)
    Q = Q[:5]
    ctx = gpu_pkg.ring.NewContextWithParams(N, Q)
    oc = oracle.Context(N, Q)
    x = gpu_pkg.sampling.uniform_poly(Q, N, 2, seed=9)
    p = ctx.NewPoly(2).set(x)
    getattr(ctx, name)(p)
    oname = {"DivFloorByLastModulusNTT": "oc_div_floor_by_last_modulus_ntt", "DivFloorByLastModulus": "oc_div_floor_by_last_modulus",
             "DivRoundByLastModulusNTT": "oc_div_round_by_last_modulus_ntt", "DivRoundByLastModulus": "oc_div_round_by_last_modulus"}[name]
    assert p.GetLenModuli() == len(Q) - 1                  # p0.Coeffs = p0.Coeffs[:level]
    got = p.get()
    for b in range(2):
        assert np.array_equal(got[b], oc.rescale_op(oname, x[b])), name
    # a second rescale on the shrunk poly
    getattr(ctx, name)(p)
    got = p.get()
    for b in range(2):
        assert np.array_equal(got[b], oc.rescale_op(oname, oc.rescale_op(oname, x[b])))


@pytest.mark.parametrize("ntt", [False, True])
@pytest.mark.parametrize("rounding", ["Floor", "Round"])
def test_rescale_many(gpu_pkg, oracle, ntt, rounding):
    N = 1 << 11
    Q = list(gpu_pkg.params.Qi60()[-5:])
    ctx = gpu_pkg.ring.NewContextWithParams(N, Q)
    oc = oracle.Context(N, Q)
    x = gpu_pkg.sampling.uniform_poly(Q, N, 2, seed=4)
    p = ctx.NewPoly(2).set(x)
    getattr(ctx, "Div%sByLastModulusMany%s" % (rounding, "NTT" if ntt else ""))(p, 3)
    got = p.get()
    for b in range(2):
        assert np.array_equal(got[b], oc.rescale_op("oc_div_%s_by_last_modulus_many" % rounding.lower(), x[b], nb=3, ntt=ntt))


@pytest.mark.parametrize("logn", [4, 10, 11, 13, 14, 15])
def test_galois(gpu_pkg, oracle, logn):
    """ring/ring_galois.go: PermuteNTT / PermuteNTTIndex / Context.Permute, and the identity the reference's
    testGaloisShift relies on: NTT(Permute(x)) == PermuteNTT(NTT(x))."""
    N = 1 << logn
    moduli = list(gpu_pkg.params.Qi60()[-3:])
    ring = gpu_pkg.ring
    ctx = ring.NewContextWithParams(N, moduli)
    oc = oracle.Context(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, 2, seed=logn)
    x[0, 0, 3] = 0                                # a zero whose sign flips becomes q in Context.Permute
    px, po = ctx.NewPoly(2).set(x), ctx.NewPoly(2)
    # (round 4: Context.Permute at N = 2^11 ... 2^14 stages the row through LDS and reads the map from the other side, the other degrees
    # keep the scatter; an even "generator" -- not an automorphism, the reference still computes something -- keeps the scatter too)
    for gen in (5, pow(5, 3, 2 * N), 2 * N - 1, 2 * N + 5, pow(5, N // 4 + 1, 2 * N)):
        ring.PermuteNTT(ctx, px, gen, po)
        for b in range(2):
            assert np.array_equal(po.get()[b], oc.permute_ntt(x[b], gen))
        ctx.Permute(px, gen, po)
        for b in range(2):
            assert np.array_equal(po.get()[b], oc.permute(x[b], gen))
    assert np.array_equal(ring.PermuteNTTIndex(5, 7, N), oc.permute_ntt_index(5, 7))
    assert np.array_equal(oc.permute_ntt_with_index(x[0], oc.permute_ntt_index(5, 7)), oc.permute_ntt(x[0], pow(5, 7, 2 * N)))
    assert ring.GenGaloisParams(N, 5)[:3] == [1, 5, 25 % (2 * N)]
    # NTT(Permute(x)) == PermuteNTT(NTT(x)) modulo q
    gen = 5
    a, bb = ctx.NewPoly(2), ctx.NewPoly(2)
    ctx.Permute(px, gen, a)
    ctx.Reduce(a, a)
    ctx.NTT(a, a)
    ctx.NTT(px, po)
    ring.PermuteNTT(ctx, po, gen, bb)
    assert np.array_equal(a.get(), bb.get())
    with pytest.raises(ring.LatticeRingError):
        ring.PermuteNTT(ctx, px, gen, px)          # "Careful, not inplace!"


def test_marshal_binary_round_trip(gpu_pkg):
    """Poly.MarshalBinary / UnmarshalBinary (ring/ring_object.go:222,252): log2 N, number of moduli, limb-major
    big-endian words -- checked against an independent numpy encoding, both directions, plus the reference's
    length check"""
    N, moduli = 1 << 10, list(gpu_pkg.params.Qi60()[-3:])
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, 2, seed=13)
    p = ctx.NewPoly(2).set(x)
    for b in range(2):
        want = bytes([10, 3]) + x[b].astype(">u8").tobytes()
        assert p.MarshalBinary(b) == want
    q = ctx.NewPoly(2)
    q.UnmarshalBinary(bytes([10, 3]) + x[1].astype(">u8").tobytes(), 0)
    q.UnmarshalBinary(bytes([10, 2]) + x[0][:2].astype(">u8").tobytes(), 1)     # fewer moduli: first rows only
    got = q.get()
    assert np.array_equal(got[0], x[1]) and np.array_equal(got[1][:2], x[0][:2])
    with pytest.raises(gpu_pkg._native.LatticeRingError):
        q.UnmarshalBinary(bytes([10, 3]) + x[1].astype(">u8").tobytes()[:-8], 0)
    with pytest.raises(gpu_pkg._native.LatticeRingError):
        q.UnmarshalBinary(bytes([11, 3]) + x[1].astype(">u8").tobytes(), 0)


@pytest.mark.parametrize("deg", [0, 1, 5, 63, 64, 65, 127, 128 + 7, 1000003])
def test_mult_by_monomial(gpu_pkg, oracle, deg):
    """Context.MultByMonomial (ring/ring.go:663): negacyclic shift; negated coefficients are q - x unreduced, so a zero
    coefficient comes out as q exactly as in the reference"""
    N, moduli = 64, list(gpu_pkg.params.Qi60()[-2:])
    ctx, oc = gpu_pkg.ring.NewContextWithParams(N, moduli), oracle.Context(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, 2, seed=17)
    x[0, :, 3] = 0
    x[1, :, N - 1] = 0
    p, r = ctx.NewPoly(2).set(x), ctx.NewPoly(2)
    ctx.MultByMonomial(p, deg, r)
    for b in range(2):
        assert np.array_equal(r.get()[b], oc.mult_by_monomial(x[b], deg)), (deg, b)


# ---- SimpleScaler (ring/ring_scaling.go:166-300) -----------------------------------------------------------------------
@pytest.mark.parametrize("t", [65537, 1 << 16, 786433, 2, (1 << 40) + 15])
@pytest.mark.parametrize("logn,limbs", [(4, 1), (10, 3), (13, 4)])
def test_simple_scaler_against_oracle(gpu_pkg, oracle, t, logn, limbs):
    """NewSimpleScaler tables (host, double-double) and Scale (device, double-double + Z_t arithmetic), bit for bit: full-range
    residues, both reduction flavours (t a power of two or not), output into a one-limb context T as bfv/encoder.go:142 does"""
    N = 1 << logn
    moduli = (list(gpu_pkg.params.Qi60()[-2:]) + gpu_pkg.params.GenerateNTTPrimes(40, max(logn, 4), 1) + gpu_pkg.params.GenerateNTTPrimes(50, max(logn, 4), 1))[:limbs]
    ring = gpu_pkg.ring
    ctx, oc = ring.NewContextWithParams(N, moduli), oracle.Context(N, moduli)
    s, os_ = ring.NewSimpleScaler(t, ctx), oracle.SimpleScaler(t, oc)
    wi, ti = s.tables()
    assert np.array_equal(wi, os_.wi)
    assert ti.tobytes() == os_.ti.tobytes()
    batch = 3
    x = gpu_pkg.sampling.uniform_poly(moduli, N, batch, seed=t % 97 + logn).reshape(batch, limbs, N)
    x[0, :, 0] = 0
    for i, q in enumerate(moduli):
        x[1, i, 1] = q - 1
    p1 = ctx.NewPoly(batch).set(x)
    tprime = gpu_pkg.params.GenerateNTTPrimes(30, max(logn, 4), 1)
    ctxT = ring.NewContextWithParams(N, tprime)
    p2 = ctxT.NewPoly(batch)
    s.Scale(p1, p2)
    got = p2.get().reshape(batch, 1, N)
    for b in range(batch):
        assert np.array_equal(got[b], os_.scale(x[b], 1)), (t, logn, b)
    # into a poly of the same context: every limb receives the value (:296-298); in place as well
    p3 = ctx.NewPoly(batch)
    s.Scale(p1, p3)
    got3 = p3.get().reshape(batch, limbs, N)
    s.Scale(p1, p1)
    got1 = p1.get().reshape(batch, limbs, N)
    for b in range(batch):
        want = os_.scale(x[b], limbs)
        assert np.array_equal(got3[b], want) and np.array_equal(got1[b], want)


def test_simple_scaler_decodes_bfv_plaintext(gpu_pkg, oracle):
    """bfv/encoder.go:142 semantics at the full PN14QP438 size: Delta*m + noise -> m"""
    from fractions import Fraction
    N, Q, _, _ = gpu_pkg.params.bfv_moduli("PN14QP438")
    Q, t = list(Q), 65537
    ctx = gpu_pkg.ring.NewContextWithParams(N, Q)
    s = gpu_pkg.ring.NewSimpleScaler(t, ctx)
    rng = np.random.default_rng(9)
    m = rng.integers(0, t, N)
    bigQ = 1
    for q in Q:
        bigQ *= q
    delta = bigQ // t
    noise = rng.integers(-(1 << 20), 1 << 20, N)
    x = np.array([[(delta * int(v) + int(e)) % q for v, e in zip(m, noise)] for q in Q], dtype=np.uint64)
    p1, p2 = ctx.NewPoly(1).set(x[None]), ctx.NewPoly(1)
    s.Scale(p1, p2)
    got = p2.get().reshape(len(Q), N)
    assert np.array_equal(got[0], m.astype(np.uint64)) and np.array_equal(got[-1], got[0])


def test_simple_scaler_errors(gpu_pkg):
    ring = gpu_pkg.ring
    moduli = list(gpu_pkg.params.Qi60()[-2:])
    ctx = ring.NewContextWithParams(16, moduli)
    with pytest.raises(gpu_pkg.ring.LatticeRingError):
        ring.NewSimpleScaler(0, ctx)                          # the reference divides by zero in BRedParams
    s = ring.NewSimpleScaler(65537, ctx)
    other = ring.NewContextWithParams(32, moduli)
    with pytest.raises(gpu_pkg.ring.LatticeRingError):
        s.Scale(ctx.NewPoly(1), other.NewPoly(1))             # degree mismatch
    with pytest.raises(gpu_pkg.ring.LatticeRingError):
        s.Scale(ctx.NewPoly(2), ctx.NewPoly(1))               # batch mismatch
    with pytest.raises(gpu_pkg.ring.LatticeRingError):
        s.Scale(ctx.NewPolyLvl(0, 1), ctx.NewPoly(1))         # p1 misses a modulus


@pytest.mark.parametrize("logn", [12, 13, 14])
def test_simple_scaling_reference_property(gpu_pkg, logn):
    """twin of ring/ring_test.go:587-624 on the device path: T = 0x3ee0001, DefaultParamsQi[logN], in place, every
    coefficient against round(T*x/Q) mod T computed with big integers"""
    from fractions import Fraction
    t, (N, moduli) = 0x3ee0001, gpu_pkg.params.DefaultParamsQi(logn)
    moduli = list(moduli)
    rng = np.random.default_rng(100 + logn)
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    s = gpu_pkg.ring.NewSimpleScaler(t, ctx)
    Q = 1
    for q in moduli:
        Q *= q
    raw = rng.bytes(96 * N)
    xs = [int.from_bytes(raw[96 * i:96 * i + 96], "little") % Q for i in range(N)]
    p = ctx.NewPoly(1).set(np.array([[x % q for x in xs] for q in moduli], dtype=np.uint64)[None])
    s.Scale(p, p)
    got = p.get().reshape(len(moduli), N)[0]
    want = np.array([int(Fraction(2 * t * x + Q, 2 * Q)) % t for x in xs], dtype=np.uint64)
    assert np.array_equal(got, want)


# ---- the alternative paths behind the environment switches of INTEGRATION.md section 7 stay verified ---------------------------
@pytest.mark.parametrize("nq,np_,logn", [(16, 16, 12), (3, 18, 10), (18, 3, 11)])
def test_modup_per_term_fallback(gpu_pkg, oracle, nq, np_, logn, monkeypatch):
    monkeypatch.setenv("LR_EXT_NARROW", "1")
    test_modup(gpu_pkg, oracle, nq, np_, logn)


@pytest.mark.parametrize("logn", [10, 12, 13])
def test_rescale_round_unfused_agrees(gpu_pkg, oracle, logn, monkeypatch):
    """the rounding rescale with the constant folded into the subtract-multiply (default) and with the explicit shifted copies"""
    for unfused in (False, True):
        if unfused:
            monkeypatch.setenv("LR_RESCALE_UNFUSED", "1")
        else:
            monkeypatch.delenv("LR_RESCALE_UNFUSED", raising=False)
        test_rescale(gpu_pkg, oracle, "DivRoundByLastModulusNTT", logn)


# ---- the HBM-bound family at the BASELINE shapes (config 2: N = 2^14, 8 limbs; the benched ring R15: N = 2^15, 16 limbs) ----------
# The kernels put the limb on blockIdx.y and the batch on blockIdx.z with 16 B per lane: a shape-dependent indexing error would not
# show at N = 2^12 x 3 limbs x batch 2.  Every coefficient of every poly against the oracle.
REAL_SHAPES = [(14, 8, 3), (15, 16, 3)]


@pytest.mark.parametrize("logn,limbs,batch", REAL_SHAPES)
@pytest.mark.parametrize("op", ["MUL_MONT", "MUL_MONT_AND_ADD_NOMOD", "MUL_MONT_AND_ADD", "MUL_COEFFS", "ADD", "SUB"])
def test_three_operand_ops_at_baseline_shapes(gpu_pkg, oracle, logn, limbs, batch, op):
    N, moduli = gpu_pkg.params.DefaultParamsQi(logn)
    moduli = list(moduli)
    assert len(moduli) == limbs
    ctx, oc = gpu_pkg.ring.NewContextWithParams(N, moduli), oracle.Context(N, moduli)
    mk = lambda s: gpu_pkg.sampling.uniform_poly(moduli, N, batch, seed=900 + s).reshape(batch, limbs, N)
    a, b, c = mk(1), mk(2), mk(3)
    pa, pb, pc = ctx.NewPoly(batch).set(a), ctx.NewPoly(batch).set(b), ctx.NewPoly(batch).set(c)
    level = limbs - 1 if op != "MUL_MONT_AND_ADD_NOMOD" else limbs - 2      # the ...Lvl form on a lower level: top limb untouched
    ctx._ew(op, level, pa, pb, pc)
    got = pc.get().reshape(batch, limbs, N)
    for i in range(batch):
        want = oc.ewise(op, a[i], b[i], out=c[i], level=level)
        assert np.array_equal(got[i, :level + 1], want[:level + 1]), (op, i)
        assert np.array_equal(got[i, level + 1:], c[i, level + 1:]), (op, i)


@pytest.mark.parametrize("logn,limbs,batch", REAL_SHAPES)
def test_two_operand_ops_at_baseline_shapes(gpu_pkg, oracle, logn, limbs, batch):
    N, moduli = gpu_pkg.params.DefaultParamsQi(logn)
    moduli = list(moduli)
    ctx, oc = gpu_pkg.ring.NewContextWithParams(N, moduli), oracle.Context(N, moduli)
    a = gpu_pkg.sampling.uniform_poly(moduli, N, batch, seed=77).reshape(batch, limbs, N)
    pa, pc = ctx.NewPoly(batch).set(a), ctx.NewPoly(batch)
    for op in ("MFORM", "INV_MFORM", "NEG"):
        ctx._ew(op, limbs - 1, pa, None, pc)
        got = pc.get().reshape(batch, limbs, N)
        for i in range(batch):
            assert np.array_equal(got[i], oc.ewise(op, a[i])), (op, i)
    ctx.MulScalar(pa, 0x123456789ABCDEF1, pc)
    got = pc.get().reshape(batch, limbs, N)
    for i in range(batch):
        assert np.array_equal(got[i], oc.ewise("MUL_SCALAR", a[i], scalars=[0x123456789ABCDEF1])), i


def test_extension_division_is_the_ieee_quotient(gpu_pkg):
    """the float64 correction index of modUpExact divides by a table constant; the kernels do that with a host reciprocal and two
    residual corrections (lr_bext.hip: div_by_const).  2^32 operand pairs on the device -- every divisor size, powers of two and
    all-ones patterns, dividends over 64 bits, below the divisor, at multiples of it +- 1 and around 2^53 -- must give the IEEE
    quotient bit for bit"""
    N, moduli = gpu_pkg.params.DefaultParamsQi(12)
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    for seed in (1, 0x4C415454):
        assert ctx.selftest_division(1 << 31, seed) == 0


@pytest.mark.parametrize("logn,limbs,batch", REAL_SHAPES)
def test_modup_split_qp_at_baseline_shapes(gpu_pkg, oracle, logn, limbs, batch):
    """ModUpSplitQP 8 -> 8 at N = 2^14 (config 2's ring) and 16 -> 16 at N = 2^15 (the benched extension), every coefficient"""
    N, Q = gpu_pkg.params.DefaultParamsQi(logn)
    _, P = gpu_pkg.params.DefaultParamsPi(logn)
    Q, P = list(Q), list(P)
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    be = ring.NewFastBasisExtender(cQ, cP)
    obe = oracle.BasisExtender(oracle.Context(N, Q), oracle.Context(N, P))
    x = gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=55).reshape(batch, limbs, N)
    px, pp = cQ.NewPoly(batch).set(x), cP.NewPoly(batch)
    be.ModUpSplitQP(limbs - 1, px, pp)
    got = pp.get().reshape(batch, len(P), N)
    for b in range(batch):
        assert np.array_equal(got[b], obe.modup_split_qp(limbs - 1, x[b])), b
    be.ModUpSplitPQ(len(P) - 1, pp, px)
    got = px.get().reshape(batch, limbs, N)
    for b in range(batch):
        assert np.array_equal(got[b], obe.modup_split_pq(len(P) - 1, obe.modup_split_qp(limbs - 1, x[b]))), b


@pytest.mark.parametrize("logn,limbs,batch", REAL_SHAPES)
@pytest.mark.parametrize("name", ["DivFloorByLastModulusNTT", "DivRoundByLastModulusNTT", "DivRoundByLastModulus"])
def test_rescale_at_baseline_shapes(gpu_pkg, oracle, logn, limbs, batch, name):
    N, moduli = gpu_pkg.params.DefaultParamsQi(logn)
    moduli = list(moduli)
    ctx, oc = gpu_pkg.ring.NewContextWithParams(N, moduli), oracle.Context(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, batch, seed=31).reshape(batch, limbs, N)
    p = ctx.NewPoly(batch).set(x)
    getattr(ctx, name)(p)
    oname = {"DivFloorByLastModulusNTT": "oc_div_floor_by_last_modulus_ntt", "DivRoundByLastModulusNTT": "oc_div_round_by_last_modulus_ntt",
             "DivRoundByLastModulus": "oc_div_round_by_last_modulus"}[name]
    got = p.get().reshape(batch, limbs - 1, N)
    for b in range(batch):
        assert np.array_equal(got[b], oc.rescale_op(oname, x[b])), (name, b)


# ---- boundary behaviour added in round 2 ----------------------------------------------------------------------------------------
def test_mult_by_monomial_in_place(gpu_pkg, oracle):
    """p1 == p2 is legal in the reference (it always copies through tmpx, ring/ring.go:682)"""
    N, moduli = 256, list(gpu_pkg.params.Qi60()[-3:])
    ctx, oc = gpu_pkg.ring.NewContextWithParams(N, moduli), oracle.Context(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, 2, seed=21)
    for deg in (1, 100, 256 + 17):
        p = ctx.NewPoly(2).set(x)
        ctx.MultByMonomial(p, deg, p)
        for b in range(2):
            assert np.array_equal(p.get()[b], oc.mult_by_monomial(x[b], deg)), (deg, b)


@pytest.mark.parametrize("logn,limbs", [(3, 2), (5, 2), (8, 3), (12, 2)])
def test_shift_and_rotate(gpu_pkg, oracle, logn, limbs):
    """Context.Shift (ring/ring.go:575) and Context.Rotate (:775, which writes into p1): against the restatement, out of place and
    in place, and the reference's own property testGaloisShift (ring_test.go:422-449): BitReverse, InvNTT, Rotate by 1, NTT,
    BitReverse, Reduce of a uniform poly equals Shift by 1.  N = 8 and 32 are below 64, where Go's mask (1 << N) - 1 is not all
    ones: n = 2^N + 3 shifts by 3 there"""
    N = 1 << logn
    moduli = list(gpu_pkg.params.Qi60()[-limbs:])
    ctx, oc = gpu_pkg.ring.NewContextWithParams(N, moduli), oracle.Context(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, 2, seed=31)
    for n in (0, 1, 5, N - 1, N):
        p, r = ctx.NewPoly(2).set(x), ctx.NewPoly(2)
        ctx.Shift(p, n, r)
        for b in range(2):
            assert np.array_equal(r.get()[b], oc.shift(x[b], n)), (n, b)
        ctx.Shift(p, n, p)
        assert np.array_equal(p.get(), r.get()), n
    if N < 64:
        p, r = ctx.NewPoly(2).set(x), ctx.NewPoly(2)
        ctx.Shift(p, (1 << N) + 3, r)
        assert np.array_equal(r.get()[0], oc.shift(x[0], 3))
    else:
        assert oc.shift(x[0], N + 1) is None                     # the reference's slice expression panics
        with pytest.raises(gpu_pkg._native.LatticeRingError):
            ctx.Shift(ctx.NewPoly(2).set(x), N + 1, ctx.NewPoly(2))
    full = gpu_pkg.sampling.random_u64((2, limbs, N), seed=32)    # Rotate reduces what it multiplies; coefficient 0 stays as it is
    for n in (0, 1, 7, N + 3):
        for src in (x, full):
            p = ctx.NewPoly(2).set(src)
            ctx.Rotate(p, n, None)
            for b in range(2):
                assert np.array_equal(p.get()[b], oc.rotate(src[b], n)), (n, b)
    # ring_test.go: testGaloisShift
    rev = np.array([int(format(j, "0%db" % logn)[::-1], 2) for j in range(N)])
    bitrev = lambda a: _bit_reverse(a, rev)
    want = ctx.NewPoly(2).set(x)
    test = ctx.NewPoly(2).set(bitrev(x))
    ctx.InvNTT(test, test)
    ctx.Rotate(test, 1, test)
    ctx.NTT(test, test)
    got = bitrev(test.get())
    ctx.Shift(want, 1, want)
    assert np.array_equal(got, want.get())


def _bit_reverse(a, rev):
    """Context.BitReverse (ring/ring.go:749): out[rev(j)] = in[j]"""
    out = np.empty_like(a)
    out[..., rev] = a
    return out


def test_last_ntt_kernel_is_reported(gpu_pkg):
    """the dispatched kernel is observable: assembly code object where one exists, the C++ kernel otherwise"""
    ring, params = gpu_pkg.ring, gpu_pkg.params
    N, moduli = params.DefaultParamsQi(15)
    ctx = ring.NewContextWithParams(N, list(moduli))
    p = ctx.NewPoly(1)
    ctx.NTT(p, p)
    assert ctx.last_ntt_kernel() == "lr_ntt_fwd15h_m1"      # 16 workgroups would leave the chip idle: two 2^14 sub-blocks per limb
    ctx.InvNTT(p, p)
    assert ctx.last_ntt_kernel() == "lr_ntt_inv15h_m1"
    big = ctx.NewPoly(16)                                   # 256 workgroups: one per transform
    ctx.NTT(big, big)
    assert ctx.last_ntt_kernel() == "lr_ntt_fwd15_m1"
    ctx.InvNTT(big, big)
    assert ctx.last_ntt_kernel() == "lr_ntt_inv15_m1"
    small = ring.NewContextWithParams(256, list(params.Qi60()[-2:]))
    q = small.NewPoly(1)
    small.NTT(q, q)
    assert small.last_ntt_kernel() == "ntt_fwd_kernel<8>"


def test_ntt_batch_beyond_grid_limit_is_chunked_on_the_same_kernel(gpu_pkg, oracle):
    """more than 65535 polys in one call: the launch is cut into chunks on the assembly kernel (grid.y limit), no change of
    code path; every poly of the batch is checked (identical inputs -> identical outputs, one of them against the oracle)"""
    N, moduli = 1 << 12, list(gpu_pkg.params.Qi60()[-1:])
    batch = 65535 + 70
    ctx, oc = gpu_pkg.ring.NewContextWithParams(N, moduli), oracle.Context(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, 2, seed=4).reshape(2, 1, N)
    host = np.empty((batch, 1, N), dtype=np.uint64)
    host[0::2], host[1::2] = x[0], x[1]
    src, dst = ctx.NewPoly(batch).set(host), ctx.NewPoly(batch)
    ctx.NTT(src, dst)
    assert ctx.last_ntt_kernel() == "lr_ntt_fwd12x_m1"
    got = dst.get().reshape(batch, 1, N)
    w0, w1 = oc.ntt(x[0]), oc.ntt(x[1])
    assert np.array_equal(got[0::2], np.broadcast_to(w0, got[0::2].shape))
    assert np.array_equal(got[1::2], np.broadcast_to(w1, got[1::2].shape))
    ctx.InvNTT(dst, dst)
    assert np.array_equal(dst.get().reshape(batch, 1, N), host)


def test_shared_context_two_threads(gpu_pkg, oracle):
    """One Context shared by two threads, each with its own polys (the reference's goroutine-per-evaluator model,
    examples/dbfv/psi/psi.go:221): the rescale temporaries are leased per call, so interleaved calls do not see each
    other's scratch.  Different batch sizes make the two threads' scratch needs differ (the regrow path)."""
    import threading
    N = 1 << 12
    _, Q, _ = gpu_pkg.params.ckks_moduli("PN15QP880")
    Q = Q[:6]
    ctx, oc = gpu_pkg.ring.NewContextWithParams(N, Q), oracle.Context(N, Q)
    results, errors = {}, []

    def work(tid, batch):
        try:
            x = gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=100 + tid).reshape(batch, len(Q), N)
            outs = []
            for it in range(12):
                p = ctx.NewPoly(batch).set(x)
                ctx.DivRoundByLastModulusNTT(p)
                if it % 3 == 0:
                    ctx.DivFloorByLastModulusNTT(p)
                outs.append((it % 3 == 0, p.get().reshape(batch, -1, N)))
            results[tid] = (x, outs)
        except Exception as exc:      # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(t, b)) for t, b in ((0, 3), (1, 7))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for tid, (x, outs) in results.items():
        for twice, got in outs:
            for b in (0, x.shape[0] - 1):
                want = oc.rescale_op("oc_div_round_by_last_modulus_ntt", x[b])
                if twice:
                    want = oc.rescale_op("oc_div_floor_by_last_modulus_ntt", want)
                assert np.array_equal(got[b], want), (tid, twice, b)


def test_per_limb_upload_download_and_host_limb_transform(gpu_pkg, oracle):
    """lr_poly_upload_limb / lr_poly_download_limb / lr_ntt_host_limb: the forms the go 1.13 shim uses (one Go pointer per call, none
    stored in C memory); same bytes as the pointer-array forms, and the package-level ring.NTT / ring.InvNTT on one limb"""
    import ctypes as C
    nat = gpu_pkg._native
    lib = nat.lib()
    N, moduli = 1 << 11, list(gpu_pkg.params.Qi60()[:3])
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, 2, seed=17)
    p = ctx.NewPoly(2)
    for b in range(2):
        for i in range(3):
            row = np.ascontiguousarray(x[b, i])
            nat.check(lib.lr_poly_upload_limb(p.h, b, i, row.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(p.get(), x)
    ctx.NTT(p, p)
    oc = oracle.Context(N, moduli)
    for b in range(2):
        want = oc.ntt(x[b])
        for i in range(3):
            got = np.empty(N, dtype=np.uint64)
            nat.check(lib.lr_poly_download_limb(p.h, b, i, got.ctypes.data_as(C.c_void_p)))
            assert np.array_equal(got, want[i])
    for bad in ((2, 0), (0, 3), (-1, 0)):
        assert lib.lr_poly_upload_limb(p.h, bad[0], bad[1], x[0, 0].ctypes.data_as(C.c_void_p)) != 0
    # one limb under modulus 1, host slices, forward then inverse in place
    row = np.ascontiguousarray(x[1, 1])
    out = np.empty(N, dtype=np.uint64)
    nat.check(lib.lr_ntt_host_limb(ctx.h, 1, 0, row.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(out, oc.ntt(x[1])[1])
    nat.check(lib.lr_ntt_host_limb(ctx.h, 1, 1, out.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(out, row)
    assert lib.lr_ntt_host_limb(ctx.h, 3, 0, row.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) != 0


@pytest.mark.parametrize("logn", [12, 13, 14, 15, 16])
@pytest.mark.parametrize("path", ["m5", "separate"])
def test_rounding_rescale_on_the_60_bit_rings(gpu_pkg, oracle, logn, path, monkeypatch):
    """DivRoundByLastModulusNTT on the reference's benchmark moduli (60-bit: integer kernels of mode 1): by default the subtract-
    multiply-add rides in the copy-out of the forward kernels "m5" (integer arithmetic, Shoup constant); LR_NO_INT_EPILOGUE keeps the
    separate pass.  Every coefficient against the oracle, values at and above q included, and a second rescale on the shrunk poly."""
    if path == "separate":
        monkeypatch.setenv("LR_NO_INT_EPILOGUE", "1")
    else:
        monkeypatch.delenv("LR_NO_INT_EPILOGUE", raising=False)
    N = 1 << logn
    Q = list(gpu_pkg.params.Qi60()[:4])
    ctx = gpu_pkg.ring.NewContextWithParams(N, Q)
    oc = oracle.Context(N, Q)
    x = gpu_pkg.sampling.uniform_poly(Q, N, 3, seed=logn)
    for i, q in enumerate(Q):
        x[0, i, :5] = [0, q - 1, q, q + 1, 2 * q - 1]          # lazy values, as the reference's callers leave them (ring_scaling.go:19,102)
    p = ctx.NewPoly(3).set(x)
    ctx.DivRoundByLastModulusNTT(p)
    kernel = ctx.last_ntt_kernel()
    assert ("_m5" in kernel) == (path == "m5"), kernel
    got = p.get()
    for b in range(3):
        assert np.array_equal(got[b], oc.rescale_op("oc_div_round_by_last_modulus_ntt", x[b])), b
    ctx.DivRoundByLastModulusNTT(p)
    got = p.get()
    for b in range(3):
        assert np.array_equal(got[b], oc.rescale_op("oc_div_round_by_last_modulus_ntt", oc.rescale_op("oc_div_round_by_last_modulus_ntt", x[b])))


@pytest.mark.parametrize("logn,nlimbs,batch", [(4, 2, 1), (10, 3, 2), (15, 4, 3)])
def test_half_vector_scalar_ops(gpu_pkg, oracle, logn, nlimbs, batch):
    """lr_half_scalar_op: the element loops of ckks.Evaluator's AddConst / MultByConst / MultByConstAndAdd / MultByi / DivByi
    (ckks/evaluator.go:429-828) -- one scalar for the coefficients below N/2, one for the rest, per limb -- against the oracle's
    restatement, on values at and above q, in place and out of place; and the identity the reference's MultByi rests on:
    multiplying twice by (psi^(N/2), -psi^(N/2)) negates the polynomial"""
    N = 1 << logn
    _, Qf, _ = gpu_pkg.params.ckks_moduli("PN15QP880")
    Q = list(Qf[:nlimbs])
    ctx = gpu_pkg.ring.NewContextWithParams(N, Q)
    oc = oracle.Context(N, Q)
    x = gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=logn)
    for i, q in enumerate(Q):
        x[0, i, :3] = [q, q + 1, 2 * q - 1]
        x[0, i, N // 2:N // 2 + 2] = [q - 1, q]
    rng = np.random.default_rng(logn)
    lo = np.array([int(rng.integers(0, q)) for q in Q], dtype=np.uint64)
    hi = np.array([int(rng.integers(0, q)) for q in Q], dtype=np.uint64)
    y0 = gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=logn + 50)
    for op, code in (("ADD", 0), ("MRED", 1), ("MRED_ADD", 2)):
        p, o = ctx.NewPoly(batch).set(x), ctx.NewPoly(batch).set(y0)
        ctx.HalfScalarOp(op, nlimbs - 1, p, lo, hi, o)
        got = o.get().reshape(batch, nlimbs, N)
        for b in range(batch):
            assert np.array_equal(got[b], oc.half_scalar_op(code, x[b], lo, hi, out=y0[b])), (op, b)
        if op != "MRED_ADD":
            ctx.HalfScalarOp(op, nlimbs - 1, p, lo, hi, p)          # in place
            assert np.array_equal(p.get().reshape(batch, nlimbs, N), got)
    # MultByi twice = Neg (ckks/evaluator.go:746-785: imag = nttPsi[i][1], Montgomery form of psi^(N/2))
    imag = np.array([int(oc.ntt_psi[i][1]) for i in range(nlimbs)], dtype=np.uint64)
    nimag = np.array([q - int(v) for q, v in zip(Q, imag)], dtype=np.uint64)
    xc = gpu_pkg.sampling.uniform_poly(Q, N, 1, seed=99)
    p = ctx.NewPoly(1).set(xc)
    ctx.HalfScalarOp("MRED", nlimbs - 1, p, imag, nimag, p)
    ctx.HalfScalarOp("MRED", nlimbs - 1, p, imag, nimag, p)
    want = np.array([[(q - int(v)) % q for v in xc[0, i]] for i, q in enumerate(Q)], dtype=np.uint64)
    assert np.array_equal(p.get().reshape(nlimbs, N), want)
    # a level below the top leaves the upper limbs alone
    if nlimbs > 2:
        p, o = ctx.NewPoly(batch).set(x), ctx.NewPoly(batch).set(y0)
        ctx.HalfScalarOp("MRED", nlimbs - 2, p, lo, hi, o)
        assert np.array_equal(o.get().reshape(batch, nlimbs, N)[:, nlimbs - 1], y0[:, nlimbs - 1])


# ---- round 4: NTT-domain ModDown and the flooring NTT-domain rescale through the forward kernels' epilogue -------------------------
@pytest.mark.parametrize("logn,kind", [(12, "qi60"), (13, "ckks"), (14, "qi60"), (15, "qi60"), (15, "ckks"), (15, "q61"), (16, "ckks"), (16, "qi60")])
def test_moddown_ntt_and_div_floor_ntt_take_the_epilogue(gpu_pkg, oracle, logn, kind, monkeypatch):
    """ModDownSplitedNTTPQ / ModDownNTTPQ (ring_basis_extension.go:163-246) and DivFloorByLastModulusNTT (ring_scaling.go:9-35) with the
    subtract-multiply inside the forward transform's copy-out (kernels m4 / m5: 60-bit rings and CKKS-size moduli; moduli above 2^60 keep
    the separate pass), out of place and in place on the Q part (the reference's benchmark calls ModDownSplitedNTTPQ(level, p0, p1, p0)),
    at a level below the top, against the oracle and against the separate-pass form (LR_NO_EPILOGUE)"""
    N = 1 << logn
    params, ring, sampling = gpu_pkg.params, gpu_pkg.ring, gpu_pkg.sampling
    if kind == "qi60":
        Q, P = list(params.Qi60()[-5:]), list(params.Pi60()[-3:])
    elif kind == "ckks":
        _, Qf, Pf = params.ckks_moduli("PN15QP880" if logn <= 15 else "PN16QP1761")
        Q, P = list(params.GenerateNTTPrimes(40, logn, 4)) + [params.GenerateNTTPrimes(50, logn, 1)[0]], list(params.GenerateNTTPrimes(51, logn, 2))
    else:
        big = [p for p in params.GenerateNTTPrimes(60, logn, 8) if p > (1 << 60)]
        Q, P = big[:4], big[4:6]
    nq, np_ = len(Q), len(P)
    B = 3
    xq, xp = sampling.uniform_poly(Q, N, B, seed=logn), sampling.uniform_poly(P, N, B, seed=logn + 1)
    obe = oracle.BasisExtender(oracle.Context(N, Q), oracle.Context(N, P))
    oc = oracle.Context(N, Q)
    got = {}
    for env in ({}, {"LR_NO_EPILOGUE": "1"}):
        monkeypatch.delenv("LR_NO_EPILOGUE", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
        be = ring.NewFastBasisExtender(cQ, cP)
        for level in (nq - 1, nq - 3):
            pq, pp, out = cQ.NewPoly(B).set(xq), cP.NewPoly(B).set(xp), cQ.NewPolyLvl(level, B)
            be.ModDownSplitedNTTPQ(level, pq, pp, out)
            g = out.get().reshape(B, level + 1, N)
            for b in range(B):
                assert np.array_equal(g[b], obe.moddown_split_ntt_pq(level, xq[b], xp[b])), (env, level, b)
            pp.set(xp)
            be.ModDownSplitedNTTPQ(level, pq, pp, pq)                        # in place on the Q part
            assert np.array_equal(pq.get().reshape(B, nq, N)[:, :level + 1], g), (env, level, "in place")
            if level < nq - 1:
                assert np.array_equal(pq.get().reshape(B, nq, N)[:, level + 1:], xq[:, level + 1:])      # limbs above the level are left alone
            joined = ring.Poly(cQ, nq + np_, B).set(np.concatenate([xq, xp], axis=1))
            be.ModDownNTTPQ(level, joined, out)
            assert np.array_equal(out.get().reshape(B, level + 1, N), g), (env, level, "joined")
            got[(bool(env), level)] = g
        p = cQ.NewPoly(B).set(xq)
        cQ.DivFloorByLastModulusNTT(p)
        f = p.get().reshape(B, nq - 1, N)
        for b in range(B):
            assert np.array_equal(f[b], oc.rescale_op("oc_div_floor_by_last_modulus_ntt", xq[b])), (env, b)
        cQ.DivFloorByLastModulusNTT(p)                                          # one level further down (another table)
        f2 = p.get().reshape(B, nq - 2, N)
        for b in range(B):
            assert np.array_equal(f2[b], oc.rescale_op("oc_div_floor_by_last_modulus_ntt", oc.rescale_op("oc_div_floor_by_last_modulus_ntt", xq[b]))), (env, b)
        got[(bool(env), "floor")] = f
    for k in [k for k in got if not k[0]]:
        assert np.array_equal(got[k], got[(True, k[1])]), k

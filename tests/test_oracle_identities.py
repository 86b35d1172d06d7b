"""Pins the CPU oracle by the big-integer identities the reference's own tests use
(ring/ring_test.go): python ints are the ground truth."""
import random

import numpy as np
import pytest

from conftest import crt_reconstruct

import __graft_entry__ as graft

params = graft.load_package().params
sampling = graft.load_package().sampling

R = 1 << 64


def test_prime_lists_regenerate_reference_tables():
    Qi, Pi = params.Qi60(), params.Pi60()
    assert len(Qi) == 100 and len(Pi) == 100
    # first/last entries of ring/params.go:28-69
    assert (Qi[0], Qi[-1]) == (1152921504606584833, 1152921504050839553)
    assert (Pi[0], Pi[-1]) == (576460752308273153, 576460752586407937)
    assert all(q % (1 << 17) == 1 for q in Qi + Pi)


def test_default_moduli_match_survey():
    N, Q, P = params.ckks_moduli("PN15QP880")
    assert N == 1 << 15 and len(Q) == 18 and len(P) == 3
    assert Q[:4] == [1125899908022273, 1099512938497, 1099514314753, 1099515691009]
    assert P == [1125899908612097, 1125899909398529, 1125899910316033]
    N, Q, P, M = params.bfv_moduli("PN14QP438")
    assert Q == [72057594038321153, 36028797019389953, 36028797019488257, 18014398510661633, 18014398511382529,
                 18014398512136193]
    assert P == [36028797020209153, 36028797020602369]
    assert M[0] == 1152921504607338497 and len(M) == 6


def test_generate_ntt_primes_agrees(oracle):
    # testGenerateNTTPrimes, ring/ring_test.go:85-103
    got = oracle.generate_ntt_primes(55, 13, 4)
    assert got == params.GenerateNTTPrimes(55, 13, 4)
    for q in got:
        assert q % (2 << 13) == 1 and params.is_prime(q)


@pytest.mark.parametrize("q", [params.Qi60()[-1], params.Pi60()[3], 1099512938497, 1152921504607338497])
def test_mred_bred_vs_bigint(oracle, q):
    # testBRed / testMRed, ring/ring_test.go:352-420
    rng = random.Random(q)
    lib = oracle.lib()
    qinv = oracle.mred_params(q)
    assert (q * qinv) % R == 1
    u = oracle.bred_params(q)
    assert (u[0] << 64) + u[1] == (1 << 128) // q
    uu = (oracle.u64 * 2)(*u)
    for _ in range(4096):
        x, y = rng.randrange(q), rng.randrange(q)
        assert lib.oc_bred(x, y, q, uu) == x * y % q
        ym = lib.oc_mform(y, q, uu)
        assert ym == y * R % q
        assert lib.oc_mred(x, ym, q, qinv) == x * y % q
        assert lib.oc_inv_mform(ym, q, qinv) == y          # testMForm, :454-473
        big = rng.randrange(R)
        assert lib.oc_bred_add(big, q, uu) == big % q
        lazy = lib.oc_mred_constant(x, ym, q, qinv)
        assert lazy < 2 * q and lazy % q == x * y % q


def test_context_constants_vs_pow(oracle):
    N, moduli = params.DefaultParamsQi(12)
    ctx = oracle.Context(N, moduli)
    for i, q in enumerate(moduli):
        g = oracle.lib().oc_primitive_root(q)
        psi = pow(g, (q - 1) // (2 * N), q)
        assert pow(psi, N, q) == q - 1                               # primitive 2N-th root
        assert int(ctx.psi_mont[i]) == psi * R % q
        assert int(ctx.n_inv[i]) == pow(N, -1, q) * R % q
        logn = N.bit_length() - 1
        for j in (0, 1, 2, 5, N // 2, N - 1):
            rev = int(format(j, "0%db" % logn)[::-1], 2)
            assert int(ctx.ntt_psi[i][rev]) == pow(psi, j, q) * R % q
            assert int(ctx.ntt_psi_inv[i][rev]) == pow(psi, -j, q) * R % q
    # rescaleParams[j-1][i] = MForm(q_j^-1 mod q_i), ring_context.go:148-158
    assert int(ctx.rescale[0][0]) == pow(moduli[1], -1, moduli[0]) * R % moduli[0]


def test_rejects_non_ntt_moduli(oracle):
    with pytest.raises(ValueError):
        oracle.Context(1 << 12, [1099512938497 + 2])           # not prime / not 1 mod 2N
    with pytest.raises(ValueError):
        oracle.Context(12, [1099512938497])                    # not a power of two


def _schoolbook_negacyclic(a, b, q):
    n = len(a)
    out = [0] * n
    for i in range(n):
        for j in range(n):
            k = i + j
            if k < n:
                out[k] = (out[k] + a[i] * b[j]) % q
            else:
                out[k - n] = (out[k - n] - a[i] * b[j]) % q
    return out


def test_mulpoly_ntt_vs_schoolbook(oracle):
    # testMulPoly, ring/ring_test.go:503-548 (MulPolyMontgomery vs MulPolyNaive), small degree
    N = 64
    moduli = [576460752303439873, 576460752303702017]
    ctx = oracle.Context(N, moduli)
    a = sampling.uniform_poly(moduli, N, 1, seed=5)[0]
    b = sampling.uniform_poly(moduli, N, 1, seed=6)[0]
    prod = ctx.intt(ctx.ewise("MUL_MONT", ctx.ewise("MFORM", ctx.ntt(a)), ctx.ntt(b)))
    for i, q in enumerate(moduli):
        want = _schoolbook_negacyclic([int(v) for v in a[i]], [int(v) for v in b[i]], q)
        assert [int(v) for v in prod[i]] == want


@pytest.mark.parametrize("nq,np_", [(2, 2), (4, 4), (8, 8), (16, 16), (3, 18), (6, 6)])
def test_extend_basis_exact(oracle, nq, np_):
    # testExtendBasis, ring/ring_test.go:550-585: ModUpSplitQP(x mod q_i) == x mod p_j for 0 <= x < Q
    N = 256
    Q = list(params.Qi60()[-nq:])
    P = list(params.Pi60()[-np_:])
    cQ, cP = oracle.Context(N, Q), oracle.Context(N, P)
    be = oracle.BasisExtender(cQ, cP)
    rng = random.Random(nq * 100 + np_)
    bigQ = 1
    for m in Q:
        bigQ *= m
    xs = [rng.randrange(bigQ) for _ in range(N)]
    pol = np.array([[x % m for x in xs] for m in Q], dtype=np.uint64)
    got = be.modup_split_qp(nq - 1, pol)
    want = np.array([[x % m for x in xs] for m in P], dtype=np.uint64)
    assert np.array_equal(got, want)


def _div_round(a, b):
    # ring.DivRound, ring/int.go:38-50 (a, b >= 0 here)
    quo, rem = divmod(a, b)
    return quo + 1 if 2 * rem >= b else quo


@pytest.mark.parametrize("rounding", ["floor", "round"])
def test_div_by_last_modulus_many(oracle, rounding):
    # testDivFloor/RoundByLastModulusMany, ring/ring_test.go:134-220
    N = 128
    moduli = list(params.Qi60()[-5:])
    ctx = oracle.Context(N, moduli)
    rng = random.Random(77)
    bigQ = 1
    for m in moduli:
        bigQ *= m
    xs = [rng.randrange(bigQ) for _ in range(N)]
    pol = np.array([[x % m for x in xs] for m in moduli], dtype=np.uint64)
    nb = 3
    name = "oc_div_%s_by_last_modulus_many" % rounding
    got = ctx.rescale_op(name, pol, nb=nb, ntt=False)
    want = list(xs)
    for k in range(nb):
        ql = moduli[len(moduli) - 1 - k]
        want = [(_div_round(x, ql) if rounding == "round" else x // ql) for x in want]
    rest = moduli[:len(moduli) - nb]
    want_pol = np.array([[x % m for x in want] for m in rest], dtype=np.uint64)
    assert np.array_equal(got, want_pol)
    # NTT-domain variant == coefficient-domain variant conjugated by the transform (:58-62,:153-157)
    got_ntt = ctx.rescale_op(name, ctx.ntt(pol), nb=nb, ntt=True)
    assert np.array_equal(got_ntt, oracle.Context(N, rest).ntt(want_pol))


@pytest.mark.parametrize("rounding", ["floor", "round"])
def test_div_by_last_modulus_ntt_single(oracle, rounding):
    N = 128
    moduli = list(params.Qi60()[-4:])
    ctx = oracle.Context(N, moduli)
    pol = sampling.uniform_poly(moduli, N, 1, seed=31)[0]
    coeff = ctx.rescale_op("oc_div_%s_by_last_modulus" % rounding, pol)
    viantt = ctx.rescale_op("oc_div_%s_by_last_modulus_ntt" % rounding, ctx.ntt(pol))
    assert np.array_equal(viantt, oracle.Context(N, moduli[:-1]).ntt(coeff))


def test_moddown_divides_by_p(oracle):
    # ModDownPQ: x over Q||P -> round-ish(x / P) over Q; exact identity: result == (x - [x]_P) / P mod q_i
    N = 64
    Q = list(params.Qi60()[-4:])
    P = list(params.Pi60()[-2:])
    cQ, cP = oracle.Context(N, Q), oracle.Context(N, P)
    be = oracle.BasisExtender(cQ, cP)
    rng = random.Random(3)
    bigQP, bigP = 1, 1
    for m in Q + P:
        bigQP *= m
    for m in P:
        bigP *= m
    xs = [rng.randrange(bigQP) for _ in range(N)]
    pol = np.array([[x % m for x in xs] for m in Q + P], dtype=np.uint64)
    got = be.moddown_pq(len(Q) - 1, pol)
    want = np.array([[((x - x % bigP) // bigP) % m for x in xs] for m in Q], dtype=np.uint64)
    assert np.array_equal(got, want)
    # NTT-domain variants agree with the coefficient-domain one
    cQP = oracle.Context(N, Q + P)
    pol_ntt = cQP.ntt(pol)
    assert np.array_equal(be.moddown_ntt_pq(len(Q) - 1, pol_ntt), cQ.ntt(want))
    assert np.array_equal(be.moddown_split_ntt_pq(len(Q) - 1, pol_ntt[:len(Q)], pol_ntt[len(Q):]), cQ.ntt(want))
    assert np.array_equal(be.moddown_split_pq(len(Q) - 1, pol[:len(Q)], pol[len(Q):]), want)


@pytest.mark.parametrize("nq,np_,level", [(6, 2, 5), (6, 2, 4), (6, 2, 2), (7, 3, 6), (7, 3, 3), (5, 1, 4), (18, 3, 17)])
def test_decompose_reconstructs_digit(oracle, nq, np_, level):
    """Decompose/DecomposeAndSplit: digit i of x (its residues on the digit's limbs, CRT-lifted)
    re-expressed in every limb of Q[0..level] and P."""
    N = 32
    Q = list(params.Qi60()[-nq:])
    P = list(params.Pi60()[-np_:])
    dec = oracle.Decomposer(Q, P)
    pol = sampling.uniform_poly(Q, N, 1, seed=nq * 31 + level)[0]
    beta = -(-(level + 1) // np_)
    for crt in range(beta):
        st = crt * np_
        ed = min(st + np_, nq, level + 1)
        digit_moduli = Q[st:ed]
        lifted = crt_reconstruct(pol[st:ed], digit_moduli)
        outQ, outP = dec.decompose_and_split(level, crt, pol)
        if ed - st == 1:
            # single usable limb: the reference copies it verbatim into every target limb without
            # reducing it (ring_basis_extension.go:490-497,613-623); later NTTs accept values >= q
            wantQ = np.tile(pol[st], (level + 1, 1))
            wantP = np.tile(pol[st], (len(P), 1))
        else:
            wantQ = np.array([[x % m for x in lifted] for m in Q[:level + 1]], dtype=np.uint64)
            wantP = np.array([[x % m for x in lifted] for m in P], dtype=np.uint64)
        assert np.array_equal(outQ, wantQ), (crt,)
        assert np.array_equal(outP, wantP), (crt,)
        joined = dec.decompose(level, crt, pol)
        assert np.array_equal(joined, np.concatenate([wantQ, wantP]))


def test_mul_scalar_bigint(oracle):
    # testMulScalarBigint, ring/ring_test.go:475-501
    N = 32
    moduli = list(params.Qi60()[-3:])
    ctx = oracle.Context(N, moduli)
    pol = sampling.uniform_poly(moduli, N, 1, seed=9)[0]
    scalar = (1 << 100) + 12345
    got = ctx.ewise("MUL_SCALAR_LIMBS", pol, scalars=[scalar % m for m in moduli])
    want = np.array([[int(v) * scalar % m for v in pol[i]] for i, m in enumerate(moduli)], dtype=np.uint64)
    assert np.array_equal(got, want)
    got1 = ctx.ewise("MUL_SCALAR", pol, scalars=[0xFFFFFFFFFFFFFFF1])
    want1 = np.array([[int(v) * 0xFFFFFFFFFFFFFFF1 % m for v in pol[i]] for i, m in enumerate(moduli)], dtype=np.uint64)
    assert np.array_equal(got1, want1)


def test_ckks_switch_keys_is_linear_in_the_key(oracle):
    """Algebraic check of the oracle's switchKeysInPlace: with evakey[i] = (D_i, 0) where D_i = P * (the
    digit-i CRT idempotent) in NTT+Montgomery form, sum_i decompose_i(c) * D_i / P == c, so p0 == c (+ rounding 0)."""
    N = 32
    Q = list(params.Qi60()[-4:])
    P = list(params.Pi60()[-2:])
    cQ, cP = oracle.Context(N, Q), oracle.Context(N, P)
    plan = oracle.CkksPlan(cQ, cP)
    level = len(Q) - 1
    alpha, beta = len(P), 2
    bigP = P[0] * P[1]
    bigQ = 1
    for m in Q:
        bigQ *= m
    QP = Q + P
    evk = np.zeros((beta, 2, len(QP), N), dtype=np.uint64)
    for i in range(beta):
        dm = Q[i * alpha:(i + 1) * alpha]
        D = 1
        for m in dm:
            D *= m
        other = bigQ // D
        idem = other * pow(other, -1, D)          # 1 mod digit moduli, 0 mod the others
        val = bigP * idem
        for k, m in enumerate(QP):
            evk[i, 0, k, :] = (val % m) * R % m    # constant polynomial in NTT domain, Montgomery form
    cx_coeff = sampling.uniform_poly(Q, N, 1, seed=77)[0]
    cx = cQ.ntt(cx_coeff)
    p0, p1 = plan.switch_keys(level, cx, evk)
    assert np.array_equal(p1, np.zeros_like(p1))
    assert np.array_equal(p0, cx)


@pytest.mark.parametrize("logn", [3, 5, 8, 11])
def test_galois_shift_property_of_the_reference(oracle, logn):
    """ring_test.go:422-449 (testGaloisShift) on the restatement: BitReverse, InvNTT, Rotate by 1, NTT, BitReverse, Reduce of a uniform
    poly equals Shift by 1 -- pins oc_rotate (ring/ring.go:775) and oc_shift (:575) on the reference's own test"""
    import __graft_entry__ as graft
    pkg = graft.load_package()
    N = 1 << logn
    moduli = list(pkg.params.Qi60()[-2:])
    oc = oracle.Context(N, moduli)
    x = pkg.sampling.uniform_poly(moduli, N, 1, seed=logn)[0]
    rev = np.array([int(format(j, "0%db" % logn)[::-1], 2) for j in range(N)])

    def bit_reverse(a):                      # Context.BitReverse, ring/ring.go:749
        out = np.empty_like(a)
        out[..., rev] = a
        return out

    t = bit_reverse(oc.ntt(oc.rotate(oc.intt(bit_reverse(x)), 1)))
    assert np.array_equal(t, oc.shift(x, 1))
    assert np.array_equal(oc.shift(x, 0), x) and np.array_equal(oc.shift(x, N), x)
    if N >= 64:
        assert oc.shift(x, N + 1) is None    # p1.Coeffs[i][n:] with n > N panics in the reference
    else:
        assert np.array_equal(oc.shift(x, (1 << N) + 2), oc.shift(x, 2))   # Go's mask (1 << N) - 1


def test_bfv_relinearize_decrypts_like_the_degree_two_ciphertext(oracle):
    """Pins the restatement of bfv.evaluator.switchKeys / relinearize (bfv/evaluator.go:480-501, 736-812), for which the reference holds
    no vectors: with a relinearisation key built the way the reference builds it (bfv/keygen.go -> newSwitchingKey: for digit i
    (-a_i*s + e_i + P*s^2 on the limbs of digit i, a_i) over Q||P, NTT + Montgomery form), c0' + c1'*s must equal c0 + c1*s + c2*s^2
    up to a small noise that is the same integer polynomial modulo every q_i.  Python integers decide."""
    import __graft_entry__ as graft
    pkg = graft.load_package()
    N = 1 << 6
    Qf, Pf = pkg.params.Qi60(), pkg.params.Pi60()
    Q, P = list(Qf[:4]), list(Pf[:2])
    QP = Q + P
    nq, np_ = len(Q), len(P)
    alpha, beta = np_, -(-nq // np_)
    ocQ, ocP, ocQP = oracle.Context(N, Q), oracle.Context(N, P), oracle.Context(N, QP)
    plan = oracle.CkksPlan(ocQ, ocP)
    rng = np.random.default_rng(5)
    res = lambda v, mods: np.array([[int(x) % q for x in v] for q in mods], dtype=np.uint64)
    mul = lambda a, b, mods: np.array([[int(x) * int(y) % q for x, y in zip(a[i], b[i])] for i, q in enumerate(mods)], dtype=np.uint64)
    add = lambda a, b, mods: np.array([[(int(x) + int(y)) % q for x, y in zip(a[i], b[i])] for i, q in enumerate(mods)], dtype=np.uint64)
    neg = lambda a, mods: np.array([[(q - int(x)) % q for x in a[i]] for i, q in enumerate(mods)], dtype=np.uint64)
    mont = lambda a, mods: np.array([[(int(x) << 64) % q for x in a[i]] for i, q in enumerate(mods)], dtype=np.uint64)
    Pprod = 1
    for p in P:
        Pprod *= p
    s = rng.integers(-1, 2, size=N)
    s_ntt = ocQP.ntt(res(s, QP))
    s2_ntt = mul(s_ntt, s_ntt, QP)
    evk = np.zeros((beta, 2, nq + np_, N), dtype=np.uint64)
    for i in range(beta):
        a_i = pkg.sampling.uniform_poly(QP, N, 1, seed=300 + i)[0]
        e_i = ocQP.ntt(res(rng.integers(-6, 7, size=N), QP))
        k0 = add(neg(mul(a_i, s_ntt, QP), QP), e_i, QP)
        for j in range(alpha):
            idx = i * alpha + j
            if idx < nq:
                q = QP[idx]
                k0[idx] = np.array([(int(x) + (Pprod % q) * int(y)) % q for x, y in zip(k0[idx], s2_ntt[idx])], dtype=np.uint64)
        evk[i, 0], evk[i, 1] = mont(k0, QP), mont(a_i, QP)
    ct = pkg.sampling.uniform_poly(Q, N, 3, seed=7)                       # any degree-2 "ciphertext", coefficient domain
    out = plan.bfv_relinearize(ct, evk)
    sq, s2q = s_ntt[:nq], s2_ntt[:nq]
    lhs = ocQ.intt(add(ocQ.ntt(out[0]), mul(ocQ.ntt(out[1]), sq, Q), Q))
    rhs = ocQ.intt(add(add(ocQ.ntt(ct[0]), mul(ocQ.ntt(ct[1]), sq, Q), Q), mul(ocQ.ntt(ct[2]), s2q, Q), Q))
    noise = None
    for i, q in enumerate(Q):
        d = [((int(x) - int(y)) % q + q // 2) % q - q // 2 for x, y in zip(lhs[i], rhs[i])]
        assert max(abs(v) for v in d) < 1 << 24, (i, max(abs(v) for v in d))
        if noise is None:
            noise = d
        assert d == noise                                                  # one integer polynomial, whatever the modulus
    # and the two halves separately: relinearize = (c0 + p0, c1 + p1) with (p0, p1) = switchKeys(c2)
    p0, p1 = plan.bfv_switch_keys(ct[2], evk)
    assert np.array_equal(out[0], add(ct[0], p0, Q)) and np.array_equal(out[1], add(ct[1], p1, Q))


@pytest.mark.parametrize("gen_kind", ["column", "row"])
def test_bfv_permute_decrypts_to_the_automorphism_of_the_plaintext(oracle, gen_kind):
    """Pins the restatement of bfv.evaluator.permute (bfv/evaluator.go:711-735; no vectors in the reference): with a rotation key built the
    way genrotKey builds it (a switching key from phi(s) to s: digit i holds (-a_i*s + e_i + P*phi(s) on its own limbs, a_i) over Q||P,
    NTT + Montgomery form), out0 + out1*s must equal phi(c0 + c1*s) up to a small noise that is one integer polynomial modulo every q_i.
    phi (X -> X^gen) is evaluated here on Python integers, independently of oc_permute."""
    import __graft_entry__ as graft
    pkg = graft.load_package()
    N = 1 << 6
    Q, P = list(pkg.params.Qi60()[:4]), list(pkg.params.Pi60()[:2])
    QP = Q + P
    nq, np_ = len(Q), len(P)
    alpha, beta = np_, -(-nq // np_)
    gen = pow(5, 3, 2 * N) if gen_kind == "column" else 2 * N - 1
    ocQ, ocP, ocQP = oracle.Context(N, Q), oracle.Context(N, P), oracle.Context(N, QP)
    plan = oracle.CkksPlan(ocQ, ocP)
    rng = np.random.default_rng(11)

    def phi(v):                                   # integer polynomial, X^i -> X^(i*gen) in Z[X]/(X^N + 1)
        out = [0] * N
        for i, x in enumerate(v):
            e = (i * gen) % (2 * N)
            out[e % N] += -int(x) if e >= N else int(x)
        return out
    res = lambda v, mods: np.array([[int(x) % q for x in v] for q in mods], dtype=np.uint64)
    mul = lambda a, b, mods: np.array([[int(x) * int(y) % q for x, y in zip(a[i], b[i])] for i, q in enumerate(mods)], dtype=np.uint64)
    add = lambda a, b, mods: np.array([[(int(x) + int(y)) % q for x, y in zip(a[i], b[i])] for i, q in enumerate(mods)], dtype=np.uint64)
    neg = lambda a, mods: np.array([[(q - int(x)) % q for x in a[i]] for i, q in enumerate(mods)], dtype=np.uint64)
    mont = lambda a, mods: np.array([[(int(x) << 64) % q for x in a[i]] for i, q in enumerate(mods)], dtype=np.uint64)
    Pprod = P[0] * P[1]
    s = [int(x) for x in rng.integers(-1, 2, size=N)]
    s_ntt = ocQP.ntt(res(s, QP))
    phis_ntt = ocQP.ntt(res(phi(s), QP))
    evk = np.zeros((beta, 2, nq + np_, N), dtype=np.uint64)
    for i in range(beta):
        a_i = pkg.sampling.uniform_poly(QP, N, 1, seed=400 + i)[0]
        e_i = ocQP.ntt(res(rng.integers(-6, 7, size=N), QP))
        k0 = add(neg(mul(a_i, s_ntt, QP), QP), e_i, QP)
        for j in range(alpha):
            idx = i * alpha + j
            if idx < nq:
                q = QP[idx]
                k0[idx] = np.array([(int(x) + (Pprod % q) * int(y)) % q for x, y in zip(k0[idx], phis_ntt[idx])], dtype=np.uint64)
        evk[i, 0], evk[i, 1] = mont(k0, QP), mont(a_i, QP)
    ct = pkg.sampling.uniform_poly(Q, N, 2, seed=17)                      # any degree-1 "ciphertext", coefficient domain
    ct[0][:, 1] = 0                                                       # a zero whose sign flips (q, not 0, in the reference)
    out = plan.bfv_permute(ct, gen, evk)
    sq = s_ntt[:nq]
    lhs = ocQ.intt(add(ocQ.ntt(out[0]), mul(ocQ.ntt(out[1]), sq, Q), Q))
    dec = ocQ.intt(add(ocQ.ntt(ct[0]), mul(ocQ.ntt(ct[1]), sq, Q), Q))  # c0 + c1*s, per limb
    noise = None
    for i, q in enumerate(Q):
        want = [x % q for x in phi([int(x) for x in dec[i]])]
        d = [((int(x) - y) % q + q // 2) % q - q // 2 for x, y in zip(lhs[i], want)]
        assert max(abs(v) for v in d) < 1 << 24, (i, max(abs(v) for v in d))
        if noise is None:
            noise = d
        assert d == noise
    # the halves: out1 = p1 and out0 = Permute(c0) + p0 with (p0, p1) = switchKeys(Permute(c1)); Permute against phi on integers
    perm1 = ocQ.permute(ct[1], gen)
    for i, q in enumerate(Q):
        assert [int(x) % q for x in perm1[i]] == [x % q for x in phi([int(x) for x in ct[1][i]])]
    p0, p1 = plan.bfv_switch_keys(perm1, evk)
    assert np.array_equal(out[1], p1)
    assert np.array_equal(out[0], add(ocQ.permute(ct[0], gen), p0, Q))


def test_bfv_square_branch_equals_the_regular_tensor_on_equal_operands(oracle):
    """tensorAndRescale's squaring case (bfv/evaluator.go:306,334-349: the operand lifted once, c1 = 2 c0[0] c0[1] by AddNoMod) and the
    regular case fed the same ciphertext twice give the same three polys: MRed(MForm(x), y) and MRed(MForm(y), x) are the same canonical
    residue, and everything after the tensor is exact.  The device library relies on this to read the first operand's slots twice."""
    from importlib import import_module  # noqa: F401
    for name, logn in (("PN12QP109", 8), ("PN13QP218", 9), ("PN14QP438", 10)):
        _, Q, _, M = params.bfv_moduli(name)
        N = 1 << logn
        plan = oracle.BfvPlan(oracle.Context(N, list(Q)), oracle.Context(N, list(M)), 65537)
        rng = np.random.default_rng(logn)
        ct = np.stack([np.array([rng.integers(0, q, size=N, dtype=np.uint64) for q in Q]) for _ in range(2)])
        sq, reg = plan.square(ct), plan.mul(ct, ct.copy())
        assert np.array_equal(sq, reg), name
        assert sq.any()

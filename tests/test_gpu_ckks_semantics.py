"""End-to-end algebra on top of the bit parity: a CKKS relinearisation key built the way the reference builds it
(ckks/keygen.go:newSwitchingKey -- for digit i: (-a_i*s + e_i + P*s^2 on the limbs of digit i, a_i) over Q||P, NTT +
Montgomery form) must make MulRelin's output decrypt to the product of the two plaintexts up to a small noise.  This
checks the meaning of the pipeline (key layout, digit decomposition, ModDown by P), not just its agreement with the
oracle's restatement of the same call sequence.  Python integers are the arbiter; the oracle only supplies NTTs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _small(N, bound, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(-bound, bound + 1, size=N)


def _residues(v, moduli):
    return np.array([[int(x) % q for x in v] for q in moduli], dtype=np.uint64)


def _mulmod(a, b, moduli):
    return np.array([[int(x) * int(y) % q for x, y in zip(a[i], b[i])] for i, q in enumerate(moduli)], dtype=np.uint64)


def _addmod(a, b, moduli):
    return np.array([[(int(x) + int(y)) % q for x, y in zip(a[i], b[i])] for i, q in enumerate(moduli)], dtype=np.uint64)


def _negmod(a, moduli):
    return np.array([[(q - int(x)) % q for x in a[i]] for i, q in enumerate(moduli)], dtype=np.uint64)


def _mont(a, moduli):
    return np.array([[(int(x) << 64) % q for x in a[i]] for i, q in enumerate(moduli)], dtype=np.uint64)


@pytest.mark.parametrize("logn,nq,np_", [(10, 3, 1), (11, 4, 2)])
def test_mulrelin_decrypts_to_the_product(gpu_pkg, oracle, logn, nq, np_):
    N = 1 << logn
    _, Qf, Pf = gpu_pkg.params.ckks_moduli("PN15QP880")
    Q, P = Qf[:nq], Pf[:np_]
    QP = Q + P
    level = nq - 1
    alpha, beta = np_, -(-nq // np_)
    ocQP, ocQ = oracle.Context(N, QP), oracle.Context(N, Q)
    Pprod = 1
    for p in P:
        Pprod *= p

    s = _small(N, 1, 1)                                            # ternary secret
    s_ntt = ocQP.ntt(_residues(s, QP))
    s2_ntt = _mulmod(s_ntt, s_ntt, QP)

    # relinearisation key: evakey[i] = (-a_i*s + e_i + P*s^2 [limbs of digit i only], a_i), ckks/keygen.go:238-270
    evk = np.zeros((beta, 2, nq + np_, N), dtype=np.uint64)
    for i in range(beta):
        a_i = gpu_pkg.sampling.uniform_poly(QP, N, 1, seed=700 + i)[0]
        e_i = ocQP.ntt(_residues(_small(N, 6, 710 + i), QP))
        k0 = _addmod(_negmod(_mulmod(a_i, s_ntt, QP), QP), e_i, QP)
        for j in range(alpha):
            idx = i * alpha + j
            if idx >= nq:
                break
            q = QP[idx]
            k0[idx] = np.array([(int(x) + (Pprod % q) * int(y)) % q for x, y in zip(k0[idx], s2_ntt[idx])], dtype=np.uint64)
        evk[i, 0], evk[i, 1] = _mont(k0, QP), _mont(a_i, QP)

    # two ciphertexts (b, a) with b = -a*s + m + e over Q, NTT domain
    def encrypt(seed):
        a = gpu_pkg.sampling.uniform_poly(Q, N, 1, seed=seed)[0]
        me = _small(N, 1 << 12, seed + 1) + _small(N, 6, seed + 2)     # message + error, what decryption returns
        me_ntt = ocQ.ntt(_residues(me, Q))
        b = _addmod(_negmod(_mulmod(a, s_ntt[:nq], Q), Q), me_ntt, Q)
        return (b, a), me

    (b0, a0), m0 = encrypt(800)
    (b1, a1), m1 = encrypt(810)

    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, 1)
    pevk = plan.NewSwitchingKey().set(evk.reshape(2 * beta, nq + np_, N))
    P_ = lambda x: cQ.NewPolyLvl(level, 1).set(x[None])
    out = (cQ.NewPolyLvl(level, 1), cQ.NewPolyLvl(level, 1))
    plan.MulRelin(level, (P_(b0), P_(a0)), (P_(b1), P_(a1)), pevk, out)
    d0, d1 = out[0].get().reshape(nq, N), out[1].get().reshape(nq, N)

    # decrypt: d0 + d1*s, back to coefficients, centred; must equal the negacyclic product m0*m1 up to the
    # relinearisation noise (digits * e_i / P plus rounding), the same small integer vector in every limb
    dec = ocQ.intt(_addmod(d0, _mulmod(d1, s_ntt[:nq], Q), Q))
    prod = np.zeros(N, dtype=object)
    m0o, m1o = [int(x) for x in m0], [int(x) for x in m1]
    for i in range(N):
        if m0o[i] == 0:
            continue
        for j in range(N):
            k = i + j
            if k < N:
                prod[k] += m0o[i] * m1o[j]
            else:
                prod[k - N] -= m0o[i] * m1o[j]
    noise = None
    for i, q in enumerate(Q):
        centred = np.array([((int(x) - int(p)) % q + q // 2) % q - q // 2 for x, p in zip(dec[i], prod)], dtype=object)
        assert max(abs(int(v)) for v in centred) < 1 << 20, (i, max(abs(int(v)) for v in centred))
        if noise is None:
            noise = centred
        else:
            assert all(int(u) == int(v) for u, v in zip(noise, centred))   # one integer polynomial, consistently in all limbs

    # control: the same key with the P*s^2 term of digit 0 left out must NOT decrypt (the check above is not vacuous)
    bad = evk.copy()
    a_0 = np.array([[int(x) * pow(1 << 64, -1, q) % q for x in bad[0, 1, i]] for i, q in enumerate(QP)], dtype=np.uint64)
    e_0 = ocQP.ntt(_residues(_small(N, 6, 710), QP))
    bad[0, 0] = _mont(_addmod(_negmod(_mulmod(a_0, s_ntt, QP), QP), e_0, QP), QP)
    pbad = plan.NewSwitchingKey().set(bad.reshape(2 * beta, nq + np_, N))
    plan.MulRelin(level, (P_(b0), P_(a0)), (P_(b1), P_(a1)), pbad, out)
    d0, d1 = out[0].get().reshape(nq, N), out[1].get().reshape(nq, N)
    dec = ocQ.intt(_addmod(d0, _mulmod(d1, s_ntt[:nq], Q), Q))
    q = Q[0]
    worst = max(abs(((int(x) - int(p)) % q + q // 2) % q - q // 2) for x, p in zip(dec[0], prod))
    assert worst > 1 << 30


def test_bfv_mul_decrypts_to_the_product(gpu_pkg, oracle):
    """bfv tensorAndRescale (bfv/evaluator.go:278): for ciphertexts (b, a), b = -a*s + floor(Q/t)*m + e, the degree-2
    output (c0, c1, c2) satisfies round(t/Q * (c0 + c1*s + c2*s^2)) = m0*m1 mod (X^N + 1, t).  Python integers decide."""
    name, t = "PN12QP109", 65537
    N, Q, _, QMul = gpu_pkg.params.bfv_moduli(name)
    ocQ = oracle.Context(N, Q)
    Qprod = 1
    for q in Q:
        Qprod *= q
    delta = Qprod // t
    s = _small(N, 1, 11)
    s_ntt = ocQ.ntt(_residues(s, Q))

    def encrypt(seed):
        a = gpu_pkg.sampling.uniform_poly(Q, N, 1, seed=seed)[0]               # coefficient domain
        m = np.random.default_rng(seed + 1).integers(0, t, size=N)
        e = _small(N, 6, seed + 2)
        as_ = ocQ.intt(_mulmod(ocQ.ntt(a), s_ntt, Q))
        body = _residues([delta * int(x) + int(y) for x, y in zip(m, e)], Q)
        return (_addmod(_negmod(as_, Q), body, Q), a), m

    (b0, a0), m0 = encrypt(900)
    (b1, a1), m1 = encrypt(910)
    ring = gpu_pkg.ring
    cQ, cM = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, QMul)
    plan = ring.BfvPlan(cQ, cM, t, 1)
    P_ = lambda x: cQ.NewPoly(1).set(x[None])
    out = (cQ.NewPoly(1), cQ.NewPoly(1), cQ.NewPoly(1))
    plan.Mul((P_(b0), P_(a0)), (P_(b1), P_(a1)), out)
    c = [o.get().reshape(len(Q), N) for o in out]

    s2_ntt = _mulmod(s_ntt, s_ntt, Q)
    acc = ocQ.ntt(c[0])
    acc = _addmod(acc, _mulmod(ocQ.ntt(c[1]), s_ntt, Q), Q)
    acc = _addmod(acc, _mulmod(ocQ.ntt(c[2]), s2_ntt, Q), Q)
    v = ocQ.intt(acc)
    # CRT lift to [0, Q), scale by t/Q with rounding, reduce mod t
    crt = [(Qprod // q) * pow(Qprod // q, -1, q) for q in Q]
    full = np.convolve(m0.astype(object), m1.astype(object))
    want = [(int(full[k]) - (int(full[k + N]) if k + N < len(full) else 0)) % t for k in range(N)]
    got = []
    for k in range(N):
        x = sum(int(v[i][k]) * crt[i] for i in range(len(Q))) % Qprod
        got.append(((t * x + Qprod // 2) // Qprod) % t)
    assert got == want

    # Relinearize (bfv/evaluator.go:512) with a key built as the reference builds it: the degree-1 result decrypts to the same product
    _, _, P, _ = gpu_pkg.params.bfv_moduli(name)
    QP = list(Q) + list(P)
    nq, np_ = len(Q), len(P)
    alpha, beta = np_, -(-nq // np_)
    ocQP = oracle.Context(N, QP)
    Pprod = 1
    for p in P:
        Pprod *= p
    sQP = ocQP.ntt(_residues(s, QP))
    s2QP = _mulmod(sQP, sQP, QP)
    evk = np.zeros((beta, 2, nq + np_, N), dtype=np.uint64)
    for i in range(beta):
        a_i = gpu_pkg.sampling.uniform_poly(QP, N, 1, seed=950 + i)[0]
        e_i = ocQP.ntt(_residues(_small(N, 6, 960 + i), QP))
        k0 = _addmod(_negmod(_mulmod(a_i, sQP, QP), QP), e_i, QP)
        for j in range(alpha):
            idx = i * alpha + j
            if idx < nq:
                q = QP[idx]
                k0[idx] = np.array([(int(x) + (Pprod % q) * int(y)) % q for x, y in zip(k0[idx], s2QP[idx])], dtype=np.uint64)
        evk[i, 0], evk[i, 1] = _mont(k0, QP), _mont(a_i, QP)
    cP = ring.NewContextWithParams(N, P)
    rl = plan.NewRelinearizer(cP, 1)
    pevk = rl.NewSwitchingKey().set(evk.reshape(2 * beta, nq + np_, N))
    lin = (cQ.NewPoly(1), cQ.NewPoly(1))
    rl.BfvRelinearize(out, pevk, lin)
    d = [o.get().reshape(len(Q), N) for o in lin]
    v1 = ocQ.intt(_addmod(ocQ.ntt(d[0]), _mulmod(ocQ.ntt(d[1]), s_ntt, Q), Q))
    got1 = []
    for k in range(N):
        x = sum(int(v1[i][k]) * crt[i] for i in range(len(Q))) % Qprod
        got1.append(((t * x + Qprod // 2) // Qprod) % t)
    assert got1 == want


@pytest.mark.parametrize("hoisted", [False, True])
def test_rotation_decrypts_to_the_permuted_plaintext(gpu_pkg, oracle, hoisted):
    """RotateColumns / RotateHoisted: with a switching key from s(X^g) to s (ckks/keygen.go: -a_i*s + e_i + P*s(X^g) on
    the limbs of digit i), the rotated ciphertext decrypts under s to the plaintext with X -> X^g applied, up to the
    key-switching noise.  The automorphism on plaintexts is Context.Permute in the coefficient domain."""
    logn, nq, np_ = 10, 4, 2
    N = 1 << logn
    _, Qf, Pf = gpu_pkg.params.ckks_moduli("PN15QP880")
    Q, P = Qf[:nq], Pf[:np_]
    QP = Q + P
    level = nq - 1
    alpha, beta = np_, -(-nq // np_)
    ocQP, ocQ = oracle.Context(N, QP), oracle.Context(N, Q)
    Pprod = P[0] * P[1]
    s = _small(N, 1, 21)
    s_ntt = ocQP.ntt(_residues(s, QP))
    gens = [pow(5, k, 2 * N) for k in (1, 7)] if hoisted else [pow(5, 3, 2 * N)]

    def rotation_key(g, seed):
        sg_ntt = ocQP.permute_ntt(s_ntt, g)                                   # s(X^g)
        key = np.zeros((beta, 2, nq + np_, N), dtype=np.uint64)
        for i in range(beta):
            a_i = gpu_pkg.sampling.uniform_poly(QP, N, 1, seed=seed + i)[0]
            e_i = ocQP.ntt(_residues(_small(N, 6, seed + 10 + i), QP))
            k0 = _addmod(_negmod(_mulmod(a_i, s_ntt, QP), QP), e_i, QP)
            for j in range(alpha):
                idx = i * alpha + j
                if idx >= nq:
                    break
                q = QP[idx]
                k0[idx] = np.array([(int(x) + (Pprod % q) * int(y)) % q for x, y in zip(k0[idx], sg_ntt[idx])], dtype=np.uint64)
            key[i, 0], key[i, 1] = _mont(k0, QP), _mont(a_i, QP)
        return key

    a = gpu_pkg.sampling.uniform_poly(Q, N, 1, seed=950)[0]
    me = _small(N, 1 << 20, 951) + _small(N, 6, 952)
    me_res = _residues(me, Q)
    b = _addmod(_negmod(_mulmod(a, s_ntt[:nq], Q), Q), ocQ.ntt(me_res), Q)

    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, 1)
    keys = [plan.NewSwitchingKey().set(rotation_key(g, 960 + 40 * n).reshape(2 * beta, nq + np_, N)) for n, g in enumerate(gens)]
    P_ = lambda x: cQ.NewPolyLvl(level, 1).set(x[None])
    ct = (P_(b), P_(a))
    outs = [(cQ.NewPolyLvl(level, 1), cQ.NewPolyLvl(level, 1)) for _ in gens]
    if hoisted:
        plan.RotateHoisted(level, ct, gens, keys, outs)
    else:
        plan.PermuteNTT(level, ct, gens[0], keys[0], outs[0])
    for g, o in zip(gens, outs):
        d0, d1 = o[0].get().reshape(nq, N), o[1].get().reshape(nq, N)
        dec = ocQ.intt(_addmod(d0, _mulmod(d1, s_ntt[:nq], Q), Q))
        want = ocQ.permute(me_res, g)                                          # m(X^g) mod q_i, coefficient domain
        for i, q in enumerate(Q):
            worst = max(abs(((int(x) - int(w)) % q + q // 2) % q - q // 2) for x, w in zip(dec[i], want[i]))
            assert worst < 1 << 16, (g, i, worst)


def test_encrypt_mulrelin_decrypt_chain(gpu_pkg, oracle):
    """The encrypt and decrypt tails of SURVEY 8(f)-2 as compositions of the ring's entry points: public-key encryption
    through the special primes (ckks/encryptor.go:207-243: MulCoeffsMontgomery, InvNTT over QP, ModDownPQ, NTT, Add),
    MulRelin, Horner decryption (ckks/decryptor.go:53-78).  decrypt(encrypt(m)) = m + small noise, and
    decrypt(MulRelin(encrypt(m0), encrypt(m1))) = m0*m1 + small noise, judged with Python integers."""
    logn, nq, np_ = 10, 4, 2
    N = 1 << logn
    _, Qf, Pf = gpu_pkg.params.ckks_moduli("PN15QP880")
    Q, P = Qf[:nq], Pf[:np_]
    QP = Q + P
    level = nq - 1
    alpha, beta = np_, -(-nq // np_)
    ocQP, ocQ = oracle.Context(N, QP), oracle.Context(N, Q)
    Pprod = P[0] * P[1]
    ring = gpu_pkg.ring
    cQ, cP, cQP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P), ring.NewContextWithParams(N, QP)
    plan = ring.CkksPlan(cQ, cP, 1)

    s = _small(N, 1, 31)
    s_ntt = ocQP.ntt(_residues(s, QP))
    s2_ntt = _mulmod(s_ntt, s_ntt, QP)
    # public key (-a*s + e, a) over QP, relinearisation key as in the first test; all in NTT + Montgomery form
    a_pk = gpu_pkg.sampling.uniform_poly(QP, N, 1, seed=1000)[0]
    pk0 = _addmod(_negmod(_mulmod(a_pk, s_ntt, QP), QP), ocQP.ntt(_residues(_small(N, 6, 1001), QP)), QP)
    ppk = (cQP.NewPoly(1).set(_mont(pk0, QP)[None]), cQP.NewPoly(1).set(_mont(a_pk, QP)[None]))
    evk = np.zeros((beta, 2, nq + np_, N), dtype=np.uint64)
    for i in range(beta):
        a_i = gpu_pkg.sampling.uniform_poly(QP, N, 1, seed=1010 + i)[0]
        k0 = _addmod(_negmod(_mulmod(a_i, s_ntt, QP), QP), ocQP.ntt(_residues(_small(N, 6, 1020 + i), QP)), QP)
        for j in range(alpha):
            idx = i * alpha + j
            if idx < nq:
                q = QP[idx]
                k0[idx] = np.array([(int(x) + (Pprod % q) * int(y)) % q for x, y in zip(k0[idx], s2_ntt[idx])], dtype=np.uint64)
        evk[i, 0], evk[i, 1] = _mont(k0, QP), _mont(a_i, QP)
    pevk = plan.NewSwitchingKey().set(evk.reshape(2 * beta, nq + np_, N))
    psk = cQ.NewPoly(1).set(_mont(s_ntt[:nq], Q)[None])

    def encrypt(m, seed):
        u = ocQP.ntt(_residues(_small(N, 1, seed), QP))
        pu = cQP.NewPoly(1).set(_mont(u, QP)[None])
        e = [cQP.NewPoly(1).set(_residues(_small(N, 6, seed + 1 + k), QP)[None]) for k in range(2)]
        pt = cQ.NewPolyLvl(level, 1).set(ocQ.ntt(_residues(m, Q))[None])
        ct = (cQ.NewPolyLvl(level, 1), cQ.NewPolyLvl(level, 1))
        plan.EncryptPk(level, pu, ppk, e, pt, ct)            # lr_ckks_encrypt_pk
        return ct

    def decrypt(ct):
        pt = cQ.NewPolyLvl(level, 1)
        plan.Decrypt(level, ct, psk, pt)
        return ocQ.intt(pt.get().reshape(nq, N))

    def centred_error(dec, want):
        worst = 0
        for i, q in enumerate(Q):
            worst = max(worst, max(abs(((int(x) - int(w)) % q + q // 2) % q - q // 2) for x, w in zip(dec[i], want)))
        return worst

    m0, m1 = _small(N, 1 << 12, 1100), _small(N, 1 << 12, 1101)
    ct0, ct1 = encrypt(m0, 1110), encrypt(m1, 1120)
    assert centred_error(decrypt(ct0), m0) < 1 << 12          # fresh noise: u*e_pk + e0 + e1*s, rounded by P
    out = (cQ.NewPolyLvl(level, 1), cQ.NewPolyLvl(level, 1))
    plan.MulRelin(level, ct0, ct1, pevk, out)
    full = np.convolve(m0.astype(object), m1.astype(object))
    prod = [int(full[k]) - (int(full[k + N]) if k + N < len(full) else 0) for k in range(N)]
    # noise of the product ~ m * fresh noise * N: far below the moduli, far above zero
    assert centred_error(decrypt(out), prod) < 1 << 34
    assert centred_error(decrypt(out), [0] * N) > 1 << 20

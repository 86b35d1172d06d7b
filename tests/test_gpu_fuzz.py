"""Seeded differential fuzzing of the C ABI against the oracle: random degrees, limb counts, levels, batches and
operations, including the shapes the structured tests do not enumerate (level 0, batch 1, odd limb counts, the small
degrees that run on the whole-limb kernel, moduli of mixed sizes, the key-switch pipeline at random levels)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _moduli(pkg, rng, logn, count):
    """a random mix of modulus sizes that are NTT-friendly for this degree"""
    pool = list(pkg.params.Qi60()[-8:]) + list(pkg.params.Pi60()[-4:])
    pool += pkg.params.GenerateNTTPrimes(40, logn, 3) + pkg.params.GenerateNTTPrimes(50, logn, 2) + pkg.params.GenerateNTTPrimes(34, logn, 1)
    pool = sorted(set(pool))
    idx = rng.choice(len(pool), size=count, replace=False)
    return [pool[i] for i in idx]


@pytest.mark.parametrize("seed", range(28))
def test_ntt_and_elementwise_fuzz(gpu_pkg, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    logn = int(rng.integers(1, 15)) if seed < 24 else 15 + seed % 2      # the last four: N = 2^15 and 2^16 (sub-block kernels, pair flags)
    N = 1 << logn
    limbs = int(rng.integers(1, 7))
    batch = int(rng.integers(1, 5))
    level = int(rng.integers(0, limbs))
    moduli = _moduli(gpu_pkg, rng, max(logn, 4), limbs)
    ctx, oc = gpu_pkg.ring.NewContextWithParams(N, moduli), oracle.Context(N, moduli)
    x = gpu_pkg.sampling.random_u64((batch, limbs, N), seed=seed)             # full-range inputs
    y = gpu_pkg.sampling.uniform_poly(moduli, N, batch, seed=seed + 1).reshape(batch, limbs, N)
    px, py, pr = ctx.NewPoly(batch).set(x), ctx.NewPoly(batch).set(y), ctx.NewPoly(batch)
    red = lambda a, b: np.array([[int(v) % q for v in a[b, i]] for i, q in enumerate(moduli)], dtype=np.uint64)

    ctx.NTTLvl(level, px, pr)
    got = pr.get().reshape(batch, limbs, N)
    for b in range(batch):
        assert np.array_equal(got[b, :level + 1], oc.ntt(red(x, b))[:level + 1]), ("ntt", logn, limbs, level, b)
    ctx.InvNTTLvl(level, py, pr)
    got = pr.get().reshape(batch, limbs, N)
    for b in range(batch):
        assert np.array_equal(got[b, :level + 1], oc.intt(y[b])[:level + 1]), ("intt", logn, limbs, level, b)

    ops = ["ADD", "SUB", "MUL_MONT", "MUL_COEFFS", "MUL_MONT_AND_ADD", "MUL_MONT_AND_SUB_NOMOD", "MUL_COEFFS_AND_ADD_NOMOD"]
    op = ops[int(rng.integers(0, len(ops)))]
    z = gpu_pkg.sampling.uniform_poly(moduli, N, batch, seed=seed + 2).reshape(batch, limbs, N)
    pz = ctx.NewPoly(batch).set(z)
    pr.set(y)                                                                  # accumulate variants read the output
    ctx._ew(op, level, pz, py, pr)
    got = pr.get().reshape(batch, limbs, N)
    for b in range(batch):
        want = oc.ewise(op, z[b], y[b], y[b].copy(), level=level)
        assert np.array_equal(got[b, :level + 1], want[:level + 1]), (op, logn, limbs, level, b)


@pytest.mark.parametrize("seed", range(10))
def test_basis_extension_and_rescale_fuzz(gpu_pkg, oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    logn = int(rng.integers(3, 13))
    N = 1 << logn
    nq, np_ = int(rng.integers(1, 9)), int(rng.integers(1, 5))
    batch = int(rng.integers(1, 4))
    mods = _moduli(gpu_pkg, rng, max(logn, 4), nq + np_)
    Q, P = mods[:nq], mods[nq:]
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    be = ring.NewFastBasisExtender(cQ, cP)
    ocQ, ocP = oracle.Context(N, Q), oracle.Context(N, P)
    obe = oracle.BasisExtender(ocQ, ocP)
    level = int(rng.integers(0, nq))
    x = gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=seed).reshape(batch, nq, N)
    px, pp = cQ.NewPoly(batch).set(x), cP.NewPoly(batch)
    be.ModUpSplitQP(level, px, pp)
    got = pp.get().reshape(batch, np_, N)
    for b in range(batch):
        assert np.array_equal(got[b], obe.modup_split_qp(level, x[b])), ("modup", nq, np_, level, b)
    if nq >= 2:
        ctx_level = nq - 1
        pr = cQ.NewPoly(batch).set(x)
        cQ.DivRoundByLastModulusNTT(pr)
        got = pr.get().reshape(batch, ctx_level, N)
        for b in range(batch):
            assert np.array_equal(got[b], ocQ.rescale_op("oc_div_round_by_last_modulus_ntt", x[b])), ("rescale", nq, b)


@pytest.mark.parametrize("seed", range(16))
def test_key_switch_fuzz(gpu_pkg, oracle, seed):
    rng = np.random.default_rng(3000 + seed)
    # seeds 8..15: the degrees of the assembly kernels (grouped launches over the digits, skipped own limbs, 2^16 sub-blocks)
    logn = int(rng.integers(4, 13)) if seed < 8 else 13 + (seed % 4)
    N = 1 << logn
    nq, np_ = int(rng.integers(2, 10)), int(rng.integers(1, 5))
    _, Qf, Pf = gpu_pkg.params.ckks_moduli("PN16QP1761" if logn == 16 or np_ > 3 else "PN15QP880")
    Q, P = Qf[:nq], Pf[:np_]
    batch = int(rng.integers(1, 4))
    level = int(rng.integers(0, nq))
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, batch)
    oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    beta = -(-nq // np_)
    evk = gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=seed + 5)
    pevk = plan.NewSwitchingKey().set(evk)
    cx = gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, batch, seed=seed).reshape(batch, level + 1, N)
    pcx = cQ.NewPolyLvl(level, batch).set(cx)
    p0, p1 = cQ.NewPolyLvl(level, batch), cQ.NewPolyLvl(level, batch)
    plan.SwitchKeysInPlace(level, pcx, pevk, p0, p1)
    g0, g1 = p0.get().reshape(batch, level + 1, N), p1.get().reshape(batch, level + 1, N)
    for b in range(batch):
        w0, w1 = oplan.switch_keys(level, cx[b], evk.reshape(beta, 2, nq + np_, N))
        assert np.array_equal(g0[b], w0) and np.array_equal(g1[b], w1), (logn, nq, np_, level, b)


def _ckks_size_moduli(pkg, rng, logn, count):
    """moduli between 30 and 56 bits: contexts that select the dual assembly kernels (FP64 body below 2^46 next to the integer one)"""
    pool = []
    for bits in (30, 34, 40, 45, 46, 50, 56):
        pool += pkg.params.GenerateNTTPrimes(bits, logn, 2)
    pool = sorted(set(pool))
    idx = rng.choice(len(pool), size=count, replace=False)
    return [pool[i] for i in idx]


@pytest.mark.parametrize("seed", range(12))
def test_dual_kernel_fuzz(gpu_pkg, oracle, seed):
    """random mixes of FP64-class and integer-class moduli at the degrees of the assembly kernels: NTT of full-range inputs, InvNTT of
    inputs anywhere in [0, 4q), at random levels and batches, in place and out of place"""
    rng = np.random.default_rng(4000 + seed)
    logn = 12 + seed % 5
    N = 1 << logn
    limbs = int(rng.integers(1, 8))
    batch = int(rng.integers(1, 4))
    level = int(rng.integers(0, limbs))
    moduli = _ckks_size_moduli(gpu_pkg, rng, logn, limbs)
    ctx, oc = gpu_pkg.ring.NewContextWithParams(N, moduli), oracle.Context(N, moduli)
    x = gpu_pkg.sampling.random_u64((batch, limbs, N), seed=seed)
    x[0, :, :3] = np.uint64(0xFFFFFFFFFFFFFFFF)
    y = x.copy()
    for i, q in enumerate(moduli):
        y[:, i] %= np.uint64(4 * q)
        y[0, i, :2] = np.uint64(4 * q - 1)
    red = lambda a, b: np.array([[int(v) % q for v in a[b, i]] for i, q in enumerate(moduli)], dtype=np.uint64)
    px, py, pr = ctx.NewPoly(batch).set(x), ctx.NewPoly(batch).set(y), ctx.NewPoly(batch)
    ctx.NTTLvl(level, px, pr)
    got = pr.get().reshape(batch, limbs, N)
    for b in range(batch):
        assert np.array_equal(got[b, :level + 1], oc.ntt(red(x, b))[:level + 1]), ("ntt", logn, moduli, level, b)
    ctx.InvNTTLvl(level, py, pr)
    got = pr.get().reshape(batch, limbs, N)
    for b in range(batch):
        assert np.array_equal(got[b, :level + 1], oc.intt(red(y, b))[:level + 1]), ("intt", logn, moduli, level, b)
    ctx.NTTLvl(level, px, px)
    ctx.InvNTTLvl(level, px, px)                     # in place, round trip
    got = px.get().reshape(batch, limbs, N)
    for b in range(batch):
        assert np.array_equal(got[b, :level + 1], red(x, b)[:level + 1]), ("round trip", logn, moduli, level, b)


@pytest.mark.parametrize("seed", range(8))
def test_mulrelin_rescale_fuzz(gpu_pkg, oracle, seed):
    """MulRelin followed by Rescale at random levels, limb counts and batches on the CKKS default moduli (dual kernels, epilogue
    kernels where the limbs allow, staging at N = 2^16) against the oracle's restatement of the same ring calls"""
    rng = np.random.default_rng(5000 + seed)
    logn = 12 + seed % 5
    N = 1 << logn
    nq, np_ = int(rng.integers(3, 9)), int(rng.integers(1, 5))
    _, Qf, Pf = gpu_pkg.params.ckks_moduli("PN16QP1761" if logn == 16 or np_ > 3 else "PN15QP880")
    Q, P = Qf[:nq], Pf[:np_]
    level = int(rng.integers(1, nq))
    batch = int(rng.integers(1, 4))
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, batch)
    oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    beta = -(-nq // np_)
    evk = gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=seed + 70)
    pevk = plan.NewSwitchingKey().set(evk)
    mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, batch, seed=s).reshape(batch, level + 1, N)
    a0, a1, b0, b1 = mk(1), mk(2), mk(3), mk(4)
    P_ = lambda x: cQ.NewPolyLvl(level, batch).set(x)
    out = (cQ.NewPolyLvl(level, batch), cQ.NewPolyLvl(level, batch))
    plan.MulRelin(level, (P_(a0), P_(a1)), (P_(b0), P_(b1)), pevk, out)
    g0, g1 = out[0].get().reshape(batch, level + 1, N), out[1].get().reshape(batch, level + 1, N)
    wants = []
    for b in range(batch):
        want = oplan.mulrelin(level, np.stack([a0[b], a1[b]]), np.stack([b0[b], b1[b]]), evk.reshape(beta, 2, nq + np_, N))
        wants.append(want)
        assert np.array_equal(g0[b], want[0]) and np.array_equal(g1[b], want[1]), ("mulrelin", logn, nq, np_, level, b)
    plan.Rescale(out)
    oc = oracle.Context(N, Q[:level + 1])
    r0, r1 = out[0].get().reshape(batch, level, N), out[1].get().reshape(batch, level, N)
    for b in range(batch):
        for k, got in ((0, r0), (1, r1)):
            assert np.array_equal(got[b], oc.rescale_op("oc_div_round_by_last_modulus_ntt", wants[b][k])), ("rescale", logn, level, b, k)


@pytest.mark.parametrize("seed", range(12))
def test_rotation_encrypt_decrypt_fuzz(gpu_pkg, oracle, seed):
    """round 4's fused forms at random shapes: permuteNTT with a random Galois element, RotateHoisted over a random set of rotations (the
    digits read through the permutation), EncryptPk (both products in one pass, the Q rows of the error in the ModDown's epilogue at the top
    level, the call-by-call form below it) and Decrypt of a random degree (one Horner pass up to degree 8)"""
    rng = np.random.default_rng(6000 + seed)
    logn = int(rng.integers(4, 13)) if seed < 6 else 12 + seed % 5
    N = 1 << logn
    nq, np_ = int(rng.integers(2, 9)), int(rng.integers(1, 5))
    _, Qf, Pf = gpu_pkg.params.ckks_moduli("PN16QP1761" if logn == 16 or np_ > 3 else "PN15QP880")
    Q, P = Qf[:nq], Pf[:np_]
    QP = Q + P
    batch = int(rng.integers(1, 4))
    level = int(rng.integers(0, nq))
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    plan = ring.CkksPlan(cQ, cP, batch)
    oplan = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    beta = -(-nq // np_)
    mk = lambda s: gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, batch, seed=s).reshape(batch, level + 1, N)
    P_ = lambda x: cQ.NewPolyLvl(level, batch).set(x)
    a0, a1 = mk(seed + 11), mk(seed + 12)

    rots = sorted(set(int(k) for k in rng.integers(1, N // 2, size=int(rng.integers(1, 5)))))
    gens = [pow(5, k, 2 * N) for k in rots]
    if rng.integers(0, 2):
        gens[-1] = 2 * N - 1                                        # conjugation
    evks = [gpu_pkg.sampling.uniform_poly(QP, N, 2 * beta, seed=seed + 300 + i) for i in range(len(gens))]
    pevks = [plan.NewSwitchingKey().set(e) for e in evks]
    evk4 = [e.reshape(beta, 2, nq + np_, N) for e in evks]
    out = (cQ.NewPolyLvl(level, batch), cQ.NewPolyLvl(level, batch))
    plan.PermuteNTT(level, (P_(a0), P_(a1)), gens[0], pevks[0], out)
    g0, g1 = out[0].get().reshape(batch, level + 1, N), out[1].get().reshape(batch, level + 1, N)
    for b in range(batch):
        want = oplan.permute_ntt(level, np.stack([a0[b], a1[b]]), gens[0], evk4[0])
        assert np.array_equal(g0[b], want[0]) and np.array_equal(g1[b], want[1]), ("permuteNTT", logn, nq, np_, level, gens[0], b)
    outs = [(cQ.NewPolyLvl(level, batch), cQ.NewPolyLvl(level, batch)) for _ in gens]
    plan.RotateHoisted(level, (P_(a0), P_(a1)), gens, pevks, outs)
    got = [(o[0].get().reshape(batch, level + 1, N), o[1].get().reshape(batch, level + 1, N)) for o in outs]
    for b in range(batch):
        want = oplan.rotate_hoisted(level, np.stack([a0[b], a1[b]]), gens, evk4)
        for r in range(len(gens)):
            assert np.array_equal(got[r][0][b], want[r][0]) and np.array_equal(got[r][1][b], want[r][1]), ("hoisted", logn, nq, np_, level, gens, r, b)

    ocQP = oracle.Context(N, QP)
    uni = lambda s, n: gpu_pkg.sampling.uniform_poly(QP, N, n, seed=s).reshape(n, nq + np_, N)
    u, e0, e1, pk0, pk1 = uni(seed + 21, batch), uni(seed + 22, batch), uni(seed + 23, batch), uni(seed + 24, 1), uni(seed + 25, 1)
    pt = mk(seed + 26)
    QPpoly = lambda x: ring.Poly(cQ, nq + np_, x.shape[0]).set(x)
    ct = (cQ.NewPolyLvl(level, batch), cQ.NewPolyLvl(level, batch))
    plan.EncryptPk(level, QPpoly(u), (QPpoly(pk0), QPpoly(pk1)), (QPpoly(e0), QPpoly(e1)), P_(pt), ct)
    c0, c1 = ct[0].get().reshape(batch, level + 1, N), ct[1].get().reshape(batch, level + 1, N)
    for b in range(batch):
        want = oplan.encrypt_pk(ocQP, level, u[b], pk0[0], pk1[0], e0[b], e1[b], pt[b])
        assert np.array_equal(c0[b], want[0][:level + 1]) and np.array_equal(c1[b], want[1][:level + 1]), ("encrypt", logn, nq, np_, level, b)

    degree = int(rng.integers(1, 10))
    comps = [mk(seed + 40 + i) for i in range(degree + 1)]
    sk = gpu_pkg.sampling.uniform_poly(Q[:level + 1], N, 1, seed=seed + 60).reshape(1, level + 1, N)
    dec = cQ.NewPolyLvl(level, batch)
    plan.Decrypt(level, tuple(P_(c) for c in comps), cQ.NewPolyLvl(level, 1).set(sk), dec)
    d = dec.get().reshape(batch, level + 1, N)
    for b in range(batch):
        assert np.array_equal(d[b], oplan.decrypt(level, np.stack([c[b] for c in comps]), sk[0])), ("decrypt", logn, level, degree, b)


@pytest.mark.parametrize("seed", range(12))
def test_moddown_divfloor_permute_fuzz(gpu_pkg, oracle, seed):
    """the ModDown family (NTT-domain form on the transform's epilogue where the kernels have one), DivFloor / DivRound in both domains
    and Context.Permute / PermuteNTT with random generators at random degrees, limb counts, levels and batches"""
    rng = np.random.default_rng(7000 + seed)
    logn = int(rng.integers(4, 13)) if seed < 6 else 11 + seed % 6
    N = 1 << logn
    nq, np_ = int(rng.integers(2, 9)), int(rng.integers(1, 5))
    batch = int(rng.integers(1, 4))
    level = int(rng.integers(0, nq))
    if seed % 3 == 0:
        _, Qf, Pf = gpu_pkg.params.ckks_moduli("PN16QP1761" if logn == 16 or np_ > 3 else "PN15QP880")
        Q, P = Qf[:nq], Pf[:np_]
    elif seed % 3 == 1:
        Q, P = list(gpu_pkg.params.Qi60()[-nq:]), list(gpu_pkg.params.Pi60()[-np_:])
    else:
        mods = _moduli(gpu_pkg, rng, max(logn, 4), nq + np_)
        Q, P = mods[:nq], mods[nq:]
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    be = ring.NewFastBasisExtender(cQ, cP)
    ocQ, ocP = oracle.Context(N, Q), oracle.Context(N, P)
    obe = oracle.BasisExtender(ocQ, ocP)
    xq = gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=seed + 1).reshape(batch, nq, N)
    xp = gpu_pkg.sampling.uniform_poly(P, N, batch, seed=seed + 2).reshape(batch, np_, N)
    pq, pp, out = cQ.NewPoly(batch).set(xq), cP.NewPoly(batch).set(xp), cQ.NewPolyLvl(level, batch)
    be.ModDownSplitedNTTPQ(level, pq, pp, out)
    got = out.get().reshape(batch, level + 1, N)
    for b in range(batch):
        assert np.array_equal(got[b], obe.moddown_split_ntt_pq(level, xq[b], xp[b])), ("moddown ntt", logn, nq, np_, level, b)
    pp.set(xp)
    be.ModDownSplitedPQ(level, pq, pp, out)
    got = out.get().reshape(batch, level + 1, N)
    for b in range(batch):
        assert np.array_equal(got[b], obe.moddown_split_pq(level, xq[b], xp[b])), ("moddown", logn, nq, np_, level, b)
    pq.set(xq)
    be.ModDownSplitedNTTPQ(level, pq, cP.NewPoly(batch).set(xp), pq)           # in place on the Q part
    got = pq.get().reshape(batch, nq, N)
    for b in range(batch):
        assert np.array_equal(got[b, :level + 1], obe.moddown_split_ntt_pq(level, xq[b], xp[b])), ("moddown ntt in place", logn, nq, np_, level, b)

    names = {"DivFloorByLastModulusNTT": "oc_div_floor_by_last_modulus_ntt", "DivFloorByLastModulus": "oc_div_floor_by_last_modulus",
             "DivRoundByLastModulusNTT": "oc_div_round_by_last_modulus_ntt", "DivRoundByLastModulus": "oc_div_round_by_last_modulus"}
    for name, oname in names.items():
        pr = cQ.NewPoly(batch).set(xq)
        getattr(cQ, name)(pr)
        got = pr.get().reshape(batch, nq - 1, N)
        for b in range(batch):
            assert np.array_equal(got[b], ocQ.rescale_op(oname, xq[b])), (name, logn, nq, b)

    po = cQ.NewPoly(batch)
    pq.set(xq)
    for gen in (pow(5, int(rng.integers(1, N)), 2 * N), 2 * N - 1, int(rng.integers(0, N)) * 2 + 1):
        ring.PermuteNTT(cQ, pq, gen, po)
        got = po.get().reshape(batch, nq, N)
        for b in range(batch):
            assert np.array_equal(got[b], ocQ.permute_ntt(xq[b], gen)), ("permute ntt", logn, gen, b)
        cQ.Permute(pq, gen, po)
        got = po.get().reshape(batch, nq, N)
        for b in range(batch):
            assert np.array_equal(got[b], ocQ.permute(xq[b], gen)), ("permute", logn, gen, b)


@pytest.mark.parametrize("seed", range(10))
def test_bfv_pipelines_fuzz(gpu_pkg, oracle, seed):
    """the BFV caller sequences at random degrees, limb counts (prefixes of the reference's parameter sets, so that |QMul| = |Q| as
    bfv/params.go has it) and batches: Mul, Square (the operand lifted once), Relinearize, RotateRows / RotateColumns with a random
    Galois element -- gathered small-batch launches and the per-operand form alike"""
    rng = np.random.default_rng(8000 + seed)
    name = ("PN12QP109", "PN13QP218", "PN14QP438", "PN15QP880")[seed % 4]
    Nfull, Qf, Pf, Mf = gpu_pkg.params.bfv_moduli(name)
    top = Nfull.bit_length() - 1
    logn = int(rng.integers(6, top + 1)) if seed < 6 else min(top, 12 + seed % 4)
    N = 1 << logn
    nq = int(rng.integers(2, len(Qf) + 1)) if len(Qf) > 2 else len(Qf)
    nq = min(nq, 7)
    np_ = int(rng.integers(1, len(Pf) + 1))
    Q, P, M = list(Qf[:nq]), list(Pf[:np_]), list(Mf[:nq])
    batch = int(rng.integers(1, 4))
    t = 65537
    ring = gpu_pkg.ring
    cQ, cP, cM = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P), ring.NewContextWithParams(N, M)
    mul, ks = ring.BfvPlan(cQ, cM, t, batch), ring.CkksPlan(cQ, cP, batch)
    omul = oracle.BfvPlan(oracle.Context(N, Q), oracle.Context(N, M), t)
    oks = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P))
    beta = -(-nq // np_)
    evk = gpu_pkg.sampling.uniform_poly(Q + P, N, 2 * beta, seed=seed + 90)
    evk4 = evk.reshape(beta, 2, nq + np_, N)
    key = ks.NewSwitchingKey().set(evk)
    mk = lambda s: gpu_pkg.sampling.uniform_poly(Q, N, batch, seed=s).reshape(batch, nq, N)
    a0, a1, b0, b1 = mk(seed + 1), mk(seed + 2), mk(seed + 3), mk(seed + 4)
    P_ = lambda x: cQ.NewPoly(batch).set(x)
    ct0, ct1 = (P_(a0), P_(a1)), (P_(b0), P_(b1))
    d2 = (cQ.NewPoly(batch), cQ.NewPoly(batch), cQ.NewPoly(batch))
    mul.Mul(ct0, ct1, d2)
    g2 = [p.get().reshape(batch, nq, N) for p in d2]
    want2 = [omul.mul(np.stack([a0[b], a1[b]]), np.stack([b0[b], b1[b]])) for b in range(batch)]
    for b in range(batch):
        for k in range(3):
            assert np.array_equal(g2[k][b], want2[b][k]), ("mul", name, logn, nq, np_, b, k)
    sq = (cQ.NewPoly(batch), cQ.NewPoly(batch), cQ.NewPoly(batch))
    mul.Mul(ct0, ct0, sq)
    gs = [p.get().reshape(batch, nq, N) for p in sq]
    for b in range(batch):
        wants = omul.square(np.stack([a0[b], a1[b]]))
        for k in range(3):
            assert np.array_equal(gs[k][b], wants[k]), ("square", name, logn, nq, b, k)
    lin = (cQ.NewPoly(batch), cQ.NewPoly(batch))
    ks.BfvRelinearize(d2, key, lin)
    gl = [p.get().reshape(batch, nq, N) for p in lin]
    for b in range(batch):
        want1 = oks.bfv_relinearize(want2[b], evk4)
        assert np.array_equal(gl[0][b], want1[0]) and np.array_equal(gl[1][b], want1[1]), ("relinearize", name, logn, nq, np_, b)
    gen = 2 * N - 1 if rng.integers(0, 2) else pow(5, int(rng.integers(1, N // 2)), 2 * N)
    rot = (cQ.NewPoly(batch), cQ.NewPoly(batch))
    ks.BfvPermute(ct0, gen, key, rot)
    gr = [p.get().reshape(batch, nq, N) for p in rot]
    for b in range(batch):
        wantr = oks.bfv_permute(np.stack([a0[b], a1[b]]), gen, evk4)
        assert np.array_equal(gr[0][b], wantr[0]) and np.array_equal(gr[1][b], wantr[1]), ("permute", name, logn, nq, np_, gen, b)

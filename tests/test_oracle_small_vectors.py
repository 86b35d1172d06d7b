"""The committed small vectors (tests/golden/oracle_small_vectors.json, written by tests/golden/make_oracle_vectors.py): the restatement
must still give them (CPU), and the HIP path through the C ABI must give them too (GPU) -- rows a4-a10 and a13 of SURVEY.md 8(a), for
which the reference itself holds no data files."""
import importlib.util
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_oracle_vectors", os.path.join(HERE, "golden", "make_oracle_vectors.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)
DATA = json.load(open(os.path.join(HERE, "golden", "oracle_small_vectors.json")))
CASES = {c["name"]: c for c in DATA["cases"]}
Q, P, N = [int(q) for q in DATA["Q"]], [int(p) for p in DATA["P"]], DATA["N"]


def test_restatement_reproduces_the_committed_vectors():
    fresh = gen.build()
    assert fresh["Q"] == DATA["Q"] and fresh["P"] == DATA["P"] and fresh["N"] == DATA["N"]
    assert [c["name"] for c in fresh["cases"]] == [c["name"] for c in DATA["cases"]]
    for f, d in zip(fresh["cases"], DATA["cases"]):
        assert f == d, d["name"]


@pytest.mark.gpu
def test_device_reproduces_the_committed_vectors(gpu_pkg):
    ring = gpu_pkg.ring
    cQ, cP = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, P)
    inp = lambda c, k: gen.dec(CASES[c]["inputs"][k])
    out = lambda c: gen.dec(CASES[c]["output"])
    up = lambda ctx, a: ctx.NewPoly(1).set(a)
    # coefficient-wise family, every op
    for name, c in CASES.items():
        if not name.startswith("ewise:"):
            continue
        op = name.split(":")[1]
        a, b, o = up(cQ, inp(name, "a")), up(cQ, inp(name, "b")), up(cQ, inp(name, "out"))
        cQ._ew(op, len(Q) - 1, a, b, o, [int(s) for s in inp(name, "scalars")])
        assert np.array_equal(o.get(), out(name)), name
    for name, fn in (("ntt", cQ.NTT), ("intt", cQ.InvNTT)):
        p, r = up(cQ, inp(name, "a")), cQ.NewPoly(1)
        fn(p, r)
        assert np.array_equal(r.get(), out(name)), name
    be = ring.NewFastBasisExtender(cQ, cP)
    r = cP.NewPoly(1)
    be.ModUpSplitQP(2, up(cQ, inp("modup_split_qp", "a")), r)
    assert np.array_equal(r.get(), out("modup_split_qp"))
    r = cQ.NewPoly(1)
    be.ModUpSplitPQ(1, up(cP, inp("modup_split_pq", "a")), r)
    assert np.array_equal(r.get(), out("modup_split_pq"))
    cQP = ring.NewContextWithParams(N, Q + P)
    for name, fn in (("moddown_pq", be.ModDownPQ), ("moddown_ntt_pq", be.ModDownNTTPQ)):
        r = cQ.NewPoly(1)
        fn(2, up(cQP, inp(name, "a")), r)
        assert np.array_equal(r.get(), out(name)), name
    r = cP.NewPoly(1)
    be.ModDownSplitedQP(2, 1, up(cQ, inp("moddown_split_qp", "a")), up(cP, inp("moddown_split_qp", "b")), r)
    assert np.array_equal(r.get(), out("moddown_split_qp"))
    for name, fn in (("rescale:oc_div_round_by_last_modulus_ntt", cQ.DivRoundByLastModulusNTT), ("rescale:oc_div_floor_by_last_modulus_ntt", cQ.DivFloorByLastModulusNTT),
                     ("rescale:oc_div_round_by_last_modulus", cQ.DivRoundByLastModulus), ("rescale:oc_div_floor_by_last_modulus", cQ.DivFloorByLastModulus)):
        p = up(cQ, inp(name, "a"))
        fn(p)
        assert np.array_equal(p.get()[:len(Q) - 1], out(name)), name
    p, r = up(cQ, inp("mult_by_monomial", "a")), cQ.NewPoly(1)
    cQ.MultByMonomial(p, CASES["mult_by_monomial"]["deg"], r)
    assert np.array_equal(r.get(), out("mult_by_monomial"))
    cQ.Shift(p, CASES["shift"]["n"], r)
    assert np.array_equal(r.get(), out("shift"))
    p2 = up(cQ, inp("rotate", "a"))
    cQ.Rotate(p2, CASES["rotate"]["n"], None)
    assert np.array_equal(p2.get(), out("rotate"))
    cQ.PermuteNTTLvl(len(Q) - 1, p, CASES["permute_ntt"]["gen"], r)
    assert np.array_equal(r.get(), out("permute_ntt"))
    # CKKS MulRelin and BFV Mul on the toy parameters of the fixture
    c1 = ring.NewContextWithParams(N, P[:1])
    plan = ring.CkksPlan(cQ, c1, 1)
    ct0, ct1, evk = inp("ckks_mulrelin", "ct0"), inp("ckks_mulrelin", "ct1"), inp("ckks_mulrelin", "evk")
    key = plan.NewSwitchingKey().set(evk.reshape(6, len(Q) + 1, N))
    o = (cQ.NewPoly(1), cQ.NewPoly(1))
    plan.MulRelin(2, (up(cQ, ct0[0]), up(cQ, ct0[1])), (up(cQ, ct1[0]), up(cQ, ct1[1])), key, o)
    assert np.array_equal(np.stack([o[0].get(), o[1].get()]), out("ckks_mulrelin"))
    bq, bm = ring.NewContextWithParams(N, Q[:2]), ring.NewContextWithParams(N, P)
    bplan = ring.BfvPlan(bq, bm, CASES["bfv_mul"]["t"], 1)
    b0, b1 = inp("bfv_mul", "ct0"), inp("bfv_mul", "ct1")
    bo = (bq.NewPoly(1), bq.NewPoly(1), bq.NewPoly(1))
    bplan.Mul((up(bq, b0[0]), up(bq, b0[1])), (up(bq, b1[0]), up(bq, b1[1])), bo)
    assert np.array_equal(np.stack([x.get() for x in bo]), out("bfv_mul"))
    # round 3: BFV Relinearize on the toy key-switch plan, and the half-vector scalar operations
    ct3 = inp("bfv_relinearize", "ct")
    c3 = [up(cQ, ct3[k]) for k in range(3)]
    lin = (cQ.NewPoly(1), cQ.NewPoly(1))
    plan.BfvRelinearize(c3, key, lin)
    assert np.array_equal(np.stack([lin[0].get(), lin[1].get()]), out("bfv_relinearize"))
    for nm in ("add", "mred", "mred_add"):
        name = "half_scalar:" + nm
        o = up(cQ, inp(name, "out"))
        cQ.HalfScalarOp({"add": "ADD", "mred": "MRED", "mred_add": "MRED_ADD"}[nm], len(Q) - 1, up(cQ, inp(name, "a")), inp(name, "lo"), inp(name, "hi"), o)
        assert np.array_equal(o.get(), out(name)), name
    # round 4: the BFV rotation body on the toy key-switch plan
    for nm in ("col", "row"):
        name = "bfv_permute:" + nm
        c2 = inp(name, "ct")
        rot = (cQ.NewPoly(1), cQ.NewPoly(1))
        plan.BfvPermute((up(cQ, c2[0]), up(cQ, c2[1])), CASES[name]["gen"], key, rot)
        assert np.array_equal(np.stack([rot[0].get(), rot[1].get()]), out(name)), name
    # round 4, second half: rotations with key, hoisted rotations, products without key (regular, squaring, plaintext), encrypt, decrypt, BFV square
    c = CASES["ckks_rotate"]
    ctr = inp("ckks_rotate", "ct")
    rot = (cQ.NewPoly(1), cQ.NewPoly(1))
    plan.PermuteNTT(2, (up(cQ, ctr[0]), up(cQ, ctr[1])), c["gen"], key, rot)
    assert np.array_equal(np.stack([rot[0].get(), rot[1].get()]), out("ckks_rotate"))
    c = CASES["ckks_rotate_hoisted"]
    key2 = plan.NewSwitchingKey().set(inp("ckks_rotate_hoisted", "evk1").reshape(6, len(Q) + 1, N))
    outs = [(cQ.NewPoly(1), cQ.NewPoly(1)) for _ in c["gens"]]
    plan.RotateHoisted(2, (up(cQ, ctr[0]), up(cQ, ctr[1])), c["gens"], [key, key2], outs)
    assert np.array_equal(np.stack([np.stack([o[0].get(), o[1].get()]) for o in outs]), out("ckks_rotate_hoisted"))
    m0, m1 = inp("ckks_mul_norelin", "ct0"), inp("ckks_mul_norelin", "ct1")
    A, B_ = (up(cQ, m0[0]), up(cQ, m0[1])), (up(cQ, m1[0]), up(cQ, m1[1]))
    d2 = (cQ.NewPoly(1), cQ.NewPoly(1), cQ.NewPoly(1))
    plan.MulRelin(2, A, B_, None, d2)
    assert np.array_equal(np.stack([x.get() for x in d2]), out("ckks_mul_norelin"))
    plan.MulRelin(2, A, A, None, d2)
    assert np.array_equal(np.stack([x.get() for x in d2]), out("ckks_square"))
    d1 = (cQ.NewPoly(1), cQ.NewPoly(1))
    mp = inp("ckks_mul_plain", "ct")
    plan.MulRelin(2, (up(cQ, inp("ckks_mul_plain", "pt")),), (up(cQ, mp[0]), up(cQ, mp[1])), None, d1)
    assert np.array_equal(np.stack([x.get() for x in d1]), out("ckks_mul_plain"))
    QPp = lambda k: ring.Poly(cQ, len(Q) + 1, 1).set(inp("ckks_encrypt_pk", k).reshape(1, len(Q) + 1, N))
    plan.EncryptPk(2, QPp("u"), (QPp("pk0"), QPp("pk1")), (QPp("e0"), QPp("e1")), up(cQ, inp("ckks_encrypt_pk", "pt")), d1)
    assert np.array_equal(np.stack([x.get() for x in d1]), out("ckks_encrypt_pk"))
    dc = inp("ckks_decrypt", "ct")
    ptd = cQ.NewPoly(1)
    plan.Decrypt(2, tuple(up(cQ, dc[k]) for k in range(3)), up(cQ, inp("ckks_decrypt", "sk")), ptd)
    assert np.array_equal(ptd.get(), out("ckks_decrypt"))
    s0 = inp("bfv_square", "ct0")
    S = (up(bq, s0[0]), up(bq, s0[1]))
    bplan.Mul(S, S, bo)
    assert np.array_equal(np.stack([x.get() for x in bo]), out("bfv_square"))

"""Generates tests/golden/oracle_small_vectors.json: small seeded input / output vectors of the rows of SURVEY.md 8(a) for which the
reference holds no data files (a4-a10, a13), computed by the CPU restatement (oracle/) at N = 16 -- the fixture SURVEY 8(c) lists next
to the reference's own test_data.  The restatement itself is pinned on the reference's golden NTT vectors and big-integer identities
(tests/test_oracle_golden.py, test_oracle_identities.py); these vectors freeze its answers so that a later edit of the oracle or of the
kernels shows up as a difference against committed data.

    python tests/golden/make_oracle_vectors.py        # rewrites the JSON next to this file
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

N = 16


def build():
    oracle, pkg = graft.load_oracle(), graft.load_package()
    P_ = pkg.params
    Q, P = [int(q) for q in P_.Qi60()[-3:]], [int(p) for p in P_.Pi60()[-2:]]
    rnd = lambda mods, seed: pkg.sampling.uniform_poly(mods, N, 1, seed=seed)[0]
    full = lambda limbs, seed: pkg.sampling.random_u64((limbs, N), seed=seed)
    ocQ, ocP = oracle.Context(N, Q), oracle.Context(N, P)
    cases = []

    def case(name, inputs, output, **extra):
        cases.append(dict(name=name, inputs={k: enc(v) for k, v in inputs.items()}, output=enc(output), **extra))

    a, b, c0 = rnd(Q, 1), rnd(Q, 2), rnd(Q, 3)
    scalars = [0x123456789ABCDEF1, 0x0FEDCBA987654321, 0x1111111111111111]
    for op in oracle.EWISE_OPS:
        sc = [7] if op == "MUL_BY_POW2" else (scalars[:1] if op == "MUL_SCALAR" else scalars)
        case("ewise:" + op, {"a": a, "b": b, "out": c0, "scalars": np.array(sc, dtype=np.uint64)}, ocQ.ewise(op, a, b, out=c0, scalars=sc))
    x = full(3, 4)
    case("ntt", {"a": x}, ocQ.ntt(x))
    y = (x % np.array(Q, dtype=np.uint64)[:, None]).astype(np.uint64)
    case("intt", {"a": y}, ocQ.intt(y))
    be = oracle.BasisExtender(ocQ, ocP)
    case("modup_split_qp", {"a": a}, be.modup_split_qp(2, a), level=2)
    pp = rnd(P, 5)
    case("modup_split_pq", {"a": pp}, be.modup_split_pq(1, pp), level=1)
    qp = np.concatenate([a, pp])
    case("moddown_pq", {"a": qp}, be.moddown_pq(2, qp), level=2)
    case("moddown_ntt_pq", {"a": qp}, be.moddown_ntt_pq(2, qp), level=2)
    case("moddown_split_qp", {"a": a, "b": pp}, be.moddown_split_qp(2, 1, a, pp), levelQ=2, levelP=1)
    dec = oracle.Decomposer(Q, P[:1])
    for crt in range(3):
        oq, op_ = dec.decompose_and_split(2, crt, a)
        case("decompose_and_split:%d" % crt, {"a": a}, np.concatenate([oq, op_]), level=2, crt=crt)
    for name in ("oc_div_round_by_last_modulus_ntt", "oc_div_floor_by_last_modulus_ntt", "oc_div_round_by_last_modulus", "oc_div_floor_by_last_modulus"):
        case("rescale:" + name, {"a": a}, ocQ.rescale_op(name, a))
    case("mult_by_monomial", {"a": a}, ocQ.mult_by_monomial(a, 21), deg=21)
    case("shift", {"a": a}, ocQ.shift(a, 5), n=5)
    case("rotate", {"a": a}, ocQ.rotate(a, 3), n=3)
    case("permute_ntt", {"a": a}, ocQ.permute_ntt(a, 5), gen=5)
    # CKKS MulRelin (ckks/evaluator.go:1016) and BFV Mul (bfv/evaluator.go:278) on toy parameters: 3 Q + 1 P limbs, alpha = 1
    plan = oracle.CkksPlan(ocQ, oracle.Context(N, P[:1]))
    evk = pkg.sampling.uniform_poly(Q + P[:1], N, 6, seed=9).reshape(3, 2, 4, N)
    ct0, ct1 = np.stack([rnd(Q, 11), rnd(Q, 12)]), np.stack([rnd(Q, 13), rnd(Q, 14)])
    case("ckks_mulrelin", {"ct0": ct0, "ct1": ct1, "evk": evk}, np.stack(plan.mulrelin(2, ct0, ct1, evk)), level=2)
    bq, bm = Q[:2], P
    bplan = oracle.BfvPlan(oracle.Context(N, bq), oracle.Context(N, bm), 65537)
    b0, b1 = np.stack([rnd(bq, 21), rnd(bq, 22)]), np.stack([rnd(bq, 23), rnd(bq, 24)])
    case("bfv_mul", {"ct0": b0, "ct1": b1}, np.stack(bplan.mul(b0, b1)), t=65537)
    # round 3: BFV Relinearize (bfv/evaluator.go:480-501, 736-812) on the same toy key-switch parameters, and the element loops of the
    # CKKS constant methods (ckks/evaluator.go:429-828)
    ct3 = np.stack([rnd(Q, 31), rnd(Q, 32), rnd(Q, 33)])
    case("bfv_relinearize", {"ct": ct3, "evk": evk}, plan.bfv_relinearize(ct3, evk))
    lo, hi = np.array(scalars, dtype=np.uint64) % np.array(Q, dtype=np.uint64), np.array(scalars[::-1], dtype=np.uint64) % np.array(Q, dtype=np.uint64)
    for code, nm in ((0, "add"), (1, "mred"), (2, "mred_add")):
        case("half_scalar:" + nm, {"a": a, "out": c0, "lo": lo, "hi": hi}, ocQ.half_scalar_op(code, a, lo, hi, out=c0), op=code)
    # round 4: the BFV rotation body (bfv/evaluator.go:711-735) on the same toy key-switch parameters, column and row generators
    ct2 = np.stack([rnd(Q, 34), rnd(Q, 35)])
    ct2[0][:, 2] = 0
    for nm, g in (("col", pow(5, 3, 2 * N)), ("row", 2 * N - 1)):
        case("bfv_permute:" + nm, {"ct": ct2, "evk": evk}, plan.bfv_permute(ct2, g, evk), gen=g)
    # round 4, second half: the remaining caller sequences of SURVEY 8(f)1-2 and the squaring cases on the same toy parameters --
    # permuteNTT / RotateHoisted (ckks/evaluator.go:1448, 1252), MulRelin without key (regular and squaring, :1038-1111), plaintext x
    # ciphertext (:1113-1131), pk-encrypt after the sampling (ckks/encryptor.go:205-234), Decrypt of degree 2 (ckks/decryptor.go:53-78),
    # BFV Mul(ct, ct) (bfv/evaluator.go:306,334-349)
    gens = [pow(5, 3, 2 * N), 2 * N - 1]
    evk2 = pkg.sampling.uniform_poly(Q + P[:1], N, 6, seed=10).reshape(3, 2, 4, N)
    case("ckks_rotate", {"ct": ct0, "evk": evk}, plan.permute_ntt(2, ct0, gens[0], evk), level=2, gen=gens[0])
    case("ckks_rotate_hoisted", {"ct": ct0, "evk0": evk, "evk1": evk2}, plan.rotate_hoisted(2, ct0, gens, [evk, evk2]), level=2, gens=gens)
    case("ckks_mul_norelin", {"ct0": ct0, "ct1": ct1}, plan.mul_norelin(2, ct0, ct1), level=2)
    case("ckks_square", {"ct0": ct0}, plan.mul_norelin(2, ct0, ct0, squaring=True), level=2)
    case("ckks_mul_plain", {"pt": a, "ct": ct1}, plan.mul_plain(2, a, ct1), level=2)
    QP1 = Q + P[:1]
    u, e0, e1, pk0, pk1 = (rnd(QP1, 41 + i) for i in range(5))
    for i, q in enumerate(QP1):
        e0[i, 0] = q                                   # SampleAndAdd's residue of -0 (ring/gaussianSampler.go:268)
    enc_out = plan.encrypt_pk(oracle.Context(N, QP1), 2, u, pk0, pk1, e0, e1, b)
    case("ckks_encrypt_pk", {"u": u, "pk0": pk0, "pk1": pk1, "e0": e0, "e1": e1, "pt": b}, enc_out, level=2)
    sk = rnd(Q, 46)
    case("ckks_decrypt", {"ct": ct3, "sk": sk}, plan.decrypt(2, ct3, sk), level=2)
    case("bfv_square", {"ct0": b0}, np.stack(bplan.square(b0)), t=65537)
    return {"N": N, "Q": [str(q) for q in Q], "P": [str(p) for p in P], "cases": cases,
            "note": "uint64 values as decimal strings, arrays flattened in C order with their shape"}


def enc(v):
    v = np.asarray(v, dtype=np.uint64)
    return {"shape": list(v.shape), "data": [str(int(t)) for t in v.ravel()]}


def dec(o):
    return np.array([int(t) for t in o["data"]], dtype=np.uint64).reshape(o["shape"])


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small_vectors.json")
    json.dump(build(), open(out, "w"), indent=0, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes")

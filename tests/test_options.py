"""lr_options (include/lattigo_ring.h): the configuration surface of the boundary.  Defaults, versioning and the size prefix are checked
without a device; that every field reaches the code path it names, and that the LR_* variables are an override of the same fields, on one."""
import ctypes as C
import re

import numpy as np
import pytest

from conftest import ROOT


def _header_fields():
    text = open(ROOT + "/include/lattigo_ring.h").read()
    body = text[text.index("typedef struct lr_options {"):text.index("} lr_options;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    return re.findall(r"\b(?:u?int(?:32|64)_t)\s+(\w+);", body)


def test_binding_mirrors_the_header_field_for_field(pkg):
    names = [n for n, _ in pkg._native.Options._fields_]
    assert names == _header_fields()
    o = pkg.ring.Options()
    assert o.struct_size == C.sizeof(pkg._native.Options) and o.version == 1
    # defaults: every flag off, every selector "by modulus size / by launch size", every threshold "built-in"
    d = o.as_dict()
    assert {k for k, v in d.items() if v == -1} == {"ntt_mode", "asm_variant", "ntt_split15", "ntt_stagger", "ntt_persist"}
    assert all(v == 0 for k, v in d.items() if k not in ("struct_size", "version") and v != -1)


def test_version_size_prefix_and_diagnostics_fields_are_checked_before_any_device_work(pkg):
    lib = pkg._native.lib()
    h = C.c_void_p()
    mod = (C.c_uint64 * 1)(1099512938497)
    o = pkg.ring.Options()
    o.version = 99
    assert lib.lr_context_create_ex(1 << 12, mod, 1, 0, C.byref(o), C.byref(h)) == 4 and b"version" in lib.lr_last_error_string()
    o = pkg.ring.Options()
    o.struct_size = 4
    assert lib.lr_context_create_ex(1 << 12, mod, 1, 0, C.byref(o), C.byref(h)) == 4 and b"struct_size" in lib.lr_last_error_string()
    if b"diag" not in lib.lr_build_info():
        # the default build ships neither the clock-stamping nor the persistent code objects: asking for them is refused, by name
        for field in ("ntt_timeline", "ntt_persist"):
            o = pkg.ring.Options(**{field: 2})
            assert lib.lr_context_create_ex(1 << 12, mod, 1, 0, C.byref(o), C.byref(h)) == 6, field
            assert b"LR_BUILD_DIAG" in lib.lr_last_error_string()
    assert lib.lr_options_init(None) == 4


def test_the_environment_is_read_in_one_place():
    """the test-only override lives in lr::Options::apply_env and nowhere else ('read in one place')"""
    import glob
    hits = []
    for f in glob.glob(ROOT + "/lattigo-fhe-by-go_amd/csrc/*.[ch]*"):
        src = open(f).read()
        for m in re.finditer(r"getenv", src):
            fn = src.rfind("void lr::Options::apply_env()", 0, m.start())
            end = src.find("\n}\n", fn) if fn >= 0 else -1
            if fn < 0 or m.start() > end:
                hits.append((f, src.count("\n", 0, m.start()) + 1))
    assert not hits, hits


@pytest.mark.gpu
def test_fields_select_the_paths_they_name_and_give_the_same_bits(gpu_pkg, oracle, monkeypatch):
    ring, params, sampling = gpu_pkg.ring, gpu_pkg.params, gpu_pkg.sampling
    N, Q = params.DefaultParamsQi(13)
    oc = oracle.Context(N, Q)
    x = sampling.uniform_poly(Q, N, 3, seed=5)
    want = np.stack([oc.ntt(x[b]) for b in range(3)])
    seen = set()
    for opt in (None, ring.Options(), ring.Options(no_asm=1), ring.Options(asm_variant=0), ring.Options(ntt_stagger=4), ring.Options(no_grid_padding=1)):
        ctx = ring.NewContextWithParams(N, Q, options=opt)
        src, dst = ctx.NewPoly(3).set(x), ctx.NewPoly(3)
        ctx.NTT(src, dst)
        assert np.array_equal(dst.get(), want), opt and opt.as_dict()
        seen.add((ctx.ntt_variants(), ctx.last_ntt_kernel()))
        eff = ctx.GetOptions()
        assert eff.split15_max_workgroups == 128 and eff.fork_below_workgroups == 256 and eff.pair_max_workgroups == 256 and eff.bfv_gather_below == 1536
        if opt is not None:
            assert eff.no_asm == opt.no_asm and eff.asm_variant == opt.asm_variant and eff.ntt_stagger == opt.ntt_stagger
    assert ((-1, -1), "ntt_fwd_kernel<13>") in seen and any(k.startswith("lr_ntt_fwd13x_m1") for _, k in seen) and any(k.startswith("lr_ntt_fwd13x_m0") for _, k in seen)
    # the environment is an override of the same fields: it switches an alternative on over the caller's struct, never off
    monkeypatch.setenv("LR_NO_FP", "1")
    monkeypatch.setenv("LR_NTT_SPLIT15_BELOW", "7")
    ctx = ring.NewContextWithParams(N, Q, options=ring.Options(no_epilogue=1))
    eff = ctx.GetOptions()
    assert eff.no_fp == 1 and eff.no_epilogue == 1 and eff.split15_max_workgroups == 7


@pytest.mark.gpu
def test_plan_options_and_thresholds(gpu_pkg, oracle):
    """lr_ckks_plan_create_ex / lr_bfv_plan_create_ex: a single ciphertext with the small-batch paths switched through the struct (what the
    LR_* variables do in the other tests), and a fork / pair threshold of one workgroup; same bits every way"""
    ring, params, sampling = gpu_pkg.ring, gpu_pkg.params, gpu_pkg.sampling
    N, Q, P = params.ckks_moduli("PN13QP218")
    Q, P = list(Q), list(P)
    nq, np_ = len(Q), len(P)
    level, beta = nq - 1, -(-nq // np_)
    evk = sampling.uniform_poly(Q + P, N, 2 * beta, seed=3)
    ops = [sampling.uniform_poly(Q, N, 1, seed=10 + k) for k in range(4)]
    want = oracle.CkksPlan(oracle.Context(N, Q), oracle.Context(N, P)).mulrelin(level, np.stack([ops[0][0], ops[1][0]]), np.stack([ops[2][0], ops[3][0]]),
                                                                                evk.reshape(beta, 2, nq + np_, N))
    for fields in ({}, {"no_pair": 1}, {"no_ext_group": 1, "keymac_narrow": 1}, {"pair_max_workgroups": 1, "fork_below_workgroups": 1}, {"no_exttop": 1, "no_invtop": 1}):
        opt = ring.Options(**fields)
        cQ, cP = ring.NewContextWithParams(N, Q, options=opt), ring.NewContextWithParams(N, P, options=opt)
        plan = ring.CkksPlan(cQ, cP, 1, options=opt)
        key = plan.NewSwitchingKey().set(evk)
        mk = lambda k: cQ.NewPoly(1).set(ops[k])
        out = (cQ.NewPoly(1), cQ.NewPoly(1))
        plan.MulRelin(level, (mk(0), mk(1)), (mk(2), mk(3)), key, out)
        assert np.array_equal(out[0].get(), want[0]) and np.array_equal(out[1].get(), want[1]), fields
    bN, bQ, _, bM = params.bfv_moduli("PN13QP218")
    bQ, bM = list(bQ), list(bM)
    bops = [sampling.uniform_poly(bQ, bN, 1, seed=20 + k) for k in range(4)]
    bwant = oracle.BfvPlan(oracle.Context(bN, bQ), oracle.Context(bN, bM), 65537).mul(np.stack([bops[0][0], bops[1][0]]), np.stack([bops[2][0], bops[3][0]]))
    for fields in ({}, {"bfv_no_gather": 1}, {"bfv_no_ext_epilogue": 1}, {"bfv_gather_below": 1}, {"ext_ieee_div": 1, "ext_narrow": 1}):
        opt = ring.Options(**fields)
        cq, cm = ring.NewContextWithParams(bN, bQ, options=opt), ring.NewContextWithParams(bN, bM, options=opt)
        bplan = ring.BfvPlan(cq, cm, 65537, 1, options=opt)
        mk = lambda k: cq.NewPoly(1).set(bops[k])
        bo = (cq.NewPoly(1), cq.NewPoly(1), cq.NewPoly(1))
        bplan.Mul((mk(0), mk(1)), (mk(2), mk(3)), bo)
        assert all(np.array_equal(bo[k].get(), bwant[k]) for k in range(3)), fields

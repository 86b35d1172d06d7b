"""Parity of the HIP NTT/InvNTT path (through the C ABI) against the reference's golden vectors
and the CPU oracle.  Integer work: the bar is bit-exact."""
import numpy as np
import pytest

from conftest import GOLDEN_SIZES, golden_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", GOLDEN_SIZES)
def test_ntt_reference_vectors(gpu_pkg, n):
    # ring/ntt_test.go:101-142 on the device, every coefficient
    ring = gpu_pkg.ring
    N, moduli, x, want = golden_pair(n)
    ctx = ring.NewContextWithParams(N, moduli)
    p = ctx.NewPoly().set(x)
    ctx.NTT(p, p)
    assert np.array_equal(p.get(), want)
    ctx.InvNTT(p, p)
    assert np.array_equal(p.get(), x)


def test_context_tables_match_oracle(gpu_pkg, oracle):
    N, moduli = gpu_pkg.params.DefaultParamsQi(13)
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    oc = oracle.Context(N, moduli)
    assert np.array_equal(ctx.GetNttPsi(), oc.ntt_psi)
    assert np.array_equal(ctx.GetNttPsiInv(), oc.ntt_psi_inv)
    assert np.array_equal(ctx.GetNttNInv(), oc.n_inv)
    assert np.array_equal(ctx.GetBredParams(), oc.bred)
    assert np.array_equal(ctx.GetMredParams(), oc.mred)
    assert np.array_equal(ctx.GetPsi(), oc.psi_mont)
    assert np.array_equal(ctx.GetPsiInv(), oc.psi_inv_mont)
    assert np.array_equal(ctx.GetRescaleParams(), oc.rescale)


# (15, 11, 3), (15, 5, 9), (14, 9, 3): limb counts whose tables overflow one XCD's L2 and that are no multiple of eight -- the assembly
# kernels' grid x is padded (lr_asm.cpp), workgroups beyond the limbs leave at once
@pytest.mark.parametrize("logn,limbs,batch", [(1, 1, 2), (3, 2, 3), (10, 2, 2), (11, 3, 2), (12, 2, 3), (13, 4, 2),
                                              (14, 8, 2), (15, 16, 2), (16, 3, 2), (15, 11, 3), (15, 5, 9), (14, 9, 3)])
def test_ntt_vs_oracle(gpu_pkg, oracle, logn, limbs, batch):
    N = 1 << logn
    moduli = list(gpu_pkg.params.Qi60()[-limbs:])
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    oc = oracle.Context(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, batch, seed=100 + logn)
    p, r = ctx.NewPoly(batch).set(x), ctx.NewPoly(batch)
    ctx.NTT(p, r)                      # out of place
    got = r.get().reshape(batch, limbs, N)
    for b in range(batch):
        assert np.array_equal(got[b], oc.ntt(x[b])), (logn, b)
    ctx.InvNTT(r, r)                   # in place
    assert np.array_equal(r.get().reshape(batch, limbs, N), x)
    assert np.array_equal(p.get().reshape(batch, limbs, N), x)   # input untouched


@pytest.mark.parametrize("logn", [10, 12, 14, 15, 16])
def test_ntt_accepts_unreduced_and_full_range_input(gpu_pkg, oracle, logn):
    """The reference feeds values >= q into NTT (ring/ring_scaling.go:19,102-105); the result is the
    canonical transform of the input mod q.  Full 64-bit inputs included."""
    N = 1 << logn
    moduli = [gpu_pkg.params.Qi60()[-1], 1099512938497 if logn <= 15 else gpu_pkg.params.ckks_moduli('PN16QP1761')[1][1], gpu_pkg.params.Pi60()[5]]
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    oc = oracle.Context(N, moduli)
    x = gpu_pkg.sampling.random_u64((1, len(moduli), N), seed=logn)
    x[0, :, :4] = np.uint64(0xFFFFFFFFFFFFFFFF)
    x[0, :, 4:8] = 0
    p = ctx.NewPoly().set(x[0])
    ctx.NTT(p, p)
    reduced = np.array([[int(v) % q for v in x[0, i]] for i, q in enumerate(moduli)], dtype=np.uint64)
    want = oc.ntt(reduced)
    assert np.array_equal(p.get(), want)


@pytest.mark.parametrize("logn", [9, 12, 13])
def test_intt_accepts_lazy_range(gpu_pkg, oracle, logn):
    """InvNTT on inputs in [q, 2q) (the reference's safe domain is [0, 2q])."""
    N = 1 << logn
    moduli = list(gpu_pkg.params.Qi60()[-2:])
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    oc = oracle.Context(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, 1, seed=3)[0]
    y = x.copy()
    for i, q in enumerate(moduli):
        y[i, ::2] += np.uint64(q)
    p = ctx.NewPoly().set(y)
    ctx.InvNTT(p, p)
    assert np.array_equal(p.get(), oc.intt(x))
    assert np.array_equal(oc.intt(y), oc.intt(x))


def test_ntt_lvl_touches_only_its_limbs(gpu_pkg, oracle):
    N, moduli = gpu_pkg.params.DefaultParamsQi(13)
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    oc = oracle.Context(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, 2, seed=8)
    p = ctx.NewPoly(2).set(x)
    ctx.NTTLvl(1, p, p)
    got = p.get()
    for b in range(2):
        assert np.array_equal(got[b, :2], oc.ntt(x[b, :2], level=1)[:2])
        assert np.array_equal(got[b, 2:], x[b, 2:])


def test_ntt_limb_under_foreign_modulus(gpu_pkg, oracle):
    """package-level ring.NTT on one limb under another limb's modulus (ring/ring_scaling.go:19)."""
    N, moduli = gpu_pkg.params.DefaultParamsQi(12)
    moduli = moduli + [gpu_pkg.params.Pi60()[0]]
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    oc = oracle.Context(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, 1, seed=4)[0]
    p, r = ctx.NewPoly().set(x), ctx.NewPoly()
    ctx.NTTLimb(2, p, 0, r, 1)     # limb 0 (values < q0, some >= q2) transformed under modulus 2 into row 1
    want = np.empty(N, dtype=np.uint64)
    lib = oracle.lib()
    lib.oc_ntt_limb(x[0].ctypes.data, want.ctypes.data, N, oc.ntt_psi[2].ctypes.data, moduli[2], int(oc.mred[2]),
                    (oracle.u64 * 2)(*[int(v) for v in oc.bred[2]]))
    assert np.array_equal(r.get()[1], want)


def test_go_boundary_limb_slices(gpu_pkg, oracle):
    """lr_poly_upload/download and lr_ntt_host take Go's [][]uint64 shape: one pointer per limb."""
    N, moduli = gpu_pkg.params.DefaultParamsQi(12)
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    oc = oracle.Context(N, moduli)
    x = gpu_pkg.sampling.uniform_poly(moduli, N, 1, seed=21)[0]
    outs = ctx.NTTHost([x[0].copy(), x[1].copy()])
    assert np.array_equal(np.stack(outs), oc.ntt(x))
    back = ctx.InvNTTHost(outs)
    assert np.array_equal(np.stack(back), x)
    p = ctx.NewPoly(2)
    p.set_limb_slices(1, [x[0], x[1]])
    assert np.array_equal(np.stack(p.get_limb_slices(1)), x)
    assert np.array_equal(p.get()[0], np.zeros_like(x))


def test_shape_errors(gpu_pkg):
    ring = gpu_pkg.ring
    N, moduli = gpu_pkg.params.DefaultParamsQi(12)
    ctx = ring.NewContextWithParams(N, moduli)
    short = ctx.NewPolyLvl(0)
    full = ctx.NewPoly()
    with pytest.raises(ring.LatticeRingError) as e:
        ctx.NTT(short, full)           # Go: index out of range panic
    assert e.value.code == 3
    with pytest.raises(ring.LatticeRingError):
        ring.NewContextWithParams(N, [moduli[0] + 2])


def test_full_size_properties_r15(gpu_pkg):
    """BASELINE headline ring R15 (N = 2^15, 16 x 60-bit limbs), batch 8: size-independent properties --
    round trip, linearity, and the convolution theorem against a sparse negacyclic product."""
    N, moduli = gpu_pkg.params.DefaultParamsQi(15)
    ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
    B = 8
    x = gpu_pkg.sampling.uniform_poly(moduli, N, B, seed=1)
    y = gpu_pkg.sampling.uniform_poly(moduli, N, B, seed=2)
    px, py, ps = ctx.NewPoly(B).set(x), ctx.NewPoly(B).set(y), ctx.NewPoly(B)
    ctx.Add(px, py, ps)
    ctx.NTT(px, px)
    ctx.NTT(py, py)
    ctx.NTT(ps, ps)
    chk = ctx.NewPoly(B)
    ctx.Add(px, py, chk)
    assert np.array_equal(chk.get(), ps.get())                       # NTT(x + y) = NTT(x) + NTT(y)
    out = px.get()
    assert all(int(out[:, i].max()) < q for i, q in enumerate(moduli))  # canonical
    ctx.InvNTT(px, px)
    assert np.array_equal(px.get(), x)                               # round trip
    # x * X^k via the NTT == negacyclic shift
    k = 12345
    mono = np.zeros((B, len(moduli), N), dtype=np.uint64)
    mono[:, :, k] = 1
    pm = ctx.NewPoly(B).set(mono)
    ctx.NTT(pm, pm)
    ctx.MForm(pm, pm)
    ctx.NTT(px, px)
    ctx.MulCoeffsMontgomery(px, pm, px)
    ctx.InvNTT(px, px)
    want = np.empty_like(x)
    want[:, :, k:] = x[:, :, :N - k]
    for i, q in enumerate(moduli):
        wrapped = x[:, i, N - k:]
        want[:, i, :k] = np.where(wrapped == 0, 0, np.uint64(q) - wrapped)
    assert np.array_equal(px.get(), want)


def _asm_moduli(pkg, kind, logn):
    """modulus sets that select each variant of the assembly kernels (lr_abi_core.cpp: asm_fwd / asm_inv)"""
    P = pkg.params
    if kind == "qi60":                      # forward variant 1 (q <= 2^60), inverse variant 1
        return list(P.Qi60()[-2:]) + [P.Qi60()[0]]
    if kind == "ckks":                      # forward variant 2 (q < 2^57): CKKS-size moduli
        return [P.GenerateNTTPrimes(50, logn, 1)[0], P.GenerateNTTPrimes(40, logn, 2)[1], P.GenerateNTTPrimes(56, logn, 2)[1]]
    if kind == "bfv60":                     # GenerateNTTPrimes(60): primes just above 2^60 (bfv QiMul) -> variant 0
        ps = P.GenerateNTTPrimes(60, logn, 3)
        assert max(ps) > (1 << 60)
        return ps
    if kind == "mixed":                     # 35-bit next to 60-bit: variant 1 covers both
        return [P.GenerateNTTPrimes(34, logn, 1)[0], P.Qi60()[-1]]
    if kind == "fp":                        # dual kernels, every limb on the FP64 body: 30 bits (below what the integer bodies
        step = 2 << logn                    # take), 40 bits, and the largest NTT prime below 2^46 (tightest range bounds)
        p = (1 << 46) - step + 1
        while not P.is_prime(p):
            p -= step
        return [P.GenerateNTTPrimes(30, logn, 1)[0], P.GenerateNTTPrimes(40, logn, 1)[0], p]
    if kind == "fpedge":                    # dual kernels: both sides of the 2^46 boundary in one launch
        return [P.GenerateNTTPrimes(46, logn, 1)[0], _asm_moduli(pkg, "fp", logn)[2], P.GenerateNTTPrimes(45, logn, 1)[0]]
    raise ValueError(kind)


ASM_CASES = [("qi60", None), ("qi60", "0"), ("ckks", None), ("ckks", "1"), ("ckks", "0"), ("bfv60", None), ("mixed", None),
             ("fp", None), ("fpedge", None)]


@pytest.mark.parametrize("kind,force", ASM_CASES)
@pytest.mark.parametrize("logn", [12, 13, 14, 15, 16])
def test_cxx_and_asm_paths_agree(gpu_pkg, oracle, logn, kind, force, monkeypatch):
    """N = 2^14 / 2^15 with moduli above 2^33 run on the hand-scheduled assembly kernels (three lazy-correction
    variants, chosen from the largest modulus; LR_ASM_VARIANT forces a more conservative one); LR_NO_ASM=1 selects
    the C++ kernel.  All must equal the oracle (full-range inputs)."""
    N = 1 << logn
    moduli = _asm_moduli(gpu_pkg, kind, logn)
    limbs = len(moduli)
    oc = oracle.Context(N, moduli)
    x = gpu_pkg.sampling.random_u64((2, limbs, N), seed=77)
    x[0, :, :5] = np.uint64(0xFFFFFFFFFFFFFFFF)
    want = [oc.ntt(np.array([[int(v) % q for v in x[b, i]] for i, q in enumerate(moduli)], dtype=np.uint64)) for b in range(2)]
    if force is not None:
        monkeypatch.setenv("LR_ASM_VARIANT", force)
    else:
        monkeypatch.delenv("LR_ASM_VARIANT", raising=False)
    # N = 2^15: a launch this small runs as two 2^14 sub-blocks per limb (the "h" kernels behind ntt_top_kernel) unless LR_NTT_SPLIT15=0
    # keeps the one-workgroup-per-transform kernels: both
    for no_asm, split in [(False, "0"), (False, "1"), (True, None)] if logn == 15 else [(False, None), (True, None)]:
        if no_asm:
            monkeypatch.setenv("LR_NO_ASM", "1")
        else:
            monkeypatch.delenv("LR_NO_ASM", raising=False)
        if split is None:
            monkeypatch.delenv("LR_NTT_SPLIT15", raising=False)
        else:
            monkeypatch.setenv("LR_NTT_SPLIT15", split)
        ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
        p, r = ctx.NewPoly(2).set(x), ctx.NewPoly(2)
        ctx.NTT(p, r)
        got = r.get()
        for b in range(2):
            assert np.array_equal(got[b], want[b]), (no_asm, split, b)
        if split is not None:
            assert ("15h" in ctx.last_ntt_kernel()) == (split == "1"), ctx.last_ntt_kernel()
        ctx.NTT(p, p)      # in place
        assert np.array_equal(p.get(), got)
    monkeypatch.delenv("LR_NTT_SPLIT15", raising=False)


@pytest.mark.parametrize("kind,force", ASM_CASES)
@pytest.mark.parametrize("logn", [12, 13, 14, 15, 16])
def test_cxx_and_asm_inverse_paths_agree(gpu_pkg, oracle, logn, kind, force, monkeypatch):
    """inverse twin of the test above: the assembly InvNTT (last stage fused with the N^-1 scaling) and the C++
    kernel both equal the oracle, on inputs anywhere in the documented lazy range [0, 4q), out of place and in place"""
    N = 1 << logn
    moduli = _asm_moduli(gpu_pkg, kind, logn)
    limbs = len(moduli)
    oc = oracle.Context(N, moduli)
    x = gpu_pkg.sampling.random_u64((2, limbs, N), seed=78)
    for i, q in enumerate(moduli):
        x[:, i] %= np.uint64(4 * q)
        x[0, i, :3] = np.uint64(4 * q - 1)
        x[1, i, -3:] = 0
    want = [oc.intt(np.array([[int(v) % q for v in x[b, i]] for i, q in enumerate(moduli)], dtype=np.uint64)) for b in range(2)]
    if force is not None:
        monkeypatch.setenv("LR_ASM_VARIANT", force)
    else:
        monkeypatch.delenv("LR_ASM_VARIANT", raising=False)
    for no_asm, split in [(False, "0"), (False, "1"), (True, None)] if logn == 15 else [(False, None), (True, None)]:
        if no_asm:
            monkeypatch.setenv("LR_NO_ASM", "1")
        else:
            monkeypatch.delenv("LR_NO_ASM", raising=False)
        if split is None:
            monkeypatch.delenv("LR_NTT_SPLIT15", raising=False)
        else:
            monkeypatch.setenv("LR_NTT_SPLIT15", split)
        ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
        p, r = ctx.NewPoly(2).set(x), ctx.NewPoly(2)
        ctx.InvNTT(p, r)
        got = r.get()
        for b in range(2):
            assert np.array_equal(got[b], want[b]), (no_asm, split, b)
        if split is not None:
            assert ("15h" in ctx.last_ntt_kernel()) == (split == "1"), ctx.last_ntt_kernel()
        ctx.InvNTT(p, p)
        assert np.array_equal(p.get(), got)
        # round trip through both assembly kernels
        ctx.NTT(p, p)
        assert np.array_equal(p.get(), np.stack([[x[b, i] % np.uint64(q) for i, q in enumerate(moduli)] for b in range(2)]))


@pytest.mark.parametrize("kind,force,kernel", [("qi60", None, "lr_ntt_inv16f_m1"), ("qi60", "0", "lr_ntt_inv16f_m0"), ("fpedge", None, "lr_ntt_inv16f_m3")])
def test_inverse_2p16_pair_flag_kernels(gpu_pkg, oracle, kind, force, kernel, monkeypatch):
    """N = 2^16 inverse: the two sub-block workgroups of a limb finish in either order and on different XCDs; whichever wave of
    a pair comes second combines both halves (gen_intt.py: fused_last).  A batch large enough that partners run far apart
    (256 polys x 3..4 limbs = 1500..2000 workgroups on 256 CUs), every output compared with the oracle, launched repeatedly (the
    flags are zeroed per launch), out of place and in place; then the same through the separate last-stage pass (LR_NO_INVFUSE)"""
    N = 1 << 16
    moduli = _asm_moduli(gpu_pkg, kind, 16)[:4]
    limbs = len(moduli)
    oc = oracle.Context(N, moduli)
    x = gpu_pkg.sampling.random_u64((2, limbs, N), seed=161)
    for i, q in enumerate(moduli):
        x[:, i] %= np.uint64(4 * q)
    want = [oc.intt(np.array([[int(v) % q for v in x[b, i]] for i, q in enumerate(moduli)], dtype=np.uint64)) for b in range(2)]
    if force is not None:
        monkeypatch.setenv("LR_ASM_VARIANT", force)
    else:
        monkeypatch.delenv("LR_ASM_VARIANT", raising=False)
    B = 256
    tiled = np.concatenate([x] * (B // 2))
    for unfused in (False, True):
        if unfused:
            monkeypatch.setenv("LR_NO_INVFUSE", "1")
        else:
            monkeypatch.delenv("LR_NO_INVFUSE", raising=False)
        ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
        p, r = ctx.NewPoly(B).set(tiled), ctx.NewPoly(B)
        for rep in range(3):
            ctx.InvNTT(p, r)
        assert ctx.last_ntt_kernel() == (kernel.replace("16f", "16s") if unfused else kernel)
        got = r.get()
        for b in range(B):
            assert np.array_equal(got[b], want[b % 2]), (unfused, b)
        ctx.InvNTT(p, p)
        assert np.array_equal(p.get(), got), unfused


@pytest.mark.parametrize("kind", ["qi60", "ckks", "bfv60", "fp"])
def test_asm_2p14_both_plans(gpu_pkg, oracle, kind, monkeypatch):
    """N = 2^14 has two assembly plans: 512 threads / two columns per thread / two workgroups per CU (default) and
    1024 threads / one workgroup per CU (LR_ASM_14_1024=1); both directions of both plans equal the oracle"""
    logn, N = 14, 1 << 14
    moduli = _asm_moduli(gpu_pkg, kind, logn)
    oc = oracle.Context(N, moduli)
    x = gpu_pkg.sampling.random_u64((3, len(moduli), N), seed=123)
    y = x.copy()
    for i, q in enumerate(moduli):
        y[:, i] %= np.uint64(4 * q)
    red = lambda a, b: np.array([[int(v) % q for v in a[b, i]] for i, q in enumerate(moduli)], dtype=np.uint64)
    for wide in (False, True):
        if wide:
            monkeypatch.setenv("LR_ASM_14_1024", "1")
        else:
            monkeypatch.delenv("LR_ASM_14_1024", raising=False)
        ctx = gpu_pkg.ring.NewContextWithParams(N, moduli)
        p, r = ctx.NewPoly(3).set(x), ctx.NewPoly(3)
        ctx.NTT(p, r)
        pi, ri = ctx.NewPoly(3).set(y), ctx.NewPoly(3)
        ctx.InvNTT(pi, ri)
        for b in range(3):
            assert np.array_equal(r.get()[b], oc.ntt(red(x, b))), (wide, b)
            assert np.array_equal(ri.get()[b], oc.intt(red(y, b))), (wide, b)


def test_cpp_host_mirror_runs_reference_ntt_test():
    """tests/cpp/test_ntt_golden.cpp is the C++ twin of ring/ntt_test.go:Test_NTT on include/lattigo_ring.hpp
    (the C++ host mirror over the C ABI).  Built here with g++ against the in-tree library and executed."""
    import os
    import subprocess

    from conftest import GOLDEN_DIR, ROOT
    exe = os.path.join(ROOT, "tests", "cpp", "test_ntt_golden")
    src = os.path.join(ROOT, "tests", "cpp", "test_ntt_golden.cpp")
    libdir = os.path.join(ROOT, "lattigo-fhe-by-go_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), src, "-L" + libdir,
                           "-llattigo_ring_hip", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe, GOLDEN_DIR], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS" in out.stdout


def test_fp64_bodies_are_selected(gpu_pkg, monkeypatch):
    """contexts whose moduli allow it run on the dual kernels (variant 3: FP64 butterflies for the limbs below 2^46); LR_NO_FP=1 and
    a modulus of 57 bits or more keep the integer variants"""
    P, ring = gpu_pkg.params, gpu_pkg.ring
    monkeypatch.delenv("LR_ASM_VARIANT", raising=False)
    monkeypatch.delenv("LR_NO_ASM", raising=False)
    monkeypatch.delenv("LR_NO_FP", raising=False)
    N, Q, Pm = P.ckks_moduli("PN15QP880")
    assert ring.NewContextWithParams(N, Q).ntt_variants() == (3, 3)
    assert ring.NewContextWithParams(N, Pm).ntt_variants() == (2, 1)        # 50-bit moduli only: nothing for the FP body
    assert ring.NewContextWithParams(1 << 13, P.ckks_moduli("PN13QP218")[1]).ntt_variants() == (3, 3)   # 30-bit moduli
    N, moduli = P.DefaultParamsQi(15)
    assert ring.NewContextWithParams(N, moduli).ntt_variants() == (1, 1)
    monkeypatch.setenv("LR_NO_FP", "1")
    N, Q, _ = P.ckks_moduli("PN15QP880")
    assert ring.NewContextWithParams(N, Q).ntt_variants() == (2, 1)


@pytest.mark.parametrize("scheme,name", [("ckks", n) for n in ("PN12QP109", "PN13QP218", "PN14QP438", "PN15QP880", "PN16QP1761")]
                         + [("bfv", n) for n in ("PN12QP109", "PN13QP218", "PN14QP438", "PN15QP880")])
def test_default_parameter_sets_full_size(gpu_pkg, oracle, scheme, name):
    """every default parameter set of ckks/params.go:36-87 and bfv/params.go:47-88 at its full degree, all limbs of Q and P (and
    QMul) in one context each: NTT of full-range input and InvNTT against the oracle (whatever kernel variant the moduli select)"""
    P = gpu_pkg.params
    if scheme == "ckks":
        N, Q, Pm = P.ckks_moduli(name)
        groups = [list(Q) + list(Pm)]
    else:
        N, Q, Pm, QMul = P.bfv_moduli(name)
        groups = [list(Q) + list(Pm), list(QMul)]
    for moduli in groups:
        limbs = len(moduli)
        ctx, oc = gpu_pkg.ring.NewContextWithParams(N, moduli), oracle.Context(N, moduli)
        x = gpu_pkg.sampling.random_u64((1, limbs, N), seed=limbs)
        red = np.array([[int(v) % q for v in x[0, i]] for i, q in enumerate(moduli)], dtype=np.uint64)
        p, r = ctx.NewPoly(1).set(x), ctx.NewPoly(1)
        ctx.NTT(p, r)
        assert np.array_equal(r.get().reshape(1, limbs, N)[0], oc.ntt(red)), (name, ctx.ntt_variants())
        ctx.InvNTT(r, r)
        assert np.array_equal(r.get().reshape(1, limbs, N)[0], red), (name, ctx.ntt_variants())
        y = np.stack([[x[0, i] % np.uint64(4 * q) for i, q in enumerate(moduli)]])
        p.set(y)
        ctx.InvNTT(p, r)
        assert np.array_equal(r.get().reshape(1, limbs, N)[0], oc.intt(np.array([[int(v) % q for v in y[0, i]] for i, q in enumerate(moduli)], dtype=np.uint64)))


@pytest.mark.parametrize("per_wg", [2, 3, 16])
@pytest.mark.parametrize("kind", ["qi60", "ckks", "q61"])
def test_persistent_forward_kernels(gpu_pkg, oracle, kind, per_wg, monkeypatch):
    """LR_NTT_PERSIST: the forward 2^15 kernels that transform several polys per workgroup and prefetch the next poly's column loads
    (lr_ntt_fwd15p_m*; measured no faster than the one-poly kernels at the package power limit and therefore off by default, DESIGN
    3.1): every output against the oracle, with a batch that leaves a short last chunk (7 polys), in place and out of place"""
    if b"diag" not in gpu_pkg._native.lib().lr_build_info():
        pytest.skip("the persistent code objects are part of the diagnostics build only (LR_BUILD_DIAG=1 csrc/build.sh); "
                    "tests/test_options.py checks that the default build refuses the request by name")
    monkeypatch.setenv("LR_NTT_PERSIST", str(per_wg))
    ring, params, sampling = gpu_pkg.ring, gpu_pkg.params, gpu_pkg.sampling
    N = 1 << 15
    if kind == "qi60":
        moduli = list(params.DefaultParamsQi(15)[1][:3])
    elif kind == "ckks":
        moduli = list(params.ckks_moduli("PN15QP880")[1][:4])       # one 50-bit limb (integer body) + three 41-bit ones (FP64 body)
    else:
        moduli = [p for p in params.GenerateNTTPrimes(60, 15, 4) if p > (1 << 60)][:2]
    ctx = ring.NewContextWithParams(N, moduli)
    oc = oracle.Context(N, moduli)
    B = 7
    x = sampling.uniform_poly(moduli, N, B, seed=per_wg)
    src, dst = ctx.NewPoly(B).set(x), ctx.NewPoly(B)
    ctx.NTT(src, dst)
    assert ctx.last_ntt_kernel().startswith("lr_ntt_fwd15p_m"), ctx.last_ntt_kernel()
    got = dst.get()
    for b in range(B):
        assert np.array_equal(got[b], oc.ntt(x[b])), b
    ctx.NTT(src, src)
    assert np.array_equal(src.get(), got)


def test_timeline_build_stamps_every_wave(gpu_pkg, oracle):
    """diagnostics build only: lr_options::ntt_timeline runs the clock-stamping build of the 2^15 kernel -- same results, and 13 increasing
    stamps per wave (lr_context_timeline; tools/timeline.py names the phases)"""
    if b"diag" not in gpu_pkg._native.lib().lr_build_info():
        pytest.skip("the clock-stamping code objects are part of the diagnostics build only (LR_BUILD_DIAG=1 csrc/build.sh)")
    ring, params, sampling = gpu_pkg.ring, gpu_pkg.params, gpu_pkg.sampling
    N, Q = params.DefaultParamsQi(15)
    Q = list(Q[:2])
    ctx = ring.NewContextWithParams(N, Q, options=ring.Options(ntt_timeline=1))
    x = sampling.uniform_poly(Q, N, 2, seed=2)
    src, dst = ctx.NewPoly(2).set(x), ctx.NewPoly(2)
    ctx.NTT(src, dst)
    assert ctx.last_ntt_kernel() == "lr_ntt_fwd15_m1t"
    oc = oracle.Context(N, Q)
    assert np.array_equal(dst.get(), np.stack([oc.ntt(x[b]) for b in range(2)]))
    st = ctx.timeline().astype(np.int64)
    assert st.shape == (4, 16, 16)
    d = (st[:, :, 1:13] - st[:, :, 0:12]) & 0xFFFFFFFF
    assert (d > 0).all() and (d < (1 << 24)).all()

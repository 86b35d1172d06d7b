"""Emulator harness of the assembly NTT kernels (test infrastructure: imports the oracle).

Runs one workgroup of a generated instruction stream (asmgen/gen_ntt.py, gen_intt.py) on the numpy SIMT emulator of
asmgen/isa.py and compares it with the CPU oracle: the same checks the device tests make, without a GPU.

    python tests/asm_emulate.py 15 --selftest [threads]          # forward kernels of 2^15
    python tests/asm_emulate.py 15 --selftest-inverse [threads]
    python tests/asm_emulate.py 16 --selftest | --selftest-inverse   # the 2^15 sub-block kernels of N = 2^16
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lattigo-fhe-by-go_amd", "csrc", "asmgen"))

from gen_ntt import FP_LIMIT, Dual, Gen  # noqa: E402,F401
from isa import Machine, Program  # noqa: E402,F401


# ------------------------------------------------------------------------------------------
# self test on the numpy emulator
# ------------------------------------------------------------------------------------------
def emulate(gen, inverse=False, q=None, geom=None):
    # geom = (x, y, z, hole, group, rows per poly): workgroup ids and the digit-group arguments of NttLaunch
    """run one workgroup of the generated program on the numpy emulator; returns (bit-exact?, summary text)"""
    import numpy as np

    from isa import Machine
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), "..", "..", ".."))
    import __graft_entry__ as graft
    oracle = graft.load_oracle()
    pkg = graft.load_package()
    logn = gen.logn
    N = 1 << logn
    q = q or pkg.params.Qi60()[-3]
    oc = oracle.Context(N, [q])
    x = pkg.sampling.random_u64((N,), seed=5)                 # full 64-bit inputs
    x[:4] = np.uint64(0xFFFFFFFFFFFFFFFF)
    if inverse:
        x = (x % np.uint64(4 * q)).astype(np.uint64)          # the inverse accepts [0, 4q)
        x[:4] = np.uint64(4 * q - 1)
    canon = np.array([[int(val) % q for val in x]], dtype=np.uint64)
    want = (oc.intt(canon) if inverse else oc.ntt(canon))[0]

    # host-side tables exactly as lr_abi_core.cpp builds them
    table = oc.ntt_psi_inv[0] if inverse else oc.ntt_psi[0]
    psi = [int(oracle.inv_mform(int(w), q)) for w in table]
    n_inv = pow(N, -1, q)
    if inverse:
        psi[0] = psi[1] * n_inv % q                           # constant of the fused last stage
    tw = np.zeros((N, 2), dtype=np.uint64)
    for i, w in enumerate(psi):
        tw[i, 0] = w
        tw[i, 1] = (w << 64) // q
    blocks = N // 16
    twf = np.zeros((15, blocks, 2), dtype=np.uint64)
    for cc in range(4):
        for j in range(1 << cc):
            for bk in range(blocks):
                twf[(1 << cc) - 1 + j, bk] = tw[((blocks + bk) << cc) + j]
    qh = (q >> 32) + 1
    g = qh.bit_length() - 1
    red_m = min((1 << (32 + g)) // qh, 0xFFFFFFFF)
    lp = np.zeros(8, dtype=np.uint64)
    lp[0] = q
    lp[2] = (1 << 128) // q >> 64
    lp[5] = n_inv
    lp[6] = (n_inv << 64) // q
    lp[7] = red_m | (g << 32)

    # flat memory image (byte addresses)
    def place(arr, addr):
        words = np.ascontiguousarray(arr).view(np.uint32).ravel()
        mem[addr // 4: addr // 4 + words.size] = words

    gx, gy, gz, hole, group, rows = (geom or (0, 0, 0, 0, 0, 1))[:6]
    item = gx + (hole if gx >= gz * hole else 0)
    where = ((gz * group + gy) * rows + item) * 8 * N          # byte offset of the addressed row
    span = ((gz * group + gy + 1) * rows + 1) * 8 * N
    A_KARG, A_IN = 0x800, 0x1000
    A_OUT = A_IN + span
    A_LP = A_OUT + span
    A_TW = A_LP + 0x1000
    A_TWF = A_TW + 16 * N + 0x1000
    A_FTW = A_TWF + 16 * 15 * blocks + 0x1000
    A_FTWF = A_FTW + 16 * N + 0x1000
    A_FLP = A_FTWF + 16 * 15 * blocks + 0x1000
    A_STAMPS = A_FLP + 0x1000                                   # timeline builds: NttLaunch::epi_x
    mem = np.zeros((A_STAMPS + 0x2000) // 4, dtype=np.uint32)
    place(x, A_IN + where)
    place(lp, A_LP)
    place(tw, A_TW)
    place(twf, A_TWF)
    fp_tables(np, q, n_inv, tw, twf, lambda a, b, c: (place(a, A_FTW), place(b, A_FTWF), place(c, A_FLP)))
    karg = np.zeros(22, dtype=np.uint64)
    karg[13], karg[14], karg[15] = A_FTW - A_TW, A_FTWF - A_TWF, A_FLP
    if getattr(gen, "epi", False):
        # x and plus: canonical polys laid out like the output; c: a random constant in (w, w / q) form on an FP64 limb, as the Shoup
        # pair (w, floor(w 2^64 / q)) on an integer one
        rng = np.random.default_rng(7)
        xe = (rng.integers(0, 1 << 62, N, dtype=np.uint64) % np.uint64(q)).astype(np.uint64)
        pe = (rng.integers(0, 1 << 62, N, dtype=np.uint64) % np.uint64(q)).astype(np.uint64)
        xe[:3], pe[:3] = [0, q - 1, q - 1], [q - 1, 0, q - 1]
        ce = int(rng.integers(1, q))
        A_X, A_P, A_EC = mem.size * 4, mem.size * 4 + span, mem.size * 4 + 2 * span
        mem = np.concatenate([mem, np.zeros((2 * span + 0x1000) // 4, dtype=np.uint32)])
        place(xe, A_X + where)
        place(pe, A_P + where)
        if q < FP_LIMIT and getattr(gen, "gf", None) is not None:      # a dual kernel runs this limb on its FP64 body
            place(np.array([ce, np.float64(ce) / np.float64(q)], dtype=np.float64), A_EC)      # modulus index 0
        else:
            place(np.array([ce, (ce << 64) // q], dtype=np.uint64), A_EC)
        karg[16], karg[17], karg[18], karg[19], karg[20] = A_X, rows * N, A_P, rows * N, A_EC
        want = np.array([((int(a) - int(b)) * ce + int(c)) % q for a, b, c in zip(xe, want, pe)], dtype=np.uint64)
    karg[0], karg[1], karg[2], karg[3] = A_IN, A_OUT, rows * N, rows * N
    karg[4] = 0 | (1 << 32)        # in_limb0, in_limb_step
    karg[5] = 0 | (1 << 32)        # out_limb0, out_limb_step
    karg[6] = 0 | (0 << 32)        # mod0, mod_step: one modulus serves every row of this harness
    karg[11] = 0 | (hole << 32)    # sub_log, hole
    karg[12] = group               # group, pad
    n_items = geom[6] if geom and len(geom) > 6 else gx + 1   # (a seventh entry: fewer items than x + 1 -- a workgroup of the padded grid)
    karg[7] = n_items | (1 << 32)  # n_items, batch
    karg[8], karg[9], karg[10] = A_LP, A_TW, A_TWF
    if getattr(gen, "profile", False):
        karg[16] = A_STAMPS
    place(karg, A_KARG)

    prog = gen.build()
    m = Machine(gen.T, 160 * 1024, mem.size)
    m.mem = mem
    m.vgpr[0] = np.arange(gen.T, dtype=np.uint32)
    m.vdef[0] = True
    m.sgpr[0], m.sgpr[1] = A_KARG, 0
    m.sgpr[gen.WGX.idx], m.sgpr[gen.WGY.idx], m.sgpr[4] = gx, gy, gz
    m.sdef[0:5] = True
    m.count_lds = bool(__import__("os").environ.get("LR_EMU_LDS_STATS"))
    m.run(prog)
    got = m.mem[(A_OUT + where) // 4: (A_OUT + where) // 4 + 2 * N].view(np.uint64)
    if n_items <= gx:                                          # a workgroup of the padded grid: leaves at once, its row stays untouched
        want = np.zeros(N, dtype=np.uint64)
        assert sum(n for op, n in m.executed.items() if op.startswith(("global_store", "ds_"))) == 0
    ok = np.array_equal(got, want)
    if m.count_lds:
        gen.lds_stats = m.lds_stats
        gen.last_prog = prog
    cnt = prog.count()
    valu = sum(n for op, n in cnt.items() if op.startswith("v_"))
    bf = (1 << gen.A) * logn // 2
    info = "%d instructions, %d VALU = %.1f per butterfly, s_nop %d" % (len(prog.ins), valu, valu / bf, cnt.get("s_nop", 0))
    if getattr(gen, "gf", None) is not None:
        ran = sum(n for op, n in m.executed.items() if op.startswith("v_"))
        info = "%s body, %d VALU executed = %.1f per butterfly" % ("FP64" if m.executed.get("v_fma_f64") else "integer", ran, ran / bf)
        ok = ok and bool(m.executed.get("v_fma_f64")) == (q < FP_LIMIT)
    if not ok:
        bad = np.nonzero(got != want)[0]
        info += "\n  mismatches: %d first: %s %s %s" % (bad.size, bad[:8], [hex(int(got[i])) for i in bad[:3]],
                                                       [hex(int(want[i])) for i in bad[:3]])
    return ok, info




def emulate_persist(make_gen, q, polys, per_wg, chunk):
    """forward persistent kernel: the workgroup of chunk `chunk` transforms polys [chunk * per_wg, min(polys, (chunk + 1) * per_wg)) of a
    batch of `polys` one-limb polys in its loop; every output row of the batch is compared (rows of other chunks must stay zero)"""
    import numpy as np

    from isa import Machine
    import __graft_entry__ as graft
    oracle = graft.load_oracle()
    pkg = graft.load_package()
    gen = make_gen()
    logn = gen.logn
    N = 1 << logn
    q = q or pkg.params.Qi60()[-3]
    oc = oracle.Context(N, [q])
    x = pkg.sampling.random_u64((polys, N), seed=21)
    x[:, :4] = np.uint64(0xFFFFFFFFFFFFFFFF)
    want = np.zeros((polys, N), dtype=np.uint64)
    lo, hi = chunk * per_wg, min(polys, (chunk + 1) * per_wg)
    for b in range(lo, hi):
        want[b] = oc.ntt(np.array([[int(val) % q for val in x[b]]], dtype=np.uint64))[0]
    psi = [int(oracle.inv_mform(int(w), q)) for w in oc.ntt_psi[0]]
    n_inv = pow(N, -1, q)
    tw = np.zeros((N, 2), dtype=np.uint64)
    for i, w in enumerate(psi):
        tw[i, 0] = w
        tw[i, 1] = (w << 64) // q
    blocks = N // 16
    twf = np.zeros((15, blocks, 2), dtype=np.uint64)
    for cc in range(4):
        for j in range(1 << cc):
            for bk in range(blocks):
                twf[(1 << cc) - 1 + j, bk] = tw[((blocks + bk) << cc) + j]
    lp = np.zeros(8, dtype=np.uint64)
    lp[0] = q
    lp[2] = (1 << 128) // q >> 64
    lp[5] = n_inv
    lp[6] = (n_inv << 64) // q
    stride_in, stride_out = 2 * N, 3 * N          # polys further apart than their rows, and differently on the two sides
    A_KARG, A_IN = 0x800, 0x1000
    A_OUT = A_IN + 8 * stride_in * polys + 0x1000
    A_LP = A_OUT + 8 * stride_out * polys + 0x1000
    A_TW = A_LP + 0x1000
    A_TWF = A_TW + 16 * N + 0x1000
    A_FTW = A_TWF + 16 * 15 * blocks + 0x1000
    A_FTWF = A_FTW + 16 * N + 0x1000
    A_FLP = A_FTWF + 16 * 15 * blocks + 0x1000
    A_STAMPS = A_FLP + 0x1000
    mem = np.zeros((A_STAMPS + 0x4000) // 4, dtype=np.uint32)

    def place(arr, addr):
        words = np.ascontiguousarray(arr).view(np.uint32).ravel()
        mem[addr // 4: addr // 4 + words.size] = words

    for b in range(polys):
        place(x[b], A_IN + 8 * stride_in * b)
    place(lp, A_LP)
    place(tw, A_TW)
    place(twf, A_TWF)
    fp_tables(np, q, n_inv, tw, twf, lambda a, b, c: (place(a, A_FTW), place(b, A_FTWF), place(c, A_FLP)))
    karg = np.zeros(22, dtype=np.uint64)
    karg[13], karg[14], karg[15] = A_FTW - A_TW, A_FTWF - A_TWF, A_FLP
    karg[0], karg[1], karg[2], karg[3] = A_IN, A_OUT, stride_in, stride_out
    karg[4] = 0 | (1 << 32)
    karg[5] = 0 | (1 << 32)
    karg[6] = 0
    karg[7] = 1 | (polys << 32)        # n_items, batch
    karg[8], karg[9], karg[10] = A_LP, A_TW, A_TWF
    karg[11] = 0                       # sub_log, hole
    karg[12] = polys | (per_wg << 32)  # group (= batch for a plain launch), fuse_top = polys per workgroup
    if getattr(gen, "profile", False):
        karg[16] = A_STAMPS
    place(karg, A_KARG)
    prog = gen.build()
    m = Machine(gen.T, 160 * 1024, mem.size)
    m.mem = mem
    m.vgpr[0] = np.arange(gen.T, dtype=np.uint32)
    m.vdef[0] = True
    m.sgpr[0], m.sgpr[1] = A_KARG, 0
    m.sgpr[gen.WGX.idx], m.sgpr[gen.WGY.idx], m.sgpr[4] = 0, chunk, 0
    m.sdef[0:5] = True
    m.run(prog)
    got = np.stack([m.mem[(A_OUT + 8 * stride_out * b) // 4: (A_OUT + 8 * stride_out * b) // 4 + 2 * N].view(np.uint64) for b in range(polys)])
    ok = bool(np.array_equal(got, want))
    ran = sum(n for op, n in m.executed.items() if op.startswith("v_"))
    info = "%d polys in the loop, %d VALU executed = %.1f per poly, at most %d vector-memory operations in flight" % (
        hi - lo, ran, ran / (hi - lo), m.max_vm_outstanding)
    if getattr(gen, "gf", None) is not None:
        ok = ok and bool(m.executed.get("v_fma_f64")) == (q < FP_LIMIT)
    if not ok:
        bad = np.argwhere(got != want)
        info += "\n  mismatches: %d first: %s" % (len(bad), bad[:4].tolist())
    return ok, info


def selftest_persist(threads=1024):
    ok = True
    cases = []
    for mode in (0, 1, 2):
        cases.append(("mode %d" % mode, (lambda mode=mode: Gen(15, mode, threads, persist=True)), test_moduli(15, mode)[0]))
    for q in fp_test_moduli(15):
        cases.append(("dual", (lambda: Dual(lambda fp: Gen(15, 2, threads, fp=fp, dual=True, persist=True))), q))
    cases.append(("dual timeline", (lambda: Dual(lambda fp: Gen(15, 2, threads, fp=fp, dual=True, persist=True, profile=True))), fp_test_moduli(15)[0]))
    for name, make, q in cases:
        # three polys in one loop (prefetch taken twice, skipped once); a trailing chunk of one poly (no prefetch at all)
        for polys, per_wg, chunk in ((3, 3, 0), (3, 2, 1)):
            good, info = emulate_persist(make, q, polys, per_wg, chunk)
            ok = ok and good
            print("forward persistent logN=15 %s q=%d (%d bits), chunk %d of %d x %d: %s; %s" % (
                name, q, q.bit_length(), chunk, polys, per_wg, "bit-exact vs oracle" if good else "MISMATCH", info), flush=True)
    return ok


def fp_tables(np, q, n_inv, tw, twf, put):
    """what lr_abi_core.cpp builds for the FP body: every (w, floor(w 2^64 / q)) becomes the pair of doubles (w, RN(w / q));
    FpLimb = {q, RN(1/q), N^-1 mod q, RN(N^-1 / q)}, all zero for a modulus the FP body does not take"""
    def conv(t):
        w = t[..., 0].astype(np.float64)
        out = np.zeros(t.shape, dtype=np.float64)
        out[..., 0] = w
        out[..., 1] = w / np.float64(q)
        return out
    flp = np.zeros(4, dtype=np.float64)
    if q < FP_LIMIT:
        flp[:] = [q, np.float64(1.0) / np.float64(q), n_inv, np.float64(n_inv) / np.float64(q)]
        put(conv(tw), conv(twf), flp)
    else:
        put(np.zeros(tw.shape), np.zeros(twf.shape), flp)


def test_moduli(logn, mode):
    """moduli at both ends of the range the mode accepts"""
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), "..", "..", ".."))
    import __graft_entry__ as graft
    params = graft.load_package().params
    lo = params.GenerateNTTPrimes(34, logn, 1)[0]
    if mode == 2:
        return [lo, params.GenerateNTTPrimes(56, logn, 2)[1]]
    if mode == 1:
        return [params.Qi60()[-3], params.Qi60()[0], lo]
    above = [p for p in params.GenerateNTTPrimes(60, logn, 4) if p > (1 << 60)]
    return [above[-1], lo]


def fp_test_moduli(logn):
    """dual kernels: the largest NTT prime below 2^46 (the FP body's range bounds at their tightest), a 30-bit one (below what the
    integer bodies accept) and one just above 2^46 (integer body of the same kernel)"""
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), "..", "..", ".."))
    import __graft_entry__ as graft
    params = graft.load_package().params
    step = 2 << logn
    p = FP_LIMIT - step + 1
    while not params.is_prime(p):
        p -= step
    return [p, params.GenerateNTTPrimes(30, logn, 1)[0], params.GenerateNTTPrimes(46, logn, 1)[0]]


def emulate_sub(make_gen, inverse, q, pretop=False, order=(0, 1)):
    """N = 2^16 through the two sub-block workgroups of one limb; returns (bit-exact?, summary)"""
    import numpy as np

    from isa import Machine
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), "..", "..", ".."))
    import __graft_entry__ as graft
    oracle = graft.load_oracle()
    pkg = graft.load_package()
    N = 1 << make_gen().logn
    NF = 2 * N
    oc = oracle.Context(NF, [q])
    x = pkg.sampling.random_u64((NF,), seed=9)
    x[:4] = np.uint64(0xFFFFFFFFFFFFFFFF)
    if inverse:
        x = (x % np.uint64(4 * q)).astype(np.uint64)
    canon = np.array([[int(val) % q for val in x]], dtype=np.uint64)
    want = (oc.intt(canon) if inverse else oc.ntt(canon))[0]
    table = oc.ntt_psi_inv[0] if inverse else oc.ntt_psi[0]
    psi = [int(oracle.inv_mform(int(w), q)) for w in table]
    n_inv = pow(NF, -1, q)
    psi[0] = psi[1] * n_inv % q if inverse else q - psi[1]     # lr_abi_core.cpp fills the unused heap entry 0 like this
    tw = np.zeros((NF, 2), dtype=np.uint64)
    for i, w in enumerate(psi):
        tw[i, 0] = w
        tw[i, 1] = (w << 64) // q
    blocks = NF // 16
    twf = np.zeros((15, blocks, 2), dtype=np.uint64)
    for cc in range(4):
        for j in range(1 << cc):
            for bk in range(blocks):
                twf[(1 << cc) - 1 + j, bk] = tw[((blocks + bk) << cc) + j]
    lp = np.zeros(8, dtype=np.uint64)
    lp[0] = q
    lp[2] = (1 << 128) // q >> 64
    lp[5] = n_inv
    lp[6] = (n_inv << 64) // q
    A_KARG, A_IN = 0x800, 0x1000
    A_OUT = A_IN + 8 * NF + 0x1000
    A_LP = A_OUT + 8 * NF + 0x1000
    A_TW = A_LP + 0x1000
    A_TWF = A_TW + 16 * NF + 0x1000
    A_FTW = A_TWF + 16 * 15 * blocks + 0x1000
    A_FTWF = A_FTW + 16 * NF + 0x1000
    A_FLP = A_FTWF + 16 * 15 * blocks + 0x1000
    mem = np.zeros((A_FLP + 0x1000) // 4, dtype=np.uint32)

    def place(arr, addr):
        words = np.ascontiguousarray(arr).view(np.uint32).ravel()
        mem[addr // 4: addr // 4 + words.size] = words

    if pretop:
        # what ntt_top_kernel leaves for the plain forward sub-blocks: X = U + V*psi[1], Y = U - V*psi[1] (lazy, < 8q)
        xs = [int(a) % q for a in x]
        w = psi[1]
        x = np.array([(xs[j] + xs[j + N] * w) % q + q for j in range(N)] + [(xs[j] - xs[j + N] * w) % q + 3 * q for j in range(N)],
                     dtype=np.uint64)
    place(x, A_IN)
    place(lp, A_LP)
    place(tw, A_TW)
    place(twf, A_TWF)
    fp_tables(np, q, n_inv, tw, twf, lambda a, b, c: (place(a, A_FTW), place(b, A_FTWF), place(c, A_FLP)))
    karg = np.zeros(22, dtype=np.uint64)
    karg[13], karg[14], karg[15] = A_FTW - A_TW, A_FTWF - A_TWF, A_FLP
    karg[0], karg[1], karg[2], karg[3] = A_IN, A_OUT, NF, NF
    karg[4] = 0 | (1 << 32)
    karg[5] = 0 | (1 << 32)
    karg[6] = 0
    karg[7] = 1 | (1 << 32)
    karg[8], karg[9], karg[10] = A_LP, A_TW, A_TWF
    karg[11] = 1                   # sub_log = 1, hole = 0
    karg[12] = 1                   # group
    fuse_last = getattr(make_gen(), "fuse_last", False)
    if fuse_last:
        # inverse with the fused last stage: the launch's pair flags (one u32 per wave of a limb's two sub-blocks), zeroed by the host
        A_FLAGS = mem.size * 4
        mem = np.concatenate([mem, np.zeros(0x1000 // 4, dtype=np.uint32)])
        karg[16] = A_FLAGS
    if getattr(make_gen(), "epi", False):
        # the epilogue of the plain forward sub-blocks: out = (x - NTT(in)) * c + plus, x / plus laid out like the output
        rng = np.random.default_rng(11)
        xe = (rng.integers(0, 1 << 62, NF, dtype=np.uint64) % np.uint64(q)).astype(np.uint64)
        pe = (rng.integers(0, 1 << 62, NF, dtype=np.uint64) % np.uint64(q)).astype(np.uint64)
        xe[:3], pe[:3] = [0, q - 1, q - 1], [q - 1, 0, q - 1]
        xe[N:N + 3], pe[N:N + 3] = [q - 1, 0, q - 1], [0, q - 1, q - 1]
        ce = int(rng.integers(1, q))
        A_X = mem.size * 4
        A_P, A_EC = A_X + 8 * NF + 0x1000, A_X + 16 * NF + 0x2000
        mem = np.concatenate([mem, np.zeros((16 * NF + 0x3000) // 4, dtype=np.uint32)])
        place(xe, A_X)
        place(pe, A_P)
        if q < FP_LIMIT and getattr(make_gen(), "gf", None) is not None:
            place(np.array([ce, np.float64(ce) / np.float64(q)], dtype=np.float64), A_EC)
        else:
            place(np.array([ce, (ce << 64) // q], dtype=np.uint64), A_EC)
        karg[16], karg[17], karg[18], karg[19], karg[20] = A_X, NF, A_P, NF, A_EC
        want = np.array([((int(a) - int(b)) * ce + int(c)) % q for a, b, c in zip(xe, want, pe)], dtype=np.uint64)
    place(karg, A_KARG)
    info = ""
    for blk in order:
        gen = make_gen()
        prog = gen.build()
        m = Machine(gen.T, 160 * 1024, mem.size)
        m.mem = mem
        m.vgpr[0] = np.arange(gen.T, dtype=np.uint32)
        m.vdef[0] = True
        m.sgpr[0], m.sgpr[1] = A_KARG, 0
        m.sgpr[gen.WGX.idx], m.sgpr[gen.WGY.idx], m.sgpr[4] = blk, 0, 0
        m.sdef[0:5] = True
        m.run(prog)
        mem = m.mem
        cnt = prog.count()
        info = "%d instructions, %d VALU" % (len(prog.ins), sum(n for op, n in cnt.items() if op.startswith("v_")))
    got = mem[A_OUT // 4: A_OUT // 4 + 2 * NF].view(np.uint64).copy()
    if fuse_last:
        assert not mem[A_FLAGS // 4: A_FLAGS // 4 + 0x400].any(), "the pair flags must be back at zero"
    elif inverse:
        # what ntt_top_kernel does next: last Gentleman-Sande stage and the scaling
        U = [int(a) for a in got[:N]]
        V = [int(a) for a in got[N:]]
        assert max(max(U), max(V)) < 8 * q
        w1n = psi[0]
        got = np.array([(u + v) * n_inv % q for u, v in zip(U, V)] + [(u - v) * w1n % q for u, v in zip(U, V)], dtype=np.uint64)
    ok = bool(np.array_equal(got, want))
    return ok, info


def GenInv_(*a, **k):
    from gen_intt import GenInv
    return GenInv(*a, **k)


def selftest(logn, inverse=False, threads=1024):
    ok = True
    for mode in ((0, 1) if inverse else (0, 1, 2)):
        for q in test_moduli(logn, mode):
            if inverse:
                from gen_intt import GenInv
                gen = GenInv(logn, mode, threads)
            else:
                gen = Gen(logn, mode, threads)
            # the last modulus of each mode also exercises the digit-group addressing (grid z, skipped limbs)
            geom = (2, 1, 1, 2, 3, 6) if q == test_moduli(logn, mode)[-1] else None
            good, info = emulate(gen, inverse, q, geom)
            ok = ok and good
            print("%s logN=%d T=%d mode %d q=%d (%d bits): %s; %s" % ("inverse" if inverse else "forward", logn, threads, mode, q, q.bit_length(),
                                                             "bit-exact vs oracle" if good else "MISMATCH", info), flush=True)
    # grid x padded to a multiple of eight (lr_asm.cpp): the workgroups beyond n_items store nothing (integer kernels: x = limb)
    for mk in ([lambda: GenInv_(logn, 1, threads)] if inverse else [lambda: Gen(logn, 1, threads), lambda: Gen(logn, 1, threads, epi=True)]):
        good, info = emulate(mk(), inverse, test_moduli(logn, 1)[0], (3, 1, 0, 0, 0, 4, 3))
        ok = ok and good
        print("%s logN=%d T=%d padding workgroup: %s; %s" % ("inverse" if inverse else "forward", logn, threads, "stores nothing" if good else "WROTE", info), flush=True)
    for q in fp_test_moduli(logn):
        if inverse:
            from gen_intt import GenInv
            gen = Dual(lambda fp: GenInv(logn, 1, threads, fp=fp, dual=True))
        else:
            gen = Dual(lambda fp: Gen(logn, 2, threads, fp=fp, dual=True))
        geom = (2, 1, 1, 2, 3, 6) if q == fp_test_moduli(logn)[0] else None
        good, info = emulate(gen, inverse, q, geom)
        ok = ok and good
        print("%s logN=%d T=%d dual q=%d (%d bits): %s; %s" % ("inverse" if inverse else "forward", logn, threads, q, q.bit_length(),
                                                        "bit-exact vs oracle" if good else "MISMATCH", info), flush=True)
    if not inverse:
        # the integer epilogue: the pure integer kernel of mode 1 ("m5") at both ends of its modulus range, and the dual kernels'
        # integer body on a modulus just above 2^46
        for q, mk in [(qq, (lambda: Gen(logn, 1, threads, epi=True))) for qq in (test_moduli(logn, 1)[0], test_moduli(logn, 1)[-1])] + \
                     [(fp_test_moduli(logn)[2], (lambda: Dual(lambda fp: Gen(logn, 2, threads, fp=fp, dual=True, epi=True))))]:
            for geom in (None, (2, 1, 1, 2, 3, 6)):
                good, info = emulate(mk(), False, q, geom)
                ok = ok and good
                print("forward logN=%d T=%d integer epilogue q=%d (%d bits): %s; %s" % (logn, threads, q, q.bit_length(),
                                                                                  "bit-exact vs oracle" if good else "MISMATCH", info), flush=True)
        # the epilogue kernels: out = (x - NTT(in)) * c + plus on the FP64 body, plain and with the digit-group addressing
        for q in fp_test_moduli(logn)[:2]:
            for geom in (None, (2, 1, 1, 2, 3, 6)):
                good, info = emulate(Dual(lambda fp: Gen(logn, 2, threads, fp=fp, dual=True, epi=True)), False, q, geom)
                ok = ok and good
                print("forward logN=%d T=%d epilogue q=%d (%d bits): %s; %s" % (logn, threads, q, q.bit_length(),
                                                                          "bit-exact vs oracle" if good else "MISMATCH", info), flush=True)
    return ok


def selftest_sub(inverse=False):
    """the sub-block kernels of N = 2^16"""
    ok = True
    for mode in ((0, 1) if inverse else (0, 1, 2)):
        q = test_moduli(16, mode)[0]
        if inverse:
            from gen_intt import GenInv
            make = lambda: GenInv(15, mode, 1024, sub=True)
        else:
            make = lambda: Gen(15, mode, 1024, sub=True)
        good, info = emulate_sub(make, inverse, q)
        if inverse:
            # the pair-flag kernels (last stage by whichever sub-block finishes second), both orders
            for order in ((0, 1), (1, 0)):
                g2, _ = emulate_sub(lambda: GenInv(15, mode, 1024, sub=True, fuse_last=True), True, q, order=order)
                good = good and g2
        if not inverse:
            # the plain variant continues from the output of the separate top-stage pass
            good2, _ = emulate_sub(lambda: Gen(15, mode, 1024, sub=True, fused=False), inverse, q, pretop=True)
            good = good and good2
        ok = ok and good
        print("%s N=2^16 sub-blocks mode %d q=%d (%d bits): %s; %s" % ("inverse" if inverse else "forward", mode, q, q.bit_length(),
                                                                    "bit-exact vs oracle" if good else "MISMATCH", info), flush=True)
    if not inverse:
        # the integer epilogue on the sub-block kernels: pure integer ("m5", fused top stage and plain), and the dual kernels' integer body
        q1 = test_moduli(16, 1)[0]
        for label, mk, pre in (("m5 fused-top", (lambda: Gen(15, 1, 1024, sub=True, epi=True)), False),
                               ("m5 plain", (lambda: Gen(15, 1, 1024, sub=True, fused=False, epi=True)), True)):
            good, info = emulate_sub(mk, False, q1, pretop=pre)
            ok = ok and good
            print("forward N=2^16 sub-blocks %s q=%d (%d bits): %s; %s" % (label, q1, q1.bit_length(), "bit-exact vs oracle" if good else "MISMATCH", info), flush=True)
        q2 = fp_test_moduli(16)[2]
        good, info = emulate_sub(lambda: Dual(lambda fp: Gen(15, 2, 1024, sub=True, fused=False, fp=fp, dual=True, epi=True)), False, q2, pretop=True)
        ok = ok and good
        print("forward N=2^16 sub-blocks dual integer-body epilogue q=%d (%d bits): %s; %s" % (q2, q2.bit_length(), "bit-exact vs oracle" if good else "MISMATCH", info), flush=True)
    for q in fp_test_moduli(16)[:2]:
        if inverse:
            from gen_intt import GenInv
            good, info = emulate_sub(lambda: Dual(lambda fp: GenInv(15, 1, 1024, sub=True, fp=fp, dual=True)), True, q)
            for order in ((0, 1), (1, 0)):
                g2, _ = emulate_sub(lambda: Dual(lambda fp: GenInv(15, 1, 1024, sub=True, fp=fp, dual=True, fuse_last=True)), True, q, order=order)
                good = good and g2
        else:
            good, info = emulate_sub(lambda: Dual(lambda fp: Gen(15, 2, 1024, sub=True, fp=fp, dual=True)), False, q)
            good2, _ = emulate_sub(lambda: Dual(lambda fp: Gen(15, 2, 1024, sub=True, fused=False, fp=fp, dual=True)), False, q, pretop=True)
            good3, _ = emulate_sub(lambda: Dual(lambda fp: Gen(15, 2, 1024, sub=True, fused=False, fp=fp, dual=True, epi=True)), False, q, pretop=True)
            good2 = good2 and good3
            good = good and good2
        ok = ok and good
        print("%s N=2^16 sub-blocks dual q=%d (%d bits): %s; %s" % ("inverse" if inverse else "forward", q, q.bit_length(),
                                                             "bit-exact vs oracle" if good else "MISMATCH", info), flush=True)
    return ok


def selftest_halves(inverse=False):
    """N = 2^15 as two 2^14 sub-blocks (small launches: twice the workgroups, half the latency): the plain forward sub-block kernels
    ("h": the stage over bit 14 applied before, by the basis extension or ntt_top_kernel) in every mode incl. the epilogues, the lazy inverse ones"""
    ok = True
    sub_q = lambda mode: test_moduli(15, mode)[0]
    cases = []
    if inverse:
        for mode in (0, 1):
            cases.append(("mode %d" % mode, (lambda mode=mode: GenInv_(14, mode, 1024, sub=True)), sub_q(mode)))
        for q in fp_test_moduli(15):
            cases.append(("dual", (lambda: Dual(lambda fp: GenInv_(14, 1, 1024, sub=True, fp=fp, dual=True))), q))
    else:
        for mode in (0, 1, 2):
            cases.append(("mode %d" % mode, (lambda mode=mode: Gen(14, mode, 1024, sub=True, fused=False)), sub_q(mode)))
        cases.append(("m5", (lambda: Gen(14, 1, 1024, sub=True, fused=False, epi=True)), sub_q(1)))
        for q in fp_test_moduli(15):
            cases.append(("dual", (lambda: Dual(lambda fp: Gen(14, 2, 1024, sub=True, fused=False, fp=fp, dual=True))), q))
            cases.append(("dual epilogue", (lambda: Dual(lambda fp: Gen(14, 2, 1024, sub=True, fused=False, fp=fp, dual=True, epi=True))), q))
    for label, mk, q in cases:
        good = True
        for order in ((0, 1), (1, 0)):
            g, info = emulate_sub(mk, inverse, q, pretop=not inverse, order=order)
            good = good and g
        ok = ok and good
        print("%s N=2^15 halves %s q=%d (%d bits): %s; %s" % ("inverse" if inverse else "forward", label, q, q.bit_length(),
                                                          "bit-exact vs oracle" if good else "MISMATCH", info), flush=True)
    return ok


if __name__ == "__main__":
    logn = int(sys.argv[1])
    inverse = len(sys.argv) > 2 and sys.argv[2] == "--selftest-inverse"
    if len(sys.argv) > 2 and sys.argv[-1] == "--halves":
        sys.exit(0 if selftest_halves(inverse=inverse) else 1)
    if logn == 16:
        sys.exit(0 if selftest_sub(inverse=inverse) else 1)
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    if len(sys.argv) > 2 and sys.argv[2] == "--selftest-persist":
        sys.exit(0 if selftest_persist() else 1)
    sys.exit(0 if selftest(logn, inverse=inverse, threads=threads) else 1)

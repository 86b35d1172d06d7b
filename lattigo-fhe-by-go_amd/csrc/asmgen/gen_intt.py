"""Generator of the hand-scheduled inverse NTT kernels for gfx950 (N = 2^14, 2^15; 1024 threads).

Mirror of gen_ntt.py (same LDS image, same twiddle tables, same wave-local LDS phase), Gentleman-Sande
order: copy-in -> stages over bits 0..3 (lane-transposed table) -> bits 4..6 (per-lane twiddles) ->
bits 7..9 (wave-uniform twiddles) -> column exchange -> the top stages in registers with SGPR twiddles.
The last stage is fused with the N^-1 scaling of ring/ntt.go:136-138: its outputs are
(U+V)*N^-1 and (U-V)*(psi_inv[1]*N^-1), the second constant lives at index 0 of the inverse table.

Lazy ranges (2^57 <= q <= 2^60): a stage over bit b leaves sums ("X type", index bit b clear) below 8q
and products ("Y type") below 4q.  Both operands of a butterfly have the same type, so only the
butterflies that add two X-type values need the conditional subtraction of 8q: half of them, known at
generation time inside a pass and applied unconditionally on the first stage of a pass.
15 VALU instructions per butterfly without correction, 19 with.  Inputs must be below 4q.

    python gen_intt.py 15 out.s
    python ../../../tests/asm_emulate.py 15 --selftest-inverse      (emulator check against the oracle)
"""
import sys

from gen_ntt import CROSS32, T, Dual, Gen, kernel_text_for
from isa import EXEC, Neg, s, v


class GenInv(Gen):
    """mode 1: every modulus in (2^33, 2^60], bound B = 8q as described above;
       mode 0: moduli up to 2^61: B = 4q and every butterfly corrects."""

    def __init__(self, logn, mode=1, threads=1024, sub=False, fp=False, dual=False, fuse_last=False, profile=False, pre1=None):
        assert mode in (0, 1)
        super().__init__(logn, mode, threads, sub, fp=fp, dual=dual, profile=profile)
        self.karg_parked = profile
        # pre1 (plans with two LDS images, one column per thread): the copy-in loads of half 1 are issued during half 0's LDS phase
        # into v32..v63, which are idle until half 0's results are parked there -- the second copy-in's memory latency (10 k of a
        # wave's 62 k clocks on the FP64 body, profiles/r03/timeline_inv15_ckks.json) hides behind arithmetic.  Half 0's columns are
        # read back into v0..v31 first and move to v32..v63 once half 1's raw data has left for the LDS image (32 v_mov_b32).
        self.pre1 = (self.HALVES == 2 and self.C == 1) if pre1 is None else pre1
        assert not self.pre1 or (self.HALVES == 2 and self.C == 1)
        # fuse_last (sub-block kernels of N = 2^16): no ntt_top_kernel pass afterwards.  Every wave stores its lazy rows, makes them
        # visible to the device and bumps the flag it shares with the same wave of the limb's other sub-block; the wave that finds
        # the flag already bumped (its partner's rows are complete) loads them and finishes both halves: last stage + scaling.
        # Nothing ever waits for the other block, so the order in which the two run does not matter.
        assert not fuse_last or sub
        self.fuse_last = fuse_last
        # FP64 body: inputs below 2^52 (the contract is < 4q, q < 2^46).  A Gentleman-Sande stage doubles the bound of its
        # sums; products come back within q.  With B the bound in units of q (4 on entry), a stage needs B <= 16 (|U - V| <=
        # 2^51 keeps the quotient estimate within one); the sums of the stage that would leave B = 32 are reduced to q/2.
        self.fp_reduce = set()
        bound = 4
        for g in range(logn):
            bound *= 2
            if bound > 16:
                self.fp_reduce.add(g)
                bound = 1
        self.stage_base = 0
        self.Q8 = self.Q4                       # s[18:19] holds the bound B here
        self.NQ2 = s(22, 2)                     # -2q (the Barrett constant is not used by the inverse)
        self.LP = s(0, 2)                       # LimbParams pointer (kernarg pointer is dead after the prologue)
        # pass A keeps X[k]; the LDS phase works in v0..v31, so the first half is parked in v32..v63
        if self.HALVES == 2:
            for c in range(self.C):
                for k in range(self.RA):
                    h, r = divmod(k, self.SPH)
                    self.X[c * self.RA + k] = v(32 * (1 - h) + 2 * (c * self.SPH + r), 2)

    # ------------------------------------------------------------------ arithmetic
    def ops_modmul_inplace(self, ts, V, tw):
        """V <- V * w - qhat * q (lazy, [0,4q)) for any 64-bit V"""
        w0, w1, s0, s1 = tw
        J = self.JUNK
        if CROSS32:      # gen_ntt.py: the cross terms by 32-bit low products and three-input adds
            return [
                ("v_mul_hi_u32", ts.T0, V.hi(), s0),
                ("v_mul_hi_u32", ts.T2, V.lo(), s1),
                ("v_mad_u64_u32", ts.Q, J, V.hi(), s1, ts.T01),
                ("v_mul_lo_u32", ts.C.lo(), V.lo(), w1),
                ("v_lshl_add_u64", ts.Q, ts.Q, 0, ts.T23),
                ("v_mul_lo_u32", ts.C.hi(), V.hi(), w0),
                ("v_mad_u64_u32", V, J, V.lo(), w0, 0),
                ("v_mad_u64_u32", V, J, ts.Q.lo(), self.NQ.lo(), V),
                ("v_mul_lo_u32", ts.T0, ts.Q.lo(), self.NQ.hi()),
                ("v_mul_lo_u32", ts.T2, ts.Q.hi(), self.NQ.lo()),
                ("v_add3_u32", ts.C.lo(), ts.C.lo(), ts.C.hi(), ts.T0),
                ("v_add3_u32", V.hi(), V.hi(), ts.C.lo(), ts.T2),
            ]
        return self.ops_mulconst(ts, V, tw)

    def ops_sum_diff(self, ts, U, V):
        """(U, V) <- (U + V, U + 8q - V)"""
        return [("v_lshl_add_u64", ts.R, U, 0, self.Q8),
                ("v_lshl_add_u64", U, U, 0, V),
                ("v_sub_co_u32", V.lo(), ts.CY, ts.R.lo(), V.lo()),
                ("v_subb_co_u32", V.hi(), ts.CY, ts.R.hi(), V.hi(), ts.CY)]

    def ops_butterfly_gs_fp(self, ts, U, V, tw, reduce):
        ops = [("v_add_f64", ts.T23, U, Neg(V)),
               ("v_add_f64", U, U, V)]
        ops += self.ops_modmul_fp(ts, ts.T23, tw, dst=V)
        if reduce:
            ops += [("v_mul_f64", ts.T01, U, self.QINV),
                    ("v_rndne_f64", ts.T01, ts.T01),
                    ("v_fma_f64", U, Neg(ts.T01), self.QD, U)]
        return ops

    def ops_ingest_small(self, ts, X):
        """an integer below 2^52 -> the same value as a double: 2^52 + x has x as its mantissa"""
        return [("v_or_b32", X.hi(), 0x43300000, X.hi()),
                ("v_add_f64", X, X, Neg(self.MAGIC))]

    def ops_butterfly(self, ts, U, V, tw, correct):
        """(U, V) <- (U + V [- 8q], (U + 8q - V) * w)"""
        if self.fp:
            return self.ops_butterfly_gs_fp(ts, U, V, tw, correct)
        ops = self.ops_sum_diff(ts, U, V)
        mm = self.ops_modmul_inplace(ts, V, tw)
        if correct:
            # U in [0, 2B) -> U - B if that is non-negative (B = 8q or 4q, at most 2^63): D = U - B has bit 63 set exactly
            # when U < B; U = D + (B & mask).  Two full-rate instructions and three plain 32-bit ones instead of four full-rate.
            D = ts.R
            M = ts.C.lo()                                     # free until the product's cross terms start
            ops += [("v_lshl_add_u64", D, U, 0, self.NQ8)]
            ops += mm[:1]
            ops += [("v_ashrrev_i32", M, 31, D.hi()),
                    ("v_and_b32", U.lo(), self.Q8.lo(), M)]
            ops += mm[1:2]
            ops += [("v_and_b32", U.hi(), self.Q8.hi(), M),
                    ("v_lshl_add_u64", U, U, 0, D)]
            ops += mm[2:]
        else:
            ops += mm
        return ops

    def ops_canon4(self, ts, X):
        """[0,4q) -> [0,q)"""
        D, M = ts.R, ts.T0
        # twice "X - c if non-negative" by the sign of the difference as a mask (X < 4q < 2^63): c = 2q, then c = q
        return [("v_lshl_add_u64", D, X, 0, self.NQ2),
                ("v_ashrrev_i32", M, 31, D.hi()),
                ("v_and_b32", X.lo(), self.Qm.lo(), M),
                ("v_and_b32", X.hi(), self.Qm.hi(), M),
                ("v_lshl_add_u64", X, X, 1, D),              # D + 2 * (q & mask)
                ("v_lshl_add_u64", D, X, 0, self.NQ),
                ("v_ashrrev_i32", M, 31, D.hi()),
                ("v_and_b32", X.lo(), self.Qm.lo(), M),
                ("v_and_b32", X.hi(), self.Qm.hi(), M),
                ("v_lshl_add_u64", X, X, 0, D)]

    def ops_last(self, ts, U, V, tw_n, tw_wn):
        """last stage fused with the scaling: canonical (U+V)*N^-1 and (U-V)*psi_inv[1]*N^-1"""
        if self.fp:
            return ([("v_add_f64", ts.T23, U, Neg(V)), ("v_add_f64", U, U, V)]
                    + self.ops_modmul_fp(ts, U, tw_n, dst=U) + self.ops_modmul_fp(ts, ts.T23, tw_wn, dst=V)
                    + self.ops_canon_fp(ts, U) + self.ops_canon_fp(ts, V))
        ops = self.ops_sum_diff(ts, U, V)
        ops += self.ops_modmul_inplace(ts, U, tw_n)
        ops += self.ops_modmul_inplace(ts, V, tw_wn)
        # U and V share the temp D and the carry register: sequential, the hazard tracker pads where the
        # other butterfly in flight does not
        return ops + self.ops_canon4(ts, U) + self.ops_canon4(ts, V)

    def corr(self, b, k0, first_pass=False):
        """does the butterfly over local bit b at position k0 subtract the bound from its sum?"""
        if self.fp:
            return (self.stage_base + b) in self.fp_reduce
        if self.mode == 0:
            return True
        if b == 0:
            return not first_pass           # operand types depend on the lane across a pass boundary
        return ((k0 >> (b - 1)) & 1) == 0   # sums of two X-type values only

    def gs_group(self, Y, R, twf, first_pass, hook=None, last=None):
        """R Gentleman-Sande stages over the local bits 0..R-1 of 2^R coefficients.
        twf(c, j): twiddle of heap position (H << c) + j, c = R-1-b."""
        n = 0
        for b in range(R):
            c = R - 1 - b
            if last is not None and b == R - 1:
                last()
                return
            blist = []
            for j in range(1 << c):
                tw = twf(c, j)
                for i in range(1 << b):
                    k0 = (j << (b + 1)) | i
                    blist.append((Y[k0], Y[k0 | (1 << b)], tw, self.corr(b, k0, first_pass)))
            self.butterflies(blist)
            for _ in range(1 << c):
                n += 1
                if hook:
                    hook(n)

    # ------------------------------------------------------------------ sections
    def prologue_tail(self):
        e = self.e
        if self.profile:
            # (s[0:1] becomes the LimbParams pointer below: park_stamp_slot() has put the kernel-argument pointer into the row-0 padding)
            self.stamp(0)
        if self.dual:
            self.mark = len(self.p.ins)
        if self.fp:
            e("s_waitcnt", "lgkmcnt(0)")
            e("s_cmp_eq_u32", self.FPL.sub(1), 0)
            e("s_cbranch_scc1", "INT_BODY")
            for ptr, d in ((self.TW, self.DTW), (self.TWF, self.DTWF)):
                e("s_add_u32", ptr.lo(), ptr.lo(), d.lo())
                e("s_addc_u32", ptr.hi(), ptr.hi(), d.hi())
            for i, dst in enumerate((self.QD, self.QINV, self.NINV, self.NINVQ)):
                e("s_mov_b64", dst, self.FPL.sub(2 * i, 2))
            e("s_mov_b32", self.MAGIC.lo(), 0)
            e("s_mov_b32", self.MAGIC.hi(), 0x43300000)
            e("v_mul_f64", self.BIAS, self.QINV, 0.5)
            return
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_mov_b64", self.Qm, s(68, 2))
        e("s_sub_u32", self.NQ.lo(), 0, self.Qm.lo())
        e("s_subb_u32", self.NQ.hi(), 0, self.Qm.hi())
        e("s_lshl_b64", self.Q8, self.Qm, 2 if self.mode == 0 else 3)
        e("s_sub_u32", self.NQ8.lo(), 0, self.Q8.lo())
        e("s_subb_u32", self.NQ8.hi(), 0, self.Q8.hi())
        e("s_lshl_b64", self.NQ2, self.NQ, 1)
        if not self.sub:
            # LimbParams pointer for the reload before the last stage
            e("s_lshl_b32", self.SC[3], self.SC[0], 6)
            e("s_add_u32", self.LP.lo(), s(52), self.SC[3])
            e("s_addc_u32", self.LP.hi(), s(53), 0)

    def sub_source(self, blk_bytes):
        """inverse: this block's half of the limb"""
        e = self.e
        e("s_add_u32", self.SRC.lo(), self.SRC.lo(), blk_bytes)
        e("s_addc_u32", self.SRC.hi(), self.SRC.hi(), 0)

    def copy_in(self, half):
        """wave w loads its 1024 contiguous coefficients (8 KiB) with 16-byte loads and lays them into its LDS block"""
        e = self.e
        a0, a1, a2 = self.A_[0], self.A_[1], self.A_[2]
        self.c("half %d: copy-in (wave-local)" % half)
        self.wave_lds_base()
        e("v_lshrrev_b32", a1, 3, self.LANE)
        e("v_lshlrev_b32", a1, 4, a1)
        e("v_lshl_add_u32", a0, self.LANE, 4, a1)             # slot(2l) * 8
        e("v_add_u32", a0, self.SC[6], a0)
        e("v_lshlrev_b32", a2, 4, self.LANE)                  # l * 16
        e("s_lshl_b32", self.SC[5], self.WAVE, 13)
        e("s_add_u32", self.TMP.lo(), self.SRC.lo(), self.SC[5])
        e("s_addc_u32", self.TMP.hi(), self.SRC.hi(), 0)
        n = 8
        early = self.pre1 and half == 1                       # the loads were issued by prefetch_half1()
        regs = [v((32 if early else 0) + 4 * i, 4) for i in range(n)]
        if not early:
            self.copy_in_loads(regs, a2)
        if half + 1 < self.HALVES:
            e("s_add_u32", self.SRC.lo(), self.SRC.lo(), self.M * 8)
            e("s_addc_u32", self.SRC.hi(), self.SRC.hi(), 0)
        # twiddles of the wave-local stages travel behind the data: bits 0..3 first, then the two bits-4..6 tasks
        fin = dict(self.final_requests(half))
        order = [("f", c, j) for c in (3, 2, 1, 0) for j in range(1 << c)]
        reqs = [(k, fin[k]) for k in order]
        for g, tag in ((0, "a"), (1, "b")):
            lane = dict(self.lane_twiddle_requests(half, g, tag))
            reqs += [((tag, c, j), lane[(tag, c, j)]) for c in (2, 1, 0) for j in range(1 << c)]
        # final_requests clobbers A_[2] (as l*16 -> t*16); the data loads above have been issued already
        self.stream_begin(reqs)
        younger = self.st_issued
        for i in range(n):
            e("s_waitcnt", "vmcnt(%d)" % (younger + n - 1 - i))
            e("ds_write_b128", a0, regs[i], offset=i * 1152)
        if early:
            # half 0's columns waited in v0..v31 (column_read) for these registers
            for k in range(self.SPH):
                for part in range(2):
                    e("v_mov_b32", self.X[k].sub(part), v(2 * k + part))

    def copy_in_loads(self, regs, a2):
        """8 x 16 bytes per lane of this wave's 8 KiB at TMP, a2 = lane * 16"""
        e = self.e
        for i in range(len(regs)):
            if i == 4:                                        # the immediate offset is 13-bit signed
                e("s_add_u32", self.TMP.lo(), self.TMP.lo(), 4096)
                e("s_addc_u32", self.TMP.hi(), self.TMP.hi(), 0)
            e("global_load_dwordx4", regs[i], a2, self.TMP, offset=(i % 4) * 1024, hint="nt")

    def prefetch_half1(self):
        """pre1: half 1's raw rows on their way into v32..v63 while half 0 is in its last LDS pass (no twiddle stream is live here:
        the vector-memory queue holds nothing younger than these loads until copy_in(1) starts its stream)"""
        e = self.e
        a2 = self.A_[2]
        self.c("prefetch of half 1's copy-in (SRC already points at half 1)")
        e("v_lshlrev_b32", a2, 4, self.LANE)
        e("s_lshl_b32", self.SC[5], self.WAVE, 13)
        e("s_add_u32", self.TMP.lo(), self.SRC.lo(), self.SC[5])
        e("s_addc_u32", self.TMP.hi(), self.SRC.hi(), 0)
        self.copy_in_loads([v(32 + 4 * i, 4) for i in range(8)], a2)

    def pass_low(self, half):
        """bits 0..3: lane l owns the 16 contiguous coefficients of block l of its wave"""
        e = self.e
        a0 = self.A_[0]
        self.c("stages over bits 0..3 (wave-local)")
        e("s_movk_i32", self.SC[4], 144)
        e("v_mul_lo_u32", a0, self.LANE, self.SC[4])
        e("v_add_u32", a0, self.SC[6], a0)
        Y = [v(2 * k, 2) for k in range(16)]
        for k in range(0, 16, 2):
            e("ds_read_b128", v(2 * k, 4), a0, offset=8 * k)
        e("s_waitcnt", "lgkmcnt(0)")
        self.stage_base = 0
        if self.fp:
            self.zip_emit([(lambda ts, x=y: self.ops_ingest_small(ts, x)) for y in Y])
        for b in range(4):
            c = 3 - b
            js = list(range(1 << c))
            for j0 in range(0, len(js), 2):
                group = js[j0:j0 + 2]
                blist = []
                for j in group:
                    tw = self.stream_wait(("f", c, j))
                    for i in range(1 << b):
                        k0 = (j << (b + 1)) | i
                        blist.append((Y[k0], Y[k0 | (1 << b)], tw, self.corr(b, k0, True)))
                self.butterflies(blist)
                for j in group:
                    self.stream_release(("f", c, j))
        for k in range(0, 16, 2):
            e("ds_write_b128", a0, v(2 * k, 4), offset=8 * k)

    def pass_lane(self, half):
        """bits 4..6: lane l owns rows (l >> 4) and (l >> 4) + 4, column l & 15"""
        e = self.e
        a0, a2 = self.A_[0], self.A_[2]
        self.c("stages over bits 4..6 (wave-local)")
        e("v_lshrrev_b32", a2, 4, self.LANE)
        e("s_movk_i32", self.SC[4], 1152)
        e("v_mul_lo_u32", a0, a2, self.SC[4])
        e("v_and_b32", a2, 15, self.LANE)
        e("v_lshl_add_u32", a0, a2, 3, a0)
        e("v_add_u32", a0, self.SC[6], a0)
        YA = [v(2 * k, 2) for k in range(8)]
        YB = [v(16 + 2 * k, 2) for k in range(8)]
        for k in range(8):
            e("ds_read_b64", YA[k], a0, offset=k * 144)
        for k in range(8):
            e("ds_read_b64", YB[k], a0, offset=k * 144 + 4608)
        self.stage_base = 4
        for tag, Y, off in (("a", YA, 0), ("b", YB, 4608)):
            first = [True]

            def twf(c, j, tag=tag):
                tw = self.stream_wait((tag, c, j), "lgkmcnt(8)" if first[0] else "")
                first[0] = False
                return tw

            order = [(tag, c, j) for c in (2, 1, 0) for j in range(1 << c)]
            self.gs_group(Y, 3, twf, False, hook=lambda n, order=order: self.stream_release(order[n - 1]))
            for k in range(8):
                e("ds_write_b64", a0, Y[k], offset=k * 144 + off)

    def pass_uniform(self, half):
        """bits 7..9: lane l owns the columns l and l + 64 of its wave's block; twiddles in SGPRs"""
        e = self.e
        a0, a1 = self.A_[0], self.A_[1]
        if self.pre1 and half == 0:
            self.prefetch_half1()
        self.c("stages over bits 7..9 (wave-local)")
        e("v_lshrrev_b32", a1, 4, self.LANE)
        e("v_lshlrev_b32", a1, 4, a1)
        e("v_lshl_add_u32", a0, self.LANE, 3, a1)             # slot(l) * 8
        e("v_add_u32", a0, self.SC[6], a0)
        YA = [v(2 * k, 2) for k in range(8)]
        YB = [v(16 + 2 * k, 2) for k in range(8)]
        for k in range(8):
            e("ds_read_b64", YA[k], a0, offset=k * 1152)
        for k in range(8):
            e("ds_read_b64", YB[k], a0, offset=k * 1152 + 576)
        twf = lambda c, j: tuple(s(self.PB + (0, 8, 16)[c] + 4 * j + i) for i in range(4))
        e("s_waitcnt", "lgkmcnt(8)")
        self.stage_base = 7
        self.gs_group(YA, 3, twf, False)
        for k in range(8):
            e("ds_write_b64", a0, YA[k], offset=k * 1152)
        e("s_waitcnt", "lgkmcnt(8)")
        self.gs_group(YB, 3, twf, False)
        for k in range(8):
            e("ds_write_b64", a0, YB[k], offset=k * 1152 + 576)

    def column_read(self, half):
        e = self.e
        a0, a1, a2 = self.A_[0], self.A_[1], self.A_[2]
        self.c("column exchange: thread t takes the coefficients {kk*1024 + t} of this half")
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_barrier")
        e("v_lshrrev_b32", a2, 4, self.TID)
        e("v_lshlrev_b32", a2, 4, a2)
        e("v_lshl_add_u32", a0, self.TID, 3, a2)              # slot(t) * 8
        e("v_add_u32", a1, 8 * 9216 if self.C == 1 else 4608, a0)
        for c in range(self.C):
            for kk in range(self.SPH):
                base, off = self.lds_col(c, kk, a0, a1)
                dst = self.X[c * self.RA + self.SPH * half + kk]
                if self.pre1 and half == 0:
                    dst = v(2 * kk, 2)                        # v32..v63 carry half 1's raw rows until copy_in(1) has stored them
                e("ds_read_b64", dst, base, offset=off)
        if half + 1 < self.HALVES:
            e("s_waitcnt", "lgkmcnt(0)")
            e("s_barrier")

    # ---- pass A: SGPR twiddles in chunks of 8 (32 dwords), two buffers
    def chunk_plan(self):
        plan = []
        for c in range(self.A - 1, 2, -1):
            for j0 in range(0, 1 << c, 8):
                plan.append(("tw", c, j0))
        plan.append(("low",))           # heap entries 0..7: c = 2, 1 and the fused constant at index 0
        if not self.sub:
            plan.append(("lp",))        # N^-1 for the fused last stage (a sub-block leaves that to ntt_top_kernel)
        return plan

    def chunk_load(self, n):
        e = self.e
        item = self.plan[n]
        buf = (self.QB, self.PB)[n % 2]
        if item[0] == "tw":
            _, c, j0 = item
            off = ((1 << c) + j0) * 16
            table = self.stage_table(c)
            e("s_load_dwordx16", s(buf, 16), table, off)
            e("s_load_dwordx16", s(buf + 16, 16), table, off + 64)
        elif item[0] == "low" and self.sub:
            # under heap root 2 + blk: stage c = 2 -> 4 entries at buf, c = 1 -> 2 entries at buf+16, c = 0 -> buf+24
            e("s_load_dwordx16", s(buf, 16), self.stage_table(2), 4 * 16)
            e("s_load_dwordx8", s(buf + 16, 8), self.stage_table(1), 2 * 16)
            e("s_load_dwordx4", s(buf + 24, 4), self.stage_table(0), 1 * 16)
        elif item[0] == "low":
            e("s_load_dwordx16", s(buf, 16), self.TW, 0)
            e("s_load_dwordx16", s(buf + 16, 16), self.TW, 64)
        else:
            e("s_load_dwordx16", s(buf, 16), self.LP, 0)
        return buf

    def pass_a(self):
        e = self.e
        A, RA = self.A, self.RA
        self.c("top %d stages in registers, wave-uniform twiddles in SGPRs" % A)
        X = self.X
        plan = self.plan
        nchunks = len(plan)
        self.stage_base = 10
        # chunks 0 and 1 were requested during the last half's LDS phase
        for n in range(nchunks):
            item = plan[n]
            buf = (self.QB, self.PB)[n % 2]
            e("s_waitcnt", "lgkmcnt(0)")
            if item[0] == "tw":
                _, c, j0 = item
                b = A - 1 - c
                blist = []
                for j in range(j0, j0 + 8):
                    tw = tuple(s(buf + 4 * (j - j0) + i) for i in range(4))
                    for i in range(1 << b):
                        k0 = (j << (b + 1)) | i
                        for col in range(self.C):
                            blist.append((X[col * RA + k0], X[col * RA + (k0 | (1 << b))], tw, self.corr(b, k0)))
                self.butterflies(blist)
                if n + 2 < nchunks:
                    self.chunk_load(n + 2)
            elif item[0] == "low":
                for c in ((2, 1, 0) if self.sub else (2, 1)):
                    if c > A - 1:
                        continue
                    b = A - 1 - c
                    blist = []
                    for j in range(1 << c):
                        if self.sub:
                            tw = tuple(s(buf + (0, 16, 24)[2 - c] + 4 * j + i) for i in range(4))
                        else:
                            tw = tuple(s(buf + 4 * ((1 << c) + j) + i) for i in range(4))
                        for i in range(1 << b):
                            k0 = (j << (b + 1)) | i
                            for col in range(self.C):
                                blist.append((X[col * RA + k0], X[col * RA + (k0 | (1 << b))], tw, self.corr(b, k0)))
                    self.butterflies(blist)
                self.low_buf = buf
            else:
                # LimbParams: n_inv at dwords 10..11, its Shoup companion at 12..13
                tw_n = (s(buf + 10), s(buf + 11), s(buf + 12), s(buf + 13))
                if self.fp:
                    tw_n = (self.NINV.lo(), self.NINV.hi(), self.NINVQ.lo(), self.NINVQ.hi())
                tw_wn = tuple(s(self.low_buf + i) for i in range(4))
                b = A - 1
                items = []
                for i in range(1 << b):
                    for col in range(self.C):
                        items.append(lambda ts, U=X[col * RA + i], V=X[col * RA + (i | (1 << b))]: self.ops_last(ts, U, V, tw_n, tw_wn))
                self.zip_emit(items)

    def store_columns(self, hint="nt"):
        e = self.e
        self.c("coalesced store of the columns {k*S + t + c*T}")
        e("v_lshlrev_b32", self.GOFF, 3, self.TID)
        if self.C > 1:
            e("v_add_u32", self.A_[2], 4096, self.GOFF)
        if self.fp and self.sub:
            # ntt_top_kernel continues on integers: canonical residues
            self.zip_emit([(lambda ts, x=x: self.ops_canon_fp(ts, x)) for x in self.X])
        for k in range(self.RA):
            for col in range(self.C):
                off, imm = self.col_addr(col)
                e("global_store_dwordx2", off, self.X[col * self.RA + k], self.DST, offset=imm, hint=hint)
            e("s_add_u32", self.DST.lo(), self.DST.lo(), self.S * 8)
            e("s_addc_u32", self.DST.hi(), self.DST.hi(), 0)

    def fused_last(self):
        e, X, sc = self.e, self.X, self.SC
        ts0 = self.ts[0]
        assert self.C == 1 and self.RA == 32
        tag = "fp" if self.fp else "int"
        # own half (lazy / canonical integers) with device-scope stores: written through this XCD's L2, complete when vmcnt says so
        # (a buffer_wbl2 per wave instead would sweep the whole L2 every time: measured 4x slower for the launch); DST ends at own base + 8N
        self.store_columns(hint="sc1")
        self.c("release: the rows are visible to the device, then the pair flag")
        e("s_waitcnt", "vmcnt(0)")
        K = 36
        self.unpark(s(K, 4))                                  # s[K:K+1] flag address, s[K+2:K+3] LimbParams address
        # one increment per wave, by lane 0 alone (64 lanes on one address would queue 1024 operations per workgroup on the one
        # cache line its sixteen flags share)
        e("v_mov_b32", ts0.T0, 0)
        e("v_mov_b32", ts0.T2, 1)
        e("s_mov_b64", EXEC, 1)
        e("global_atomic_add", ts0.Q.lo(), ts0.T0, ts0.T2, s(K, 2), hint="sc0")
        e("s_mov_b64", EXEC, -1)
        W = tuple(s(K + 4 + i) for i in range(4))              # psi_inv[1] * N^-1: index 0 of the inverse table (FP table in the FP body)
        e("s_load_dwordx4", s(K + 4, 4), self.TW, 0)
        if self.fp:
            tw_n = (self.NINV.lo(), self.NINV.hi(), self.NINVQ.lo(), self.NINVQ.hi())
        else:
            tw_n = tuple(s(K + 8 + i) for i in range(4))
            e("s_load_dwordx4", s(K + 8, 4), s(K + 2, 2), 40)  # LimbParams: N^-1 and its Shoup companion (dwords 10..13)
        e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
        e("v_readfirstlane_b32", sc[0], ts0.Q.lo())
        e("s_nop", 4)
        e("s_cmp_ge_u32", sc[0], 1)
        e("s_cbranch_scc1", "L_second_" + tag)
        e("s_endpgm")                                          # first of the pair: the partner finishes the limb
        self.p.label("L_second_" + tag)
        # acquire: the partner's rows are read with device-scope loads below (no buffer_inv: it would drop the XCD's whole L2,
        # twiddle tables included, once per wave); the flag goes back to 0
        e("s_mov_b64", EXEC, 1)
        e("global_store_dword", ts0.T0, ts0.T0, s(K, 2))
        e("s_mov_b64", EXEC, -1)
        # bases: low half = DST - BLK1 * 8N (DST sits 8N past this block's rows), the partner's rows = the other half
        LOB, HIB, PART = s(K + 16, 2), s(K + 18, 2), s(K + 20, 2)
        e("s_lshl_b32", sc[1], self.BLK1, self.logn + 3)
        e("s_sub_u32", LOB.lo(), self.DST.lo(), sc[1])
        e("s_subb_u32", LOB.hi(), self.DST.hi(), 0)
        e("s_add_u32", HIB.lo(), LOB.lo(), self.N * 8)
        e("s_addc_u32", HIB.hi(), LOB.hi(), 0)
        # (mine - partner) * w' with w' = w for block 0 and -w for block 1, so that the products are (A - B) * w either way
        NW = tuple(s(K + 12 + i) for i in range(4))
        if self.fp:
            e("s_mov_b32", NW[0], W[0])
            e("s_xor_b32", NW[1], W[1], 0x80000000)
            e("s_mov_b32", NW[2], W[2])
            e("s_xor_b32", NW[3], W[3], 0x80000000)
        else:
            e("s_sub_u32", NW[0], self.Qm.lo(), W[0])
            e("s_subb_u32", NW[1], self.Qm.hi(), W[1])
            e("s_not_b32", NW[2], W[2])                       # floor((q - w) * 2^64 / q) = 2^64 - 1 - floor(w * 2^64 / q), 0 < w < q
            e("s_not_b32", NW[3], W[3])
        e("s_cmp_eq_u32", self.BLK1, 1)
        for i in range(4):
            e("s_cselect_b32", W[i], W[i], NW[i])
        e("s_cselect_b32", PART.lo(), HIB.lo(), LOB.lo())
        e("s_cselect_b32", PART.hi(), HIB.hi(), LOB.hi())
        P = [v(self.tw_base + 2 * i, 2) for i in range(16)]
        for chunk in range(2):
            for i in range(16):
                e("global_load_dwordx2", P[i], self.GOFF, PART, hint="sc1")
                e("s_add_u32", PART.lo(), PART.lo(), self.S * 8)
                e("s_addc_u32", PART.hi(), PART.hi(), 0)
            for i in range(0, 16, 2):
                # loads return in order; the stores of the previous chunk still in flight only make the wait longer
                e("s_waitcnt", "vmcnt(%d)" % (14 - i))
                items = []
                for d in range(2):
                    U, V = X[16 * chunk + i + d], P[i + d]
                    if self.fp:
                        items.append(lambda ts, U=U, V=V: self.ops_ingest_small(ts, U) + self.ops_ingest_small(ts, V) + self.ops_last(ts, U, V, tw_n, W))
                    else:
                        items.append(lambda ts, U=U, V=V: self.ops_last(ts, U, V, tw_n, W))
                self.zip_emit(items)
            for i in range(16):
                e("global_store_dwordx2", self.GOFF, X[16 * chunk + i], LOB, hint="nt")
                e("global_store_dwordx2", self.GOFF, P[i], HIB, hint="nt")
                for ptr in (LOB, HIB):
                    e("s_add_u32", ptr.lo(), ptr.lo(), self.S * 8)
                    e("s_addc_u32", ptr.hi(), ptr.hi(), 0)

    # stamps of the timeline build, in order (tools/timeline.py names the intervals between them)
    STAMP_NAMES = ["start"] + ["%s (half %d)" % (n, h) for h in range(2) for n in (
        "copy-in: loads returned, LDS written", "stages over bits 0..3", "stages over bits 4..6", "stages over bits 7..9",
        "column exchange")] + ["top stages + last stage", "stores issued"]

    def build(self):
        self.plan = self.chunk_plan()
        self.prologue()
        for half in range(self.HALVES):
            last = half + 1 == self.HALVES
            self.uniform_twiddle_loads(half)
            if last:
                self.chunk_load(0)
            self.copy_in(half)
            self.stamp(1 + 5 * half)
            self.pass_low(half)
            self.stamp(2 + 5 * half)
            self.pass_lane(half)
            self.stamp(3 + 5 * half)
            self.pass_uniform(half)
            self.stamp(4 + 5 * half)
            if last:
                self.chunk_load(1)
            self.column_read(half)
            self.stamp(5 + 5 * half)
        self.pass_a()
        self.stamp(1 + 5 * self.HALVES)
        if self.fuse_last:
            self.fused_last()
        else:
            self.store_columns()
        self.stamp(2 + 5 * self.HALVES)
        if self.profile:
            self.flush_stamps(3 + 5 * self.HALVES)
        self.e("s_endpgm")
        return self.p


if __name__ == "__main__":
    logn = int(sys.argv[1])
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    threads = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    # mode 3: FP64 body for the limbs below 2^46, the integer body of mode 1 for the others
    def make(logn_, threads_, **kw):
        if mode == 3:
            return Dual(lambda fp: GenInv(logn_, 1, threads_, fp=fp, dual=True, **kw))
        return GenInv(logn_, mode, threads_, **kw)

    if logn == 16:      # "s": lazy sub-blocks, ntt_top_kernel follows; "f": the last stage fused (pair flags in NttLaunch::epi_x)
        fused = len(sys.argv) > 5 and sys.argv[5] == "fused"
        open(sys.argv[2], "w").write(kernel_text_for(make(15, 1024, sub=True, fuse_last=fused), "lr_ntt_inv16%s_m%d" % ("f" if fused else "s", mode)))
        sys.exit(0)
    if len(sys.argv) > 5 and sys.argv[5] == "halves":        # N = 2^15 as two 2^14 sub-blocks with lazy outputs, ntt_top_kernel follows: small launches
        assert logn == 15
        open(sys.argv[2], "w").write(kernel_text_for(make(14, 1024, sub=True), "lr_ntt_inv15h_m%d" % mode))
        sys.exit(0)
    if len(sys.argv) > 5 and sys.argv[5] == "timeline":      # diagnostics build with per-phase clock stamps (Options::timeline)
        open(sys.argv[2], "w").write(kernel_text_for(make(logn, threads, profile=True), "lr_ntt_inv%d_m%dt" % (logn, mode)))
        sys.exit(0)
    name = "lr_ntt_inv%d%s_m%d" % (logn, "x" if threads < 1024 else "", mode)
    open(sys.argv[2], "w").write(kernel_text_for(make(logn, threads), name))

"""Generator of the hand-scheduled forward NTT kernels for gfx950 (N = 2^14, 2^15; 1024 threads).

Same algorithm, data layout and twiddle tables as the C++ kernel `ntt_fwd_kernel<LOGN, 1>` in
lr_ntt.hip (pass A in registers -> LDS passes 3+3+4 -> coalesced copy-out, lazy Shoup butterflies
with correction every second stage); what changes is the instruction stream: registers are
assigned by hand, the zero halves of the multiply-accumulate addends live in dedicated registers,
conditional subtractions use the high-word compare, and nothing spills.  15 VALU instructions per
butterfly without correction, 19 with (the compiler's version: 21-26).

Valid for contexts whose moduli all satisfy 2^57 <= q <= 2^60 (lazy mode 1 of lr_ntt.hip); the
host falls back to the C++ kernels otherwise.

    python gen_ntt.py 15 out.s        # assembly text
    python gen_ntt.py 15 --selftest   # emulate one workgroup with numpy and compare with a reference NTT
"""
import sys

from isa import VCC, Program, s, v

T = 1024
LOGT = 10


class Gen:
    def __init__(self, logn):
        assert logn in (14, 15)
        self.logn = logn
        self.N = 1 << logn
        self.A = logn - 10            # bits consumed by pass A
        self.RA = 1 << self.A         # coefficients per thread
        self.S = self.N >> self.A     # = T
        self.HALVES = self.RA // 16
        self.M = self.N // self.HALVES
        self.RH = 16
        assert self.S == T and self.M == 16384
        self.p = Program()
        self.pos = 0                  # issue position for the carry hazard tracker
        self.carry_pos = {}
        # ---- VGPR map
        self.X = [v(2 * k, 2) for k in range(self.RA)]
        base = 2 * self.RA if self.RA >= 32 else 64
        self.T0, self.Z1 = v(base + 0), v(base + 1)
        self.T2, self.Z3 = v(base + 2), v(base + 3)
        self.T01, self.T23 = v(base + 0, 2), v(base + 2, 2)
        self.Q = v(base + 4, 2)
        self.R = v(base + 6, 2)
        self.C = v(base + 8, 2)
        self.TID = v(base + 10)
        self.GOFF = v(base + 11)
        self.A_ = [v(base + 12 + i) for i in range(4)]
        self.tw_base = base + 16
        self.n_tw_slots = 12
        self.vgpr_count = self.tw_base + 4 * self.n_tw_slots
        assert self.vgpr_count <= 128
        # ---- SGPR map
        self.KARG = s(0, 2)
        self.WGX, self.WGY = s(2), s(3)
        self.SRC, self.DST = s(4, 2), s(6, 2)
        self.TW, self.TWF, self.TWFR = s(8, 2), s(10, 2), s(12, 2)
        self.Qm, self.NQ, self.Q4, self.NQ8 = s(14, 2), s(16, 2), s(18, 2), s(20, 2)
        self.REDM, self.REDG, self.WAVE = s(22), s(23), s(24)
        self.SC = [s(25 + i) for i in range(7)]      # s25..s31 scratch
        self.JUNK = s(32, 2)
        self.TMP = s(34, 2)
        self.PB, self.QB = 36, 68                     # two 32-dword twiddle buffers

    # ------------------------------------------------------------------ emission helpers
    def e(self, op, *args, **mods):
        """emit with the VALU-writes-carry -> VALU-reads-carry hazard (2 wait states) handled"""
        reads = []
        if op in ("v_subb_co_u32",):
            reads.append(args[4])
        if op == "v_cndmask_b32":
            reads.append(args[3])
        for r in reads:
            key = repr(r)
            if key in self.carry_pos:
                gap = self.pos - self.carry_pos[key] - 1
                if gap < 2:
                    self.p.emit("s_nop", 1 - gap)
                    self.pos += 2 - gap
        self.p.emit(op, *args, **mods)
        self.pos += 1
        writes = []
        if op in ("v_sub_co_u32", "v_subb_co_u32"):
            writes.append(args[1])
        if op in ("v_cmp_lt_u32", "v_cmp_gt_i32"):
            writes.append(args[0])
        for w in writes:
            self.carry_pos[repr(w)] = self.pos - 1

    def c(self, text):
        self.p.comment(text)

    def tw_slot(self, i):
        return v(self.tw_base + 4 * i, 4)

    # ------------------------------------------------------------------ arithmetic macros
    def modmul(self, V, tw):
        """R <- V * w - qhat * q (lazy, [0,4q)); tw = (w0, w1, s0, s1) registers (SGPR or VGPR)."""
        w0, w1, s0, s1 = tw
        e = self.e
        e("v_mul_hi_u32", self.T0, V.hi(), s0)
        e("v_mul_hi_u32", self.T2, V.lo(), s1)
        e("v_mad_u64_u32", self.Q, self.JUNK, V.hi(), s1, self.T01)
        e("v_mad_u64_u32", self.R, self.JUNK, V.lo(), w0, 0)
        e("v_mad_u64_u32", self.C, self.JUNK, V.lo(), w1, 0)
        e("v_lshl_add_u64", self.Q, self.Q, 0, self.T23)
        e("v_mad_u64_u32", self.C, self.JUNK, V.hi(), w0, self.C)
        e("v_mad_u64_u32", self.R, self.JUNK, self.Q.lo(), self.NQ.lo(), self.R)
        e("v_mad_u64_u32", self.C, self.JUNK, self.Q.lo(), self.NQ.hi(), self.C)
        e("v_mad_u64_u32", self.C, self.JUNK, self.Q.hi(), self.NQ.lo(), self.C)
        e("v_add_u32", self.R.hi(), self.R.hi(), self.C.lo())

    def butterfly(self, U, V, tw, correct):
        """(U, V) <- (U + V*w, U - V*w + 4q); optional U <- U - 8q if U >= 8q first."""
        e = self.e
        if correct:
            D = self.C
            e("v_lshl_add_u64", D, U, 0, self.NQ8)
            e("v_cmp_lt_u32", VCC, D.hi(), U.hi())
            e("v_mul_hi_u32", self.T0, V.hi(), tw[2])      # fills the two wait states
            e("v_mul_hi_u32", self.T2, V.lo(), tw[3])
            e("v_cndmask_b32", U.lo(), U.lo(), D.lo(), VCC)
            e("v_cndmask_b32", U.hi(), U.hi(), D.hi(), VCC)
            w0, w1, s0, s1 = tw
            e("v_mad_u64_u32", self.Q, self.JUNK, V.hi(), s1, self.T01)
            e("v_mad_u64_u32", self.R, self.JUNK, V.lo(), w0, 0)
            e("v_mad_u64_u32", self.C, self.JUNK, V.lo(), w1, 0)
            e("v_lshl_add_u64", self.Q, self.Q, 0, self.T23)
            e("v_mad_u64_u32", self.C, self.JUNK, V.hi(), w0, self.C)
            e("v_mad_u64_u32", self.R, self.JUNK, self.Q.lo(), self.NQ.lo(), self.R)
            e("v_mad_u64_u32", self.C, self.JUNK, self.Q.lo(), self.NQ.hi(), self.C)
            e("v_mad_u64_u32", self.C, self.JUNK, self.Q.hi(), self.NQ.lo(), self.C)
            e("v_add_u32", self.R.hi(), self.R.hi(), self.C.lo())
        else:
            self.modmul(V, tw)
        e("v_lshl_add_u64", V, U, 0, self.Q4)            # Y = U + 4q ...
        e("v_lshl_add_u64", U, U, 0, self.R)             # X = U + r
        e("v_sub_co_u32", V.lo(), VCC, V.lo(), self.R.lo())
        e("v_subb_co_u32", V.hi(), VCC, V.hi(), self.R.hi(), VCC)   # ... - r

    def reduce_2q(self, X):
        """X <- X - floor~(X/q) * q in [0, 2q) for any 64-bit X (quotient under-estimated by <= 1)."""
        e = self.e
        k = self.T0
        e("v_mul_hi_u32", k, X.hi(), self.REDM)
        e("v_lshrrev_b32", k, self.REDG, k)
        e("v_mad_u64_u32", self.R, self.JUNK, k, self.Qm.lo(), 0)
        e("v_mul_lo_u32", self.T2, k, self.Qm.hi())
        e("v_add_u32", self.R.hi(), self.R.hi(), self.T2)
        e("v_sub_co_u32", X.lo(), VCC, X.lo(), self.R.lo())
        e("v_subb_co_u32", X.hi(), VCC, X.hi(), self.R.hi(), VCC)

    def canon(self, X):
        """X in [0, 16q) -> canonical [0, q)."""
        e = self.e
        self.reduce_2q(X)
        D = self.C
        e("v_lshl_add_u64", D, X, 0, self.NQ)
        e("v_cmp_gt_i32", VCC, 0, D.hi())
        e("v_cndmask_b32", X.lo(), D.lo(), X.lo(), VCC)
        e("v_cndmask_b32", X.hi(), D.hi(), X.hi(), VCC)

    def correct_flag(self, stage):
        return stage >= 2 and stage % 2 == 0

    # ------------------------------------------------------------------ kernel sections
    def prologue(self):
        e, S_ = self.e, self
        logn, N = self.logn, self.N
        self.c("kernel arguments (NttLaunch, 88 bytes)")
        e("s_load_dwordx8", s(36, 8), self.KARG, 0)
        e("s_load_dwordx8", s(44, 8), self.KARG, 32)
        e("s_load_dwordx4", s(52, 4), self.KARG, 64)
        e("s_load_dwordx2", s(56, 2), self.KARG, 80)
        e("v_mov_b32", self.TID, v(0))
        e("v_mov_b32", self.Z1, 0)
        e("v_mov_b32", self.Z3, 0)
        e("v_lshlrev_b32", self.GOFF, 3, self.TID)
        e("v_readfirstlane_b32", self.WAVE, self.TID)
        e("s_nop", 4)
        e("s_lshr_b32", self.WAVE, self.WAVE, 6)
        e("s_waitcnt", "lgkmcnt(0)")
        sc = self.SC
        e("s_mul_i32", sc[0], self.WGX, s(49))
        e("s_add_u32", sc[0], sc[0], s(48))          # modulus index
        e("s_mul_i32", sc[1], self.WGX, s(45))
        e("s_add_u32", sc[1], sc[1], s(44))          # input row
        e("s_mul_i32", sc[2], self.WGX, s(47))
        e("s_add_u32", sc[2], sc[2], s(46))          # output row
        for (row, stride_lo, stride_hi, base_lo, base_hi, dst) in ((sc[1], s(40), s(41), s(36), s(37), self.SRC),
                                                                     (sc[2], s(42), s(43), s(38), s(39), self.DST)):
            e("s_mul_i32", self.TMP.lo(), self.WGY, stride_lo)
            e("s_mul_hi_u32", self.TMP.hi(), self.WGY, stride_lo)
            e("s_mul_i32", sc[3], self.WGY, stride_hi)
            e("s_add_u32", self.TMP.hi(), self.TMP.hi(), sc[3])
            e("s_lshl_b32", sc[3], row, logn)
            e("s_add_u32", self.TMP.lo(), self.TMP.lo(), sc[3])
            e("s_addc_u32", self.TMP.hi(), self.TMP.hi(), 0)
            e("s_lshl_b64", self.TMP, self.TMP, 3)
            e("s_add_u32", dst.lo(), base_lo, self.TMP.lo())
            e("s_addc_u32", dst.hi(), base_hi, self.TMP.hi())
        # LimbParams (64 bytes) of this modulus
        e("s_lshl_b32", sc[3], sc[0], 6)
        e("s_add_u32", self.TMP.lo(), s(52), sc[3])
        e("s_addc_u32", self.TMP.hi(), s(53), 0)
        e("s_load_dwordx16", s(68, 16), self.TMP, 0)
        # twiddle table bases
        e("s_mul_i32", sc[3], sc[0], N * 16)
        e("s_add_u32", self.TW.lo(), s(54), sc[3])
        e("s_addc_u32", self.TW.hi(), s(55), 0)
        e("s_mul_i32", sc[3], sc[0], 15 * N)
        e("s_add_u32", self.TWF.lo(), s(56), sc[3])
        e("s_addc_u32", self.TWF.hi(), s(57), 0)
        self.c("coalesced load of the column {k*S + t}")
        for k in range(self.RA):
            e("global_load_dwordx2", self.X[k], self.GOFF, self.SRC)
            e("s_add_u32", self.SRC.lo(), self.SRC.lo(), self.S * 8)
            e("s_addc_u32", self.SRC.hi(), self.SRC.hi(), 0)
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_mov_b64", self.Qm, s(68, 2))
        e("s_mov_b32", self.REDM, s(82))
        e("s_mov_b32", self.REDG, s(83))
        e("s_sub_u32", self.NQ.lo(), 0, self.Qm.lo())
        e("s_subb_u32", self.NQ.hi(), 0, self.Qm.hi())
        e("s_lshl_b64", self.Q4, self.Qm, 2)
        e("s_lshl_b64", self.TMP, self.Qm, 3)
        e("s_sub_u32", self.NQ8.lo(), 0, self.TMP.lo())
        e("s_subb_u32", self.NQ8.hi(), 0, self.TMP.hi())

    def pass_a(self):
        e = self.e
        A, RA = self.A, self.RA
        self.c("pass A: top %d stages in registers, wave-uniform twiddles in SGPRs" % A)

        def load_stage(c, buf, half=None):
            ntw = 1 << c
            first, count = (1 << c), ntw
            if half is not None:
                count = ntw // 2
                first += half * count
            off, dw, dst = first * 16, count * 4, buf
            while dw > 0:
                n = 16 if dw >= 16 else dw
                e({4: "s_load_dwordx4", 8: "s_load_dwordx8", 16: "s_load_dwordx16"}[n], s(dst, n), self.TW, off)
                off += n * 4
                dst += n
                dw -= n

        bufs = [self.PB, self.QB]
        load_stage(0, bufs[0])
        if A > 1:
            load_stage(1, bufs[1])
        for c in range(A):
            b = A - 1 - c
            split = (1 << c) * 4 > 32                 # stage needs both buffers (16 twiddles)
            e("s_waitcnt", "lgkmcnt(0)")
            # prefetch the next stage into the buffer the previous stage just released
            if c >= 1 and c + 1 < A:
                nsplit = (1 << (c + 1)) * 4 > 32
                load_stage(c + 1, bufs[(c + 1) % 2], 0 if nsplit else None)
            cur = bufs[c % 2]
            for j in range(1 << c):
                if split and j == (1 << c) // 2:
                    # second half of the twiddles goes where the previous stage's lived
                    load_stage(c, bufs[(c + 1) % 2], 1)
                    e("s_waitcnt", "lgkmcnt(0)")
                    cur = bufs[(c + 1) % 2] - 4 * j
                tw = tuple(s(cur + 4 * j + i) for i in range(4))
                for i in range(1 << b):
                    k0 = (j << (b + 1)) | i
                    k1 = k0 | (1 << b)
                    if c == 0:
                        e("s_waitcnt", "vmcnt(%d)" % (RA - 1 - k1))
                        self.reduce_2q(self.X[k0])       # first-stage U operands: any 64-bit value accepted
                    self.butterfly(self.X[k0], self.X[k1], tw, self.correct_flag(c))

    def lds_write_columns(self, half):
        e = self.e
        a0, a1, a2 = self.A_[0], self.A_[1], self.A_[2]
        self.c("half %d -> LDS image (16 B of padding per 16 coefficients)" % half)
        e("v_lshrrev_b32", a2, 4, self.TID)
        e("v_lshlrev_b32", a2, 4, a2)
        e("v_add_u32", a0, self.GOFF, a2)                    # slot(t) * 8
        e("v_add_u32", a1, 8 * 9216, a0)
        for kk in range(self.RH):
            base, off = (a0, kk * 9216) if kk < 8 else (a1, (kk - 8) * 9216)
            e("ds_write_b64", base, self.X[half * self.RH + kk], offset=off)
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_barrier")

    def lds_pass_uniform(self, half):
        """bits 9..7 (R = 3, PLO = 7): twiddles are wave-uniform."""
        e = self.e
        a0, a1, a2 = self.A_[0], self.A_[1], self.A_[2]
        sc = self.SC
        stage0 = self.logn - 10
        self.c("LDS pass over bits 9..7")
        e("v_and_b32", a2, 127, self.TID)
        e("v_lshrrev_b32", a1, 4, a2)
        e("v_lshlrev_b32", a1, 4, a1)
        e("v_lshl_add_u32", a0, a2, 3, a1)
        e("s_lshr_b32", sc[0], self.WAVE, 1)                 # u_hi of task 0
        e("s_mul_i32", sc[1], sc[0], 9216)
        e("v_add_u32", a0, sc[1], a0)
        e("v_add_u32", a1, 8 * 9216, a0)
        # twiddles: H = 2^(logn-10) + 16*half + u_hi (+8 for the second task)
        for g, buf in ((0, self.PB), (1, self.QB)):
            e("s_add_u32", sc[2], sc[0], (1 << (self.logn - 10)) + 16 * half + 8 * g)
            e("s_lshl_b32", sc[3], sc[2], 4)
            e("s_load_dwordx4", s(buf, 4), self.TW, sc[3])
            e("s_lshl_b32", sc[3], sc[2], 5)
            e("s_load_dwordx8", s(buf + 8, 8), self.TW, sc[3])
            e("s_lshl_b32", sc[3], sc[2], 6)
            e("s_load_dwordx16", s(buf + 16, 16), self.TW, sc[3])
        Y = [v(2 * k, 2) for k in range(8)]
        for g, (addr, buf) in enumerate(((a0, self.PB), (a1, self.QB))):
            for k in range(8):
                e("ds_read_b64", Y[k], addr, offset=k * 1152)
            e("s_waitcnt", "lgkmcnt(0)")
            self.radix8(Y, lambda c, j: tuple(s(buf + (0, 8, 16)[c] + 4 * j + i) for i in range(4)), stage0)
            for k in range(8):
                e("ds_write_b64", addr, Y[k], offset=k * 1152)
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_barrier")

    def radix8(self, Y, twf, stage0):
        for c in range(3):
            b = 2 - c
            for j in range(1 << c):
                tw = twf(c, j)
                for i in range(1 << b):
                    k0 = (j << (b + 1)) | i
                    self.butterfly(Y[k0], Y[k0 | (1 << b)], tw, self.correct_flag(stage0 + c))

    def lds_pass_lane(self, half):
        """bits 6..4 (R = 3, PLO = 4): per-lane twiddles from the heap-ordered table."""
        e = self.e
        a0, a1, a2, a3 = self.A_
        stage0 = self.logn - 7
        self.c("LDS pass over bits 6..4")
        e("v_lshrrev_b32", a2, 4, self.TID)                   # u_hi of task 0
        e("s_movk_i32", self.SC[4], 1152)                    # VOP3 takes no literal on gfx9
        e("v_mul_lo_u32", a0, a2, self.SC[4])
        e("v_and_b32", a3, 15, self.TID)
        e("v_lshl_add_u32", a0, a3, 3, a0)
        e("v_add_u32", a1, 8 * 9216, a0)
        Y = [v(2 * k, 2) for k in range(8)]
        for g, addr in enumerate((a0, a1)):
            # H = 2^(logn-7) + 128*half + u_hi + 64*g ; byte offsets H*16, H*32 + 16j, H*64 + 16j
            e("v_add_u32", a3, (1 << (self.logn - 7)) + 128 * half + 64 * g, a2)
            e("v_lshlrev_b32", a3, 4, a3)
            slots = [self.tw_slot(i) for i in range(7)]
            e("global_load_dwordx4", slots[0], a3, self.TW)
            e("v_lshlrev_b32", a3, 1, a3)
            for j in range(2):
                e("global_load_dwordx4", slots[1 + j], a3, self.TW, offset=16 * j)
            e("v_lshlrev_b32", a3, 1, a3)
            for j in range(4):
                e("global_load_dwordx4", slots[3 + j], a3, self.TW, offset=16 * j)
            for k in range(8):
                e("ds_read_b64", Y[k], addr, offset=k * 144)
            e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
            self.radix8(Y, lambda c, j: tuple(slots[(0, 1, 3)[c] + j].sub(i) for i in range(4)), stage0)
            for k in range(8):
                e("ds_write_b64", addr, Y[k], offset=k * 144)
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_barrier")

    def lds_pass_final(self, half):
        """bits 3..0 (R = 4, PLO = 0): 16 contiguous coefficients, lane-transposed twiddle table."""
        e = self.e
        a0, a1, a2 = self.A_[0], self.A_[1], self.A_[2]
        stage0 = self.logn - 4
        self.c("LDS pass over bits 3..0")
        e("s_movk_i32", self.SC[4], 144)
        e("v_mul_lo_u32", a0, self.TID, self.SC[4])
        e("v_lshlrev_b32", a2, 4, self.TID)                   # t * 16
        Y = [v(2 * k, 2) for k in range(16)]
        for k in range(0, 16, 2):
            e("ds_read_b128", v(2 * k, 4), a0, offset=8 * k)
        # running pointer over the 15 slots of the transposed table: slot stride N/16 entries
        e("s_add_u32", self.TWFR.lo(), self.TWF.lo(), half * 16384)
        e("s_addc_u32", self.TWFR.hi(), self.TWF.hi(), 0)
        slot_regs = {}
        order = [(c, j) for c in range(4) for j in range(1 << c)]
        free = list(range(self.n_tw_slots))

        def issue(cj):
            r = free.pop(0)
            slot_regs[cj] = r
            e("global_load_dwordx4", self.tw_slot(r), a2, self.TWFR)
            e("s_add_u32", self.TWFR.lo(), self.TWFR.lo(), (self.N // 16) * 16)
            e("s_addc_u32", self.TWFR.hi(), self.TWFR.hi(), 0)

        pending = list(order)
        while pending and free:
            issue(pending.pop(0))
        e("s_waitcnt", "lgkmcnt(0)")
        issued = len(order) - len(pending)
        done = 0
        for c in range(4):
            b = 3 - c
            for j in range(1 << c):
                # wait until this twiddle has landed: loads complete in order
                idx = order.index((c, j))
                outstanding_allowed = issued - 1 - idx
                e("s_waitcnt", "vmcnt(%d)" % outstanding_allowed)
                r = slot_regs[(c, j)]
                tw = tuple(self.tw_slot(r).sub(i) for i in range(4))
                for i in range(1 << b):
                    k0 = (j << (b + 1)) | i
                    self.butterfly(Y[k0], Y[k0 | (1 << b)], tw, self.correct_flag(stage0 + c))
                free.append(r)
                done += 1
                if pending:
                    issue(pending.pop(0))
                    issued += 1
        for k in range(0, 16, 2):
            e("ds_write_b128", a0, v(2 * k, 4), offset=8 * k)
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_barrier")

    def copy_out(self, half):
        e = self.e
        a0, a1, a2 = self.A_[0], self.A_[1], self.A_[2]
        self.c("copy-out: canonical reduction + coalesced 16-byte stores")
        e("v_lshrrev_b32", a1, 3, self.TID)
        e("v_lshlrev_b32", a1, 4, a1)
        e("v_lshl_add_u32", a0, self.TID, 4, a1)              # slot(2t) * 8
        e("v_add_u32", a1, 4 * 18432, a0)
        e("v_lshlrev_b32", a2, 4, self.TID)                   # t * 16
        n = self.M // (2 * T)
        regs = [v(4 * i, 4) for i in range(n)]
        for i in range(n):
            base, off = (a0, i * 18432) if i < 4 else (a1, (i - 4) * 18432)
            e("ds_read_b128", regs[i], base, offset=off)
        for i in range(n):
            r = regs[i]
            e("s_waitcnt", "lgkmcnt(%d)" % (n - 1 - i))
            self.canon(r.sub(0, 2))
            self.canon(r.sub(2, 2))
            e("global_store_dwordx4", a2, r, self.DST)
            e("s_add_u32", self.DST.lo(), self.DST.lo(), T * 16)
            e("s_addc_u32", self.DST.hi(), self.DST.hi(), 0)
        if half + 1 < self.HALVES:
            e("s_waitcnt", "lgkmcnt(0)")
            e("s_barrier")

    def build(self):
        self.prologue()
        self.pass_a()
        for half in range(self.HALVES):
            self.lds_write_columns(half)
            self.lds_pass_uniform(half)
            self.lds_pass_lane(half)
            self.lds_pass_final(half)
            self.copy_out(half)
        self.e("s_endpgm")
        return self.p


def kernel_text(logn, name):
    g = Gen(logn)
    prog = g.build()
    lds_bytes = (16384 + 2048) * 8
    hdr = """  .amdgcn_target "amdgcn-amd-amdhsa--gfx950"
  .text
  .globl {name}
  .p2align 8
  .type {name},@function
{name}:
""".format(name=name)
    desc = """
  .rodata
  .p2align 6
  .amdhsa_kernel {name}
    .amdhsa_group_segment_fixed_size {lds}
    .amdhsa_private_segment_fixed_size 0
    .amdhsa_kernarg_size 88
    .amdhsa_user_sgpr_count 2
    .amdhsa_user_sgpr_kernarg_segment_ptr 1
    .amdhsa_system_sgpr_workgroup_id_x 1
    .amdhsa_system_sgpr_workgroup_id_y 1
    .amdhsa_system_sgpr_workgroup_id_z 0
    .amdhsa_system_vgpr_workitem_id 0
    .amdhsa_next_free_vgpr {vgpr}
    .amdhsa_next_free_sgpr 100
    .amdhsa_accum_offset {accum}
    .amdhsa_reserve_vcc 1
    .amdhsa_float_denorm_mode_32 3
    .amdhsa_float_denorm_mode_16_64 3
    .amdhsa_dx10_clamp 1
    .amdhsa_ieee_mode 1
  .end_amdhsa_kernel
  .text

  .amdgpu_metadata
---
amdhsa.kernels:
  - .args:
      - .offset: 0
        .size: 88
        .value_kind: by_value
    .group_segment_fixed_size: {lds}
    .kernarg_segment_align: 8
    .kernarg_segment_size: 88
    .max_flat_workgroup_size: 1024
    .name: {name}
    .private_segment_fixed_size: 0
    .sgpr_count: 106
    .symbol: {name}.kd
    .vgpr_count: {vgpr}
    .wavefront_size: 64
amdhsa.target: amdgcn-amd-amdhsa--gfx950
amdhsa.version: [1, 2]
...
  .end_amdgpu_metadata
""".format(name=name, lds=lds_bytes, vgpr=128, accum=128)
    return hdr + prog.text() + desc, g, prog


# ------------------------------------------------------------------------------------------
# self test on the numpy emulator
# ------------------------------------------------------------------------------------------
def selftest(logn):
    import numpy as np

    from isa import Machine
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), "..", "..", ".."))
    import __graft_entry__ as graft
    oracle = graft.load_oracle()
    pkg = graft.load_package()
    N = 1 << logn
    q = pkg.params.Qi60()[-3]
    oc = oracle.Context(N, [q])
    x = pkg.sampling.random_u64((N,), seed=5)                 # full 64-bit inputs
    x[:4] = np.uint64(0xFFFFFFFFFFFFFFFF)
    want = oc.ntt(np.array([[int(val) % q for val in x]], dtype=np.uint64))[0]

    # host-side tables exactly as lr_abi.cpp builds them
    R64 = 1 << 64
    psi = [int(oracle.inv_mform(int(w), q)) for w in oc.ntt_psi[0]]
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
    lp[7] = red_m | (g << 32)

    # flat memory image (byte addresses)
    def place(arr, addr):
        words = np.ascontiguousarray(arr).view(np.uint32).ravel()
        mem[addr // 4: addr // 4 + words.size] = words

    A_IN, A_OUT, A_LP, A_TW, A_TWF, A_KARG = 0x1000, 0x1000 + 8 * N, 0x100000 * 2, 0x300000, 0x300000 + 16 * N + 0x1000, 0x800
    mem = np.zeros((A_TWF + 16 * 15 * blocks + 0x1000) // 4, dtype=np.uint32)
    place(x, A_IN)
    place(lp, A_LP)
    place(tw, A_TW)
    place(twf, A_TWF)
    karg = np.zeros(11, dtype=np.uint64)
    karg[0], karg[1], karg[2], karg[3] = A_IN, A_OUT, N, N
    karg[4] = 0 | (1 << 32)        # in_limb0, in_limb_step
    karg[5] = 0 | (1 << 32)        # out_limb0, out_limb_step
    karg[6] = 0 | (1 << 32)        # mod0, mod_step
    karg[7] = 1 | (1 << 32)        # n_items, batch
    karg[8], karg[9], karg[10] = A_LP, A_TW, A_TWF
    place(karg, A_KARG)

    gen = Gen(logn)
    prog = gen.build()
    m = Machine(T, 160 * 1024, mem.size)
    m.mem = mem
    m.vgpr[0] = np.arange(T, dtype=np.uint32)
    m.vdef[0] = True
    m.sgpr[0], m.sgpr[1] = A_KARG, 0
    m.sgpr[2], m.sgpr[3] = 0, 0
    m.sdef[0:4] = True
    m.run(prog)
    got = m.mem[A_OUT // 4: A_OUT // 4 + 2 * N].view(np.uint64)
    ok = np.array_equal(got, want)
    cnt = prog.count()
    valu = sum(n for op, n in cnt.items() if op.startswith("v_"))
    bf = (1 << gen.A) * logn // 2
    print("logN=%d emulated workgroup: %s; %d instructions, %d VALU = %.1f per butterfly, s_nop %d" %
          (logn, "bit-exact vs oracle" if ok else "MISMATCH", len(prog.ins), valu, valu / bf, cnt.get("s_nop", 0)))
    if not ok:
        bad = np.nonzero(got != want)[0]
        print("  mismatches:", bad.size, "first:", bad[:8], [hex(int(got[i])) for i in bad[:3]], [hex(int(want[i])) for i in bad[:3]])
    return ok


if __name__ == "__main__":
    logn = int(sys.argv[1])
    if len(sys.argv) > 2 and sys.argv[2] == "--selftest":
        sys.exit(0 if selftest(logn) else 1)
    text, _, _ = kernel_text(logn, "lr_ntt_fwd%d_asm" % logn)
    open(sys.argv[2], "w").write(text)

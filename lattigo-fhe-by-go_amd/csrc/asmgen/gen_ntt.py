"""Generator of the hand-scheduled forward NTT kernels for gfx950 (N = 2^14, 2^15; 1024 threads).

Same algorithm, data layout and twiddle tables as the C++ kernel `ntt_fwd_kernel<LOGN, 1>` in
lr_ntt.hip (pass A in registers -> LDS passes 3+3+4 -> coalesced copy-out, lazy Shoup butterflies
with correction every second stage); what changes is the instruction stream: registers are
assigned by hand, the zero halves of the multiply-accumulate addends live in dedicated registers,
conditional subtractions use the high-word compare, the butterfly sum rides on the multiply-accumulate
chain, and nothing spills.  14 VALU instructions per butterfly without correction, 18 with (the
compiler's version: 21-26).

Three variants per degree (Gen.mode) cover every modulus in (2^33, 2^61); the host picks the cheapest one
the context's largest modulus allows and falls back to the C++ kernels for smaller moduli or degrees.

    python gen_ntt.py 15 out.s [mode] # assembly text
    python gen_ntt.py 15 --selftest   # emulate one workgroup with numpy and compare with a reference NTT
"""
import os
import sys

from isa import VCC, Neg, Program, Reg, s, v

T = 1024
LOGT = 10


# experiment switch: cross terms of the 64-bit products by 32-bit low multiplies (LR_GEN_CROSS32=1 at generation time)
CROSS32 = bool(os.environ.get("LR_GEN_CROSS32"))


class Gen:
    """mode = lazy-correction cadence, all moduli of the launch in (2^33, limit]:
         1: q <= 2^60, U <- U - 8q (if U >= 8q) before every second stage;
         2: q <  2^57, no corrections at all (15 stages x 4q of growth stay below 2^64);
         0: q <  2^61, U <- U - 4q (if U >= 4q) before every stage."""

    def __init__(self, logn, mode=1, threads=1024, sub=False, fused=True, fp=False, dual=False, epi=False, profile=False, persist=False):
        assert logn in (12, 13, 14, 15) and mode in (0, 1, 2) and threads in (256, 512, 1024)
        # persist (forward 2^15, 1024 threads): the workgroup transforms NttLaunch::fuse_top (= polys per workgroup here) consecutive
        # polys of ONE limb in a loop -- constants, table bases and the wave's LDS block stay; the next poly's column loads are
        # issued while this one is in its second LDS image (rows 16..31 into the registers that image has just left, rows 0..15 behind
        # the last copy-out's stores), so the load phase that a one-poly workgroup waits out with nothing else to run (12 k of a
        # wave's 51 k clocks on the FP64 body, 21 k for the wave the first barrier waits for: profiles/r03/timeline_fwd15_ckks.json)
        # and the launch gap between two workgroups of a CU overlap with arithmetic
        assert not persist or (logn == 15 and threads == 1024 and not sub and not epi)
        assert not epi or mode in (1, 2)
        self.persist = persist
        self.karg_parked = False      # timeline builds of kernels that reuse s[0:1]: the inverse ones (gen_intt.py), the persistent ones
        # profile: the timeline build of a plain integer kernel (diagnostics only, Options::timeline): every wave stamps the shader
        # clock at the phase boundaries and the stamps go to the buffer NttLaunch::epi_x points at; the transform itself is unchanged
        assert not profile or not (sub or epi)
        self.profile = profile
        # dual: the kernel carries two bodies behind one prologue (class Dual): `fp` = the FP64 body for moduli below 2^46
        # (error-free products by v_mul_f64 / v_fma_f64, quotients by v_rndne_f64, no lazy corrections: 8 instructions per
        # butterfly), otherwise the integer body of `mode`; the workgroup picks by its limb's entry in NttLaunch::fp_lp
        assert not fp or dual
        # epi: the FP64 body ends in out = (x - NTT(in)) * c + plus (mod q) instead of out = NTT(in): the subtract-multiply of
        # ModDownSplitedNTTPQ (ring_basis_extension.go:237-239) with the addition that follows it in MulRelin
        # (ckks/evaluator.go:1103-1104), x / plus addressed like the output rows (NttLaunch::epi_*).  Launched on FP64 limbs only.
        # (N = 2^16: only the plain sub-block kernels, i.e. after the top stage has been applied by the basis extension)
        # (the integer bodies carry the same epilogue in integer arithmetic -- ops_epilogue_int -- with the constant as a Shoup pair in the
        # same 16 bytes: the dual kernels' integer body for the limbs of 2^46 and more, and the pure integer kernels "m1e")
        self.fp, self.dual, self.epi = fp, dual, epi
        self.karg_parked = bool(profile and persist)
        self.fuse_last = False        # inverse sub-block kernels (gen_intt.py): the last stage by whichever block of the pair finishes second
        self.mark = None
        assert not sub or (logn in (14, 15) and threads == 1024)
        self.fused = fused            # forward sub-block kernels: compute the top stage while loading (out of place only)
        # sub: the kernel transforms one 2^15 half ("sub-block" blk = workgroup x & 1) of an N = 2^16 limb; twiddles come
        # from the 2^16 tables under heap root 2 + blk.  (logn = 14: the halves of an N = 2^15 limb, plain form only -- the "h" kernels
        # that small launches use to put two workgroups on a transform.)  Forward: the stage over bit 15 is computed while loading
        # (X = U + V*psi[1] for blk 0, Y = U - V*psi[1] for blk 1, both blocks read both halves).  Inverse: the
        # outputs stay lazy; ntt_top_kernel finishes with the last stage and the scaling.
        self.sub = sub
        self.NFULL = (2 if sub else 1) << logn
        self.mode = mode
        self.logn = logn
        self.N = 1 << logn
        self.T = threads              # 1024; 512 (N = 2^13, 2^14) / 256 (N = 2^12): smaller LDS image, several workgroups per CU
        self.WAVES = threads // 64
        self.A = logn - 10            # bits consumed by pass A
        self.RA = 1 << self.A         # rows k of a column {k*S + t}
        self.S = self.N >> self.A     # row length = 1024 = size of the sub-transforms the LDS phase works on
        self.C = self.S // self.T     # columns per thread (t and t + T)
        self.NX = self.C * self.RA    # coefficients per thread
        self.HALVES = self.NX // 16
        self.SPH = self.RA // self.HALVES   # sub-transforms resident in LDS at a time: one per wave
        self.M = self.SPH * self.S    # coefficients of one LDS image
        self.RH = 16
        assert self.S == 1024 and self.SPH == self.WAVES and self.NX in (16, 32)
        self.p = Program()
        self.pos = 0                  # issue position for the carry hazard tracker
        self.carry_pos = {}
        # ---- VGPR map.  X[c*RA + k] = coefficient k*S + t + c*T; the values of LDS image h (rows SPH*h .. SPH*h+SPH-1
        # of every column) sit in v[32h .. 32h+31]
        self.X = [None] * self.NX
        for c in range(self.C):
            for k in range(self.RA):
                h, r = divmod(k, self.SPH)
                self.X[c * self.RA + k] = v(32 * h + 2 * (c * self.SPH + r), 2)
        base = 64

        class TS:
            """temporaries of one butterfly in flight; two sets so that two butterflies interleave"""
            def __init__(ts, b, carry):
                ts.T0, ts.Z1, ts.T2, ts.Z3 = v(b + 0), v(b + 1), v(b + 2), v(b + 3)
                ts.T01, ts.T23 = v(b + 0, 2), v(b + 2, 2)
                ts.Q, ts.R, ts.C = v(b + 4, 2), v(b + 6, 2), v(b + 8, 2)
                ts.CY = carry

        self.ts = [TS(base, VCC), TS(base + 10, s(100, 2))]
        self.TID = v(base + 20)
        self.LANE = v(base + 21)
        self.A_ = [v(base + 22 + i) for i in range(4)]
        self.GOFF = self.A_[3]        # t*8, recomputed where needed (pass A loads, column exchange)
        self.tw_base = base + 26
        self.n_tw_slots = 9
        self.vgpr_count = self.tw_base + 4 * self.n_tw_slots
        assert self.vgpr_count <= 128
        # ---- SGPR map
        self.KARG = s(0, 2)
        # launch grid (WGX = limb of the launch, WGY = polynomial of the digit group, WGZ = digit group).  The dispatcher deals
        # consecutive workgroups out to the eight XCDs in turn.  Limb fastest (x = limb): XCD j only sees the limbs j mod 8,
        # whose twiddles stay in its L2 -- best when all limbs cost the same and n_items <= 16.  Polynomial fastest (x = poly):
        # every XCD works on the same limb at any time and sees every modulus -- needed by the dual kernels (a launch that
        # mixes FP64 and integer limbs otherwise waits for the XCD that holds the integer ones) and better for N = 2^16
        # (64 sub-block ids: four limbs' tables per XCD do not fit its L2)
        self.swap_grid = bool(dual or sub)
        self.WGX, self.WGY, self.WGZ = (s(3), s(2), s(4)) if self.swap_grid else (s(2), s(3), s(4))   # WGZ is consumed before SRC (s[4:5]) is formed
        self.SRC, self.DST = s(4, 2), s(6, 2)
        self.TW, self.TWF, self.TWFR = s(8, 2), s(10, 2), s(12, 2)
        self.Qm, self.NQ, self.Q4, self.NQ8 = s(14, 2), s(16, 2), s(18, 2), s(20, 2)   # NQ8: -8q (mode 1) or -4q (mode 0)
        self.U0, self.WAVE = s(22), s(24)
        self.BLK1, self.HI = s(0), s(2, 2)            # sub-block kernels: 1 + blk; high half of the limb (loads only)
        self.TWS = s(2, 2)                            # ... and afterwards the per-stage table base
        self.SC = [s(25 + i) for i in range(7)]      # s25..s31 scratch
        self.JUNK = s(32, 2)
        self.TMP = s(34, 2)
        self.PB, self.QB = 36, 68                     # two 32-dword twiddle buffers
        # dual kernels: table deltas (FP table - integer table, bytes), FP constants of the limb
        # (loaded into s[84:95], inside the second twiddle buffer; the FP body moves what it keeps into the registers
        # that hold the integer body's constants)
        self.DTW, self.DTWF = s(84, 2), s(86, 2)
        self.FPL = s(88, 8)                           # FpLimb: q, 1/q, N^-1 mod q, N^-1 / q as doubles (lr_device.hpp)
        self.QD, self.QINV = s(14, 2), s(16, 2)
        self.MAGIC = s(18, 2)                         # 2^52
        self.NINV, self.NINVQ = s(20, 2), s(22, 2)
        self.BIAS = v(126, 2)                         # 1/(2q), per lane (a second scalar operand is not allowed)
        # persistent kernels: s[0:3] (kernel-argument pointer, workgroup ids) are dead after the prologue
        self.IN_STEP, self.OUT_STEP, self.CNT = s(0, 2), s(2, 2), s(23)
        assert self.vgpr_count <= 126

    # ------------------------------------------------------------------ emission helpers
    def e(self, op, *args, **mods):
        """emit with the VALU-writes-carry -> VALU-reads-carry hazard (2 wait states) handled"""
        reads = []
        if op in ("v_subb_co_u32",):
            reads.append(args[4])
        if op == "v_cndmask_b32":
            reads.append(args[3])
        if op == "s_nop":
            self.pos += int(args[0])
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

    def col_addr(self, c):
        """(offset VGPR, immediate) addressing column t + c*T of a row: GOFF = t*8, A_[2] = GOFF + 4096"""
        off = c * self.T * 8
        return (self.GOFF if off < 4096 else self.A_[2]), off % 4096

    def lds_col(self, c, kk, a0, a1):
        """(base VGPR, immediate) of row kk, column t + c*T in the LDS image: block kk at slot(t + c*T) = slot(t) + 9T/8 * c.
        a1 = a0 + 8 blocks (one column per thread) or a0 + 4608 (several)"""
        if self.C == 1:
            return (a0, kk * 9216) if kk < 8 else (a1, (kk - 8) * 9216)
        off = kk * 9216 + c * self.T * 9
        return (a0, off) if off < 65536 else (a1, off - 4608)

    N_STAMPS = 16

    def stamp_real(self, idx):
        """the same with the constant 100 MHz real-time counter: (shader-clock delta) / (real-time delta) x 100 MHz between the two
        pairs of stamps (13, 14) at the start and (last phase stamp, 15) at the end is the clock the CU ran at"""
        self.stamp(idx, op="s_memrealtime")

    def stamp(self, idx, op="s_memtime"):
        """timeline builds: wave w parks the low word of the shader clock in the padding of row idx of its LDS block (bytes
        w*9216 + idx*144 + 128, never touched by the transform); flush_stamps() copies them out at the end.  s[32:33] is the
        carry-out dump of the multiply-adds (always dead); the stamps sit between phases, where the first butterfly's temporaries
        Q are free in the integer and in the FP64 body alike."""
        if not self.profile:
            return
        assert idx < self.N_STAMPS
        e, J = self.e, self.JUNK
        va, vd = self.ts[0].Q.lo(), self.ts[0].Q.hi()
        e(op, J)
        e("s_waitcnt", "lgkmcnt(0)")
        e("v_mov_b32", vd, J.lo())
        e("s_mul_i32", J.lo(), self.WAVE, 9216)
        e("s_add_u32", J.lo(), J.lo(), idx * 144 + 128)
        e("v_mov_b32", va, J.lo())
        e("ds_write_b32", va, vd)

    def park_stamp_slot(self):
        """timeline builds, in the prologue: the byte offset of this wave's stamps in the stamp buffer -- [workgroup = y * n_items +
        item][wave][N_STAMPS] u32, y = the workgroup's poly (persistent kernels: its chunk of polys) -- waits in bytes 132..135 of
        the wave's row-0 padding, the kernel-argument pointer in bytes 136..143 where the kernel reuses s[0:1]"""
        if not self.profile:
            return
        e, sc = self.e, self.SC
        e("s_mul_i32", sc[0], self.WGY, s(50))                  # n_items
        e("s_add_u32", sc[0], sc[0], self.WGX)
        e("s_lshl_b32", sc[0], sc[0], 4)
        e("s_add_u32", sc[0], sc[0], self.WAVE)
        e("s_lshl_b32", sc[0], sc[0], 6)                        # x N_STAMPS x 4 bytes
        e("s_mul_i32", sc[1], self.WAVE, 9216)
        e("v_mov_b32", v(4), sc[1])
        e("v_mov_b32", v(2), sc[0])
        e("ds_write_b32", v(4), v(2), offset=132)
        if self.karg_parked:
            e("v_mov_b32", v(2), self.KARG.lo())
            e("v_mov_b32", v(3), self.KARG.hi())
            e("ds_write_b64", v(4), v(2, 2), offset=136)
        self.stamp(13)
        self.stamp_real(14)

    def flush_stamps(self, count):
        """stamp buffer (NttLaunch::epi_x): [workgroup][wave][N_STAMPS] u32"""
        e, sc = self.e, self.SC
        self.stamp_real(15)
        count = self.N_STAMPS
        e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
        KA = self.KARG
        e("s_mul_i32", sc[1], self.WAVE, 9216)
        e("v_mov_b32", v(0), sc[1])
        if self.karg_parked:
            KA = s(36, 2)
            e("ds_read_b64", v(2, 2), v(0), offset=136)
            e("s_waitcnt", "lgkmcnt(0)")
            e("v_readfirstlane_b32", KA.lo(), v(2))
            e("v_readfirstlane_b32", KA.hi(), v(3))
            e("s_nop", 4)
        e("s_load_dwordx2", self.TMP, KA, 128)
        e("ds_read_b32", v(126), v(0), offset=132)
        e("s_waitcnt", "lgkmcnt(0)")
        for idx in range(count):
            e("ds_read_b32", v(1), v(0), offset=idx * 144 + 128)
            e("s_waitcnt", "lgkmcnt(0)")
            e("global_store_dword", v(126), v(1), self.TMP, offset=idx * 4)
        e("s_waitcnt", "vmcnt(0)")

    def tw_slot(self, i):
        return v(self.tw_base + 4 * i, 4)

    # ------------------------------------------------------------------ arithmetic macros
    # Each macro returns a list of instructions for one temp set; `zip_emit` interleaves the lists of two
    # independent items so that consecutive instructions of a wave rarely depend on each other.
    @staticmethod
    def pair(r):
        assert r.idx % 2 == 0
        return Reg(r.kind, r.idx, 2)

    # ---- FP64 body.  Values are integer-valued doubles, |x| < 2^50 throughout (q < 2^46: a product term is at most
    # 0.67q in magnitude, 16 stages add at most 11q to the 2^32 + q/2 the ingest leaves), so every sum is exact; a
    # twiddle is the pair (w, RN(w/q)) of doubles.  r = V*w - rint(V * (w/q)) * q: h = RN(V*w), l = V*w - h (exact, fma),
    # h - b*q is an integer below 2^51 in magnitude (exact, fma), r = (h - b*q) + l.
    def ops_modmul_fp(self, ts, V, tw, dst=None):
        """dst (default ts.Q) <- V*w mod q, centred (|.| <= q for |V| <= 2^51, <= 0.67q for |V| < 2^50); dst may be V"""
        W, WQ = self.pair(tw[0]), self.pair(tw[2])
        return [("v_mul_f64", ts.Q, V, W),
                ("v_mul_f64", ts.R, V, WQ),
                ("v_fma_f64", ts.C, V, W, Neg(ts.Q)),
                ("v_rndne_f64", ts.R, ts.R),
                ("v_fma_f64", ts.Q, Neg(ts.R), self.QD, ts.Q),
                ("v_add_f64", dst or ts.Q, ts.Q, ts.C)]

    def ops_butterfly_fp(self, ts, U, V, tw):
        return self.ops_modmul_fp(ts, V, tw) + [("v_add_f64", V, U, Neg(ts.Q)),
                                                ("v_add_f64", U, U, ts.Q)]

    def ops_ingest(self, ts, X):
        """any 64-bit integer -> an integer-valued double congruent to it, |.| <= q/2 + 2^32:
        hi * 2^32 is reduced with the quotient trick (exact: a multiple of 2^32 below 2^64), the low word is added"""
        return [("v_cvt_f64_u32", ts.Q, X.hi()),
                ("v_cvt_f64_u32", ts.C, X.lo()),
                ("v_ldexp_f64", ts.Q, ts.Q, 32),
                ("v_mul_f64", ts.R, ts.Q, self.QINV),
                ("v_rndne_f64", ts.R, ts.R),
                ("v_fma_f64", ts.Q, Neg(ts.R), self.QD, ts.Q),
                ("v_add_f64", X, ts.Q, ts.C)]

    def ops_canon_fp(self, ts, X):
        """integer-valued double, |x| < 2^50 -> canonical residue as a 64-bit integer.  floor((x + 1/2) / q) is the exact
        quotient: x/q is an integer or at least 1/q > 2^-46 away from one, the computed (x + 1/2) * RN(1/q) is off by less
        than 1/(2q) (q < 2^46, |x/q| below 2^12 + 12), so the bias of 1/(2q) keeps it on the right side of every integer.
        Then y = x - k*q is in [0, q) exactly and the mantissa of 2^52 + y is y."""
        return [("v_fma_f64", ts.R, X, self.QINV, self.BIAS),
                ("v_floor_f64", ts.R, ts.R),
                ("v_fma_f64", X, Neg(ts.R), self.QD, X),
                ("v_add_f64", X, X, self.MAGIC),
                ("v_and_b32", X.hi(), 0xFFFFF, X.hi())]

    def ops_butterfly(self, ts, U, V, tw, correct):
        """(U, V) <- (U + r, U + 4q - r), r = V*w - qhat*q in [0,4q); optional U <- U - 8q if U >= 8q first.
        The product accumulates straight onto U (X = U + r costs nothing) and Y = (2U + 4q) - X; all of it
        modulo 2^64, exact because the true X and Y are below 2^64."""
        if self.fp:
            return self.ops_butterfly_fp(ts, U, V, tw)
        w0, w1, s0, s1 = tw
        J = self.JUNK
        ops = []
        hi = [("v_mul_hi_u32", ts.T0, V.hi(), s0),
              ("v_mul_hi_u32", ts.T2, V.lo(), s1)]
        if correct:
            # U <- U - B (B = 8q in mode 1, 4q in mode 0) iff bit 63 of U is set.  Invariant: U < 2^64 before a corrected stage;
            # afterwards U < max(2^63, 2^64 - B), and the stages up to the next correction add at most B <= 2^63, so the
            # values stay below 2^64 (B <= 2^63 because q <= 2^60 resp. q < 2^61).  The test is the sign bit: one full-rate
            # instruction (the 64-bit add) and three plain 32-bit ones that issue in the shadow of the multiplies, instead
            # of add + compare + two selects at full cost each (tools/asm_ubench: v_add_u32 class 2.3 clocks, the others 4.2-4.7).
            M = ts.C
            ops += [("v_ashrrev_i32", M.hi(), 31, U.hi())]
            ops += hi[:1]
            ops += [("v_and_b32", M.lo(), self.NQ8.lo(), M.hi()),
                    ("v_and_b32", M.hi(), self.NQ8.hi(), M.hi())]
            ops += hi[1:]
            ops += [("v_lshl_add_u64", U, U, 0, M)]
        else:
            ops += hi
        if CROSS32:
            # the four cross terms only feed bits 32..63 of the result: 32-bit low products summed by three-input adds instead of
            # 64-bit multiply-adds chained through a 64-bit register (profiles/r02/asm_energy.txt: v_mul_lo_u32 0.65 nJ against
            # 1.8 nJ for v_mad_u64_u32 with a VGPR addend; the kernel runs at the package power limit)
            ops += [
                ("v_lshl_add_u64", ts.R, U, 1, self.Q4),        # 2U + 4q
                ("v_mad_u64_u32", ts.Q, J, V.hi(), s1, ts.T01),
                ("v_mad_u64_u32", U, J, V.lo(), w0, U),
                ("v_mul_lo_u32", ts.C.lo(), V.lo(), w1),
                ("v_lshl_add_u64", ts.Q, ts.Q, 0, ts.T23),      # T0 / T2 are free from here on
                ("v_mul_lo_u32", ts.C.hi(), V.hi(), w0),
                ("v_mad_u64_u32", U, J, ts.Q.lo(), self.NQ.lo(), U),
                ("v_mul_lo_u32", ts.T0, ts.Q.lo(), self.NQ.hi()),
                ("v_mul_lo_u32", ts.T2, ts.Q.hi(), self.NQ.lo()),
                ("v_add3_u32", ts.C.lo(), ts.C.lo(), ts.C.hi(), ts.T0),
                ("v_add3_u32", U.hi(), U.hi(), ts.C.lo(), ts.T2),   # X
                ("v_sub_co_u32", V.lo(), ts.CY, ts.R.lo(), U.lo()),
                ("v_subb_co_u32", V.hi(), ts.CY, ts.R.hi(), U.hi(), ts.CY)]   # Y
            return ops
        ops += [
            ("v_lshl_add_u64", ts.R, U, 1, self.Q4),        # 2U + 4q
            ("v_mad_u64_u32", ts.Q, J, V.hi(), s1, ts.T01),
            ("v_mad_u64_u32", U, J, V.lo(), w0, U),
            ("v_mad_u64_u32", ts.C, J, V.lo(), w1, 0),
            ("v_lshl_add_u64", ts.Q, ts.Q, 0, ts.T23),
            ("v_mad_u64_u32", ts.C, J, V.hi(), w0, ts.C),
            ("v_mad_u64_u32", U, J, ts.Q.lo(), self.NQ.lo(), U),
            ("v_mad_u64_u32", ts.C, J, ts.Q.lo(), self.NQ.hi(), ts.C),
            ("v_mad_u64_u32", ts.C, J, ts.Q.hi(), self.NQ.lo(), ts.C),
            ("v_add_u32", U.hi(), U.hi(), ts.C.lo()),       # X
            ("v_sub_co_u32", V.lo(), ts.CY, ts.R.lo(), U.lo()),
            ("v_subb_co_u32", V.hi(), ts.CY, ts.R.hi(), U.hi(), ts.CY)]   # Y
        return ops

    def ops_mulconst(self, ts, V, tw):
        """V <- V * w - qhat * q, lazy in [0, 4q), for any 64-bit V (the inverse kernels' in-place product, gen_intt.py)"""
        w0, w1, s0, s1 = tw
        J = self.JUNK
        return [("v_mul_hi_u32", ts.T0, V.hi(), s0),
                ("v_mul_hi_u32", ts.T2, V.lo(), s1),
                ("v_mad_u64_u32", ts.Q, J, V.hi(), s1, ts.T01),
                ("v_mad_u64_u32", ts.C, J, V.lo(), w1, 0),
                ("v_lshl_add_u64", ts.Q, ts.Q, 0, ts.T23),
                ("v_mad_u64_u32", ts.C, J, V.hi(), w0, ts.C),
                ("v_mad_u64_u32", V, J, V.lo(), w0, 0),
                ("v_mad_u64_u32", V, J, ts.Q.lo(), self.NQ.lo(), V),
                ("v_mad_u64_u32", ts.C, J, ts.Q.lo(), self.NQ.hi(), ts.C),
                ("v_mad_u64_u32", ts.C, J, ts.Q.hi(), self.NQ.lo(), ts.C),
                ("v_add_u32", V.hi(), V.hi(), ts.C.lo())]

    def ops_epilogue_int(self, ts, Y, X, P, EC):
        """Y (the transform's value, lazy) <- canonical ((x - Y) * c + plus) mod q on the integer pipe; x, plus: any values below 2^63
        (the callers pass canonical residues); c as the Shoup pair (c, floor(c 2^64 / q)) in the 16 bytes an FP64 limb keeps (c, c/q) in:
        Y -> [0, 2q) (5), D = x + 4q - Y in (x + 2q, x + 4q] (3), D * c lazy in [0, 4q) (11), + plus (1), canonical (10)"""
        ops = self.ops_reduce_2q(ts, Y)
        ops += [("v_lshl_add_u64", X, X, 0, self.Q4),
                ("v_sub_co_u32", Y.lo(), ts.CY, X.lo(), Y.lo()),
                ("v_subb_co_u32", Y.hi(), ts.CY, X.hi(), Y.hi(), ts.CY)]
        ops += self.ops_mulconst(ts, Y, EC)
        ops += [("v_lshl_add_u64", Y, Y, 0, P)]
        return ops + self.ops_canon(ts, Y)

    def ops_reduce_2q(self, ts, X):
        """X <- X - floor(X * u0 / 2^64) * q in [0, 2q) for any 64-bit X; u0 = floor(2^64 / q) < 2^32
        (the quotient of ring.BRedAdd, modular_reduction.go:137-146, which is at most one too small)."""
        return [
            ("v_mul_hi_u32", ts.T0, X.lo(), self.U0),
            ("v_mad_u64_u32", ts.Q, self.JUNK, X.hi(), self.U0, ts.T01),      # Q.hi = floor(X * u0 / 2^64)
            ("v_mad_u64_u32", X, self.JUNK, ts.Q.hi(), self.NQ.lo(), X),      # X + k * (2^64 - q), low word and carry
            ("v_mul_lo_u32", ts.T2, ts.Q.hi(), self.NQ.hi()),
            ("v_add_u32", X.hi(), X.hi(), ts.T2),
        ]

    def ops_canon(self, ts, X):
        """X in [0, 16q) -> canonical [0, q)."""
        if self.fp:
            return self.ops_canon_fp(ts, X)
        D = ts.C
        # X in [0, 2q) -> X - q if that is non-negative: the sign of D = X - q as a mask (X < 2q <= 2^62, so bit 63 of D is
        # set exactly when X < q), X = D + (q & mask)
        return self.ops_reduce_2q(ts, X) + [
            ("v_lshl_add_u64", D, X, 0, self.NQ),
            ("v_ashrrev_i32", ts.T0, 31, D.hi()),
            ("v_and_b32", X.lo(), self.Qm.lo(), ts.T0),
            ("v_and_b32", X.hi(), self.Qm.hi(), ts.T0),
            ("v_lshl_add_u64", X, X, 0, D),
        ]

    def zip_emit(self, items):
        """items: list of callables ts -> op list; emitted two at a time, interleaved instruction by instruction"""
        i = 0
        while i < len(items):
            a = items[i](self.ts[0])
            b = items[i + 1](self.ts[1]) if i + 1 < len(items) else []
            for k in range(max(len(a), len(b))):
                if k < len(a):
                    self.e(*a[k])
                if k < len(b):
                    self.e(*b[k])
            i += 2

    def butterflies(self, blist):
        """blist: [(U, V, tw, correct)], all independent"""
        self.zip_emit([(lambda ts, x=x: self.ops_butterfly(ts, *x)) for x in blist])

    def correct_flag(self, stage):
        if self.mode == 2:
            return False
        if self.mode == 0:
            return stage >= 1
        return stage >= 2 and stage % 2 == 0

    # ------------------------------------------------------------------ kernel sections
    def stagger(self):
        """De-synchronise the CUs (NttLaunch::stagger_unit > 0).  All workgroups of a launch start together, one per CU, so every
        CU is in its load phase at the same time (HBM saturated: the phase takes 22 k clocks where the first wave's data is back
        after 6 k), then all compute (HBM idle), then all store.  The workgroups of the first round -- linear id below 256 -- wait
        ((7 * id) mod 16) * stagger_unit kilo-clocks before they start; every CU keeps that phase offset for the rest of the launch
        because its next workgroup starts when this one ends.  Raw hardware ids: x = s2, y = s3, z = s4 (before the prologue
        rewrites them); stagger_gx = the grid's x extent."""
        e, sc = self.e, self.SC
        e("s_load_dwordx2", self.TMP, self.KARG, 168)          # stagger_gx, stagger_unit
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_cmp_eq_u32", self.TMP.hi(), 0)
        e("s_cbranch_scc1", "L_go")
        e("s_cmp_lg_u32", s(4), 0)
        e("s_cbranch_scc1", "L_go")
        e("s_mul_i32", sc[0], s(3), self.TMP.lo())
        e("s_add_u32", sc[0], sc[0], s(2))
        e("s_cmp_ge_u32", sc[0], 256)
        e("s_cbranch_scc1", "L_go")
        e("s_mul_i32", sc[0], sc[0], 7)
        e("s_and_b32", sc[0], sc[0], 15)
        e("s_mul_i32", sc[0], sc[0], self.TMP.hi())
        e("s_cmp_eq_u32", sc[0], 0)
        e("s_cbranch_scc1", "L_go")
        self.p.label("L_wait")
        e("s_sleep", 16)                                        # ~1024 clocks
        e("s_sub_u32", sc[0], sc[0], 1)
        e("s_cmp_lg_u32", sc[0], 0)
        e("s_cbranch_scc1", "L_wait")
        self.p.label("L_go")

    def prologue(self):
        e, S_ = self.e, self
        logn, N = self.logn, self.N
        self.c("kernel arguments (NttLaunch, 176 bytes)")
        e("s_load_dwordx8", s(36, 8), self.KARG, 0)
        e("s_load_dwordx8", s(44, 8), self.KARG, 32)
        e("s_load_dwordx4", s(52, 4), self.KARG, 64)
        e("s_load_dwordx2", s(56, 2), self.KARG, 80)
        e("s_load_dwordx4", s(60, 4), self.KARG, 88)      # sub_log (must be 0 here), hole, group, pad
        if self.dual:
            e("s_load_dwordx4", s(84, 4), self.KARG, 104)     # fp_tw_delta, fp_fin_delta
            e("s_load_dwordx2", s(92, 2), self.KARG, 120)     # fp_lp
        if self.fuse_last:
            e("s_load_dwordx2", s(64, 2), self.KARG, 128)     # epi_x: the pair flags of the launch
        e("v_mov_b32", self.TID, v(0))
        for ts in self.ts:
            e("v_mov_b32", ts.Z1, 0)
            e("v_mov_b32", ts.Z3, 0)
        e("v_lshlrev_b32", self.GOFF, 3, self.TID)
        e("v_and_b32", self.LANE, 63, self.TID)
        e("v_readfirstlane_b32", self.WAVE, self.TID)
        e("s_nop", 4)
        e("s_lshr_b32", self.WAVE, self.WAVE, 6)
        e("s_waitcnt", "lgkmcnt(0)")
        sc = self.SC
        if not self.swap_grid:
            # x = limb of the launch, and consecutive workgroups go to the eight XCDs in turn: the host pads grid x to a multiple of
            # eight (lr_asm.cpp) so that an XCD keeps seeing the same limbs -- their twiddle tables and the epilogue's shared rows stay in
            # its L2 -- and the workgroups beyond the launch's n_items (s50) have nothing to do
            e("s_cmp_lt_u32", self.WGX, s(50))
            e("s_cbranch_scc1", "L_limb_ok")
            e("s_endpgm")
            self.p.label("L_limb_ok")
        self.stagger()
        self.park_stamp_slot()
        if self.sub and self.epi:
            self.park_kernarg()
        if self.persist:
            # grid y counts chunks of s63 (NttLaunch::fuse_top = polys per workgroup) polys inside the group of s62 polys (the host
            # passes group = batch for plain launches): first poly y * P, count = min(P, group - y * P) >= 1
            e("s_mul_i32", self.WGY, self.WGY, s(63))
            e("s_sub_u32", self.CNT, s(62), self.WGY)
            e("s_min_u32", self.CNT, self.CNT, s(63))
        if self.sub:
            e("s_and_b32", self.BLK1, self.WGX, 1)
            e("s_add_u32", self.BLK1, self.BLK1, 1)      # 1 + blk: heap root 2 + blk = 1 + BLK1
            e("s_lshr_b32", self.WGX, self.WGX, 1)
        # grid z = group of polys (key-switch digit): poly = z * group + y, and the limbs [z*hole, (z+1)*hole)
        # of that group are skipped: item = x + (x >= z*hole ? hole : 0).  hole = 0 / z = 0 for plain launches.
        e("s_mul_i32", sc[0], self.WGZ, s(61))
        e("s_cmp_ge_u32", self.WGX, sc[0])
        e("s_cselect_b32", sc[1], s(61), 0)
        e("s_add_u32", self.WGX, self.WGX, sc[1])
        e("s_mul_i32", sc[0], self.WGZ, s(62))
        e("s_add_u32", self.WGY, self.WGY, sc[0])
        if self.sub and self.epi:
            self.park_ids()
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
            e("s_lshl_b32", sc[3], row, logn + (1 if self.sub else 0))
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
        if self.fuse_last:
            self.park_pair_state(sc)
        if self.dual:
            # FpLimb (32 bytes) of this modulus: q = 0.0 marks a limb the FP body does not take
            e("s_lshl_b32", sc[3], sc[0], 5)
            e("s_add_u32", self.TMP.lo(), s(92), sc[3])
            e("s_addc_u32", self.TMP.hi(), s(93), 0)
            e("s_load_dwordx4", s(88, 4), self.TMP, 0)
            e("s_load_dwordx4", s(92, 4), self.TMP, 16)
        # twiddle table bases
        e("s_mul_i32", sc[3], sc[0], self.NFULL * 16)
        e("s_add_u32", self.TW.lo(), s(54), sc[3])
        e("s_addc_u32", self.TW.hi(), s(55), 0)
        e("s_mul_i32", sc[3], sc[0], 15 * self.NFULL)
        e("s_add_u32", self.TWF.lo(), s(56), sc[3])
        e("s_addc_u32", self.TWF.hi(), s(57), 0)
        if self.sub:
            # this block's half of the limb and of the lane-transposed rows
            e("s_sub_u32", sc[3], self.BLK1, 1)
            e("s_lshl_b32", sc[3], sc[3], logn)                    # blk * N
            e("s_add_u32", self.TWF.lo(), self.TWF.lo(), sc[3])   # blk * (N/16) entries of 16 bytes
            e("s_addc_u32", self.TWF.hi(), self.TWF.hi(), 0)
            e("s_lshl_b32", sc[3], sc[3], 3)
            e("s_add_u32", self.DST.lo(), self.DST.lo(), sc[3])
            e("s_addc_u32", self.DST.hi(), self.DST.hi(), 0)
            self.sub_source(sc[3])
        if self.persist:
            # loop state: bytes from one poly of the launch to the next (the output pointer has moved on by one LDS image per poly)
            e("s_lshl_b64", self.IN_STEP, s(40, 2), 3)
            e("s_lshl_b64", self.OUT_STEP, s(42, 2), 3)
            e("s_sub_u32", self.OUT_STEP.lo(), self.OUT_STEP.lo(), self.M * 8 * (self.HALVES - 1))
            e("s_subb_u32", self.OUT_STEP.hi(), self.OUT_STEP.hi(), 0)
        self.prologue_tail()

    def park_kernarg(self):
        """sub-block kernels reuse s[0:3] (BLK1, HI / TWS): what the epilogue needs of them -- the kernel-argument pointer, the
        limb and the polynomial of the workgroup -- waits in the padding of row 0 of the wave's LDS block (bytes w*9216 + 128 ..
        143, never touched by the transform; the timeline build keeps its stamps in the same padding)"""
        e = self.e
        e("s_mul_i32", self.SC[0], self.WAVE, 9216)
        e("s_add_u32", self.SC[0], self.SC[0], 128)
        e("v_mov_b32", v(4), self.SC[0])
        e("v_mov_b32", v(2), self.KARG.lo())
        e("v_mov_b32", v(3), self.KARG.hi())
        e("ds_write_b64", v(4), v(2, 2))

    def park_pair_state(self, sc):
        """inverse sub-block kernels with the fused last stage: the address of this wave's pair flag (flags[(poly * n_items + limb) * 16
        + wave], u32) and of the limb's LimbParams wait in the same padding as park_kernarg's"""
        e = self.e
        e("s_mul_i32", sc[3], self.WGY, s(50))                 # poly * n_items
        e("s_add_u32", sc[3], sc[3], self.WGX)
        e("s_lshl_b32", sc[3], sc[3], 4)
        e("s_add_u32", sc[3], sc[3], self.WAVE)
        e("s_lshl_b32", sc[3], sc[3], 2)
        e("s_add_u32", s(64), s(64), sc[3])
        e("s_addc_u32", s(65), s(65), 0)
        e("s_mul_i32", sc[3], self.WAVE, 9216)
        e("s_add_u32", sc[3], sc[3], 128)
        e("v_mov_b32", v(4), sc[3])
        e("v_mov_b32", v(2), s(64))
        e("v_mov_b32", v(3), s(65))
        e("ds_write_b64", v(4), v(2, 2))
        e("v_mov_b32", v(2), self.TMP.lo())
        e("v_mov_b32", v(3), self.TMP.hi())
        e("ds_write_b64", v(4), v(2, 2), offset=8)

    def park_ids(self):
        e = self.e
        e("v_mov_b32", v(2), self.WGX)
        e("v_mov_b32", v(3), self.WGY)
        e("ds_write_b64", v(4), v(2, 2), offset=8)

    def unpark(self, dst):
        """dst[0:1] <- kernel-argument pointer, dst[2] <- limb of the launch, dst[3] <- polynomial"""
        e = self.e
        ts = self.ts[0]
        e("s_mul_i32", self.SC[0], self.WAVE, 9216)
        e("s_add_u32", self.SC[0], self.SC[0], 128)
        e("v_mov_b32", ts.T0, self.SC[0])
        e("ds_read_b128", v(ts.Q.idx, 4), ts.T0)
        e("s_waitcnt", "lgkmcnt(0)")
        for i in range(4):
            e("v_readfirstlane_b32", dst.sub(i), v(ts.Q.idx + i))
        e("s_nop", 4)

    def sub_source(self, blk_bytes):
        """forward: SRC stays at the low half, HI = the high half (fused top stage reads both); without the fusion
        the block reads its own half, which ntt_top_kernel has prepared"""
        e = self.e
        if not self.fused:
            e("s_add_u32", self.SRC.lo(), self.SRC.lo(), blk_bytes)
            e("s_addc_u32", self.SRC.hi(), self.SRC.hi(), 0)
            return
        e("s_add_u32", self.HI.lo(), self.SRC.lo(), self.N * 8)
        e("s_addc_u32", self.HI.hi(), self.SRC.hi(), 0)

    def stage_table(self, c):
        """SGPR pair addressing the twiddles of pass-A stage c: under heap root 2 + blk the index (1 << c) + j of a whole
        transform becomes ((2 + blk) << c) + j"""
        if not self.sub:
            return self.TW
        e = self.e
        e("s_lshl_b32", self.SC[3], self.BLK1, c + 4)
        e("s_add_u32", self.TWS.lo(), self.TW.lo(), self.SC[3])
        e("s_addc_u32", self.TWS.hi(), self.TW.hi(), 0)
        return self.TWS

    def ops_mulacc(self, ts, U, V, tw):
        """U <- U + (V*w - qhat*q), the product in [0,4q) for any 64-bit V"""
        if self.fp:
            return self.ops_modmul_fp(ts, V, tw) + [("v_add_f64", U, U, ts.Q)]
        w0, w1, s0, s1 = tw
        J = self.JUNK
        return [("v_mul_hi_u32", ts.T0, V.hi(), s0),
                ("v_mul_hi_u32", ts.T2, V.lo(), s1),
                ("v_mad_u64_u32", ts.Q, J, V.hi(), s1, ts.T01),
                ("v_mad_u64_u32", U, J, V.lo(), w0, U),
                ("v_mad_u64_u32", ts.C, J, V.lo(), w1, 0),
                ("v_lshl_add_u64", ts.Q, ts.Q, 0, ts.T23),
                ("v_mad_u64_u32", ts.C, J, V.hi(), w0, ts.C),
                ("v_mad_u64_u32", U, J, ts.Q.lo(), self.NQ.lo(), U),
                ("v_mad_u64_u32", ts.C, J, ts.Q.lo(), self.NQ.hi(), ts.C),
                ("v_mad_u64_u32", ts.C, J, ts.Q.hi(), self.NQ.lo(), ts.C),
                ("v_add_u32", U.hi(), U.hi(), ts.C.lo())]

    def fused_top(self):
        """N = 2^16, forward: X[k] <- U_k + V_k * w with U from the low half of the limb, V from the high half and
        w = psi[1] (sub-block 0) or q - psi[1] (sub-block 1; kept at index 0 of the forward table)."""
        e = self.e
        W1 = tuple(s(96 + i) for i in range(4))
        e("s_sub_u32", self.SC[3], 2, self.BLK1)
        e("s_lshl_b32", self.SC[3], self.SC[3], 4)
        e("s_load_dwordx4", s(96, 4), self.TW, self.SC[3])
        e("s_mov_b64", self.TMP, self.SRC)
        for k in range(self.RA):
            e("global_load_dwordx2", self.X[k], self.GOFF, self.TMP, hint="nt")
            e("s_add_u32", self.TMP.lo(), self.TMP.lo(), self.S * 8)
            e("s_addc_u32", self.TMP.hi(), self.TMP.hi(), 0)
        V = [v(self.tw_base + 2 * i, 2) for i in range(16)]     # the twiddle pool is idle until the LDS phase

        def v_loads():
            for i in range(16):
                e("global_load_dwordx2", V[i], self.GOFF, self.HI, hint="nt")
                e("s_add_u32", self.HI.lo(), self.HI.lo(), self.S * 8)
                e("s_addc_u32", self.HI.hi(), self.HI.hi(), 0)

        v_loads()
        self.constants()
        if self.fp:
            e("s_load_dwordx4", s(96, 4), self.TW, self.SC[3])     # the same entry of the FP table
            e("s_waitcnt", "lgkmcnt(0)")
        for chunk in range(2):
            for i in range(0, 16, 2):
                e("s_waitcnt", "vmcnt(%d)" % (14 - i))
                if self.fp:
                    self.zip_emit([(lambda ts, U=self.X[16 * chunk + i + d], Vr=V[i + d]:
                                    self.ops_ingest(ts, U) + self.ops_ingest(ts, Vr) + self.ops_mulacc(ts, U, Vr, W1)) for d in range(2)])
                else:
                    self.zip_emit([(lambda ts, U=self.X[16 * chunk + i + d], Vr=V[i + d]:
                                    self.ops_reduce_2q(ts, U) + self.ops_mulacc(ts, U, Vr, W1)) for d in range(2)])
            if chunk == 0:
                v_loads()

    def constants(self):
        e = self.e
        if self.dual:
            self.mark = len(self.p.ins)               # everything before this point is common to the two bodies
        if self.fp:
            e("s_waitcnt", "lgkmcnt(0)")
            e("s_cmp_eq_u32", self.FPL.sub(1), 0)
            e("s_cbranch_scc1", "INT_BODY")
            for i, dst in enumerate((self.QD, self.QINV)):       # (N^-1 is the inverse's; s23 counts polys in the persistent kernels)
                e("s_mov_b64", dst, self.FPL.sub(2 * i, 2))
            for ptr, d in ((self.TW, self.DTW), (self.TWF, self.DTWF)):
                e("s_add_u32", ptr.lo(), ptr.lo(), d.lo())
                e("s_addc_u32", ptr.hi(), ptr.hi(), d.hi())
            e("s_mov_b32", self.MAGIC.lo(), 0)
            e("s_mov_b32", self.MAGIC.hi(), 0x43300000)
            e("v_mul_f64", self.BIAS, self.QINV, 0.5)
            return
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_mov_b64", self.Qm, s(68, 2))
        e("s_mov_b32", self.U0, s(72))                # low word of bred_hi = floor(2^64 / q)
        e("s_sub_u32", self.NQ.lo(), 0, self.Qm.lo())
        e("s_subb_u32", self.NQ.hi(), 0, self.Qm.hi())
        e("s_lshl_b64", self.Q4, self.Qm, 2)
        e("s_lshl_b64", self.TMP, self.Qm, 2 if self.mode == 0 else 3)
        e("s_sub_u32", self.NQ8.lo(), 0, self.TMP.lo())
        e("s_subb_u32", self.NQ8.hi(), 0, self.TMP.hi())

    def prologue_tail(self):
        e = self.e
        if self.sub and self.fused:
            self.fused_top()
            return
        if self.persist:
            self.c("first poly: rows 16..31, then rows 0..15 (the order every later poly's prefetch has)")
            self.load_rows(self.RA // 2, self.RA)
            self.load_rows(0, self.RA // 2)
            self.constants()
            return
        self.c("coalesced load of the columns {k*S + t + c*T}")
        self.stamp(0)
        # order 0, RA/2, 1, RA/2+1, ... (per column): the first-stage butterflies can start after two loads
        e("s_add_u32", self.TMP.lo(), self.SRC.lo(), (self.RA // 2) * self.S * 8)
        e("s_addc_u32", self.TMP.hi(), self.SRC.hi(), 0)
        if self.C > 1:
            e("v_add_u32", self.A_[2], 4096, self.GOFF)
        for k in range(self.RA // 2):
            for c in range(self.C):
                off, imm = self.col_addr(c)
                e("global_load_dwordx2", self.X[c * self.RA + k], off, self.SRC, offset=imm, hint="nt")
                e("global_load_dwordx2", self.X[c * self.RA + k + self.RA // 2], off, self.TMP, offset=imm, hint="nt")
            for ptr in (self.SRC, self.TMP):
                e("s_add_u32", ptr.lo(), ptr.lo(), self.S * 8)
                e("s_addc_u32", ptr.hi(), ptr.hi(), 0)
        self.constants()

    def load_rows(self, k0, k1):
        """persistent kernels: column loads of rows k0..k1-1 of the poly at SRC (t*8 is recomputed: GOFF is a scratch register of the
        LDS passes); TMP walks, SRC stays"""
        e = self.e
        e("v_lshlrev_b32", self.GOFF, 3, self.TID)
        e("s_add_u32", self.TMP.lo(), self.SRC.lo(), k0 * self.S * 8)
        e("s_addc_u32", self.TMP.hi(), self.SRC.hi(), 0)
        for k in range(k0, k1):
            e("global_load_dwordx2", self.X[k], self.GOFF, self.TMP, hint="nt")
            if k + 1 < k1:
                e("s_add_u32", self.TMP.lo(), self.TMP.lo(), self.S * 8)
                e("s_addc_u32", self.TMP.hi(), self.TMP.hi(), 0)

    def prefetch_rows(self, k0, k1, advance):
        """persistent kernels: the next poly's rows k0..k1-1, if there is a next poly (CNT counts this one too)"""
        e = self.e
        tag = "%s_%d" % ("fp" if self.fp else "int", k0)
        e("s_cmp_lt_u32", self.CNT, 2)
        e("s_cbranch_scc1", "L_nopf_" + tag)
        if advance:
            e("s_add_u32", self.SRC.lo(), self.SRC.lo(), self.IN_STEP.lo())
            e("s_addc_u32", self.SRC.hi(), self.SRC.hi(), self.IN_STEP.hi())
        self.load_rows(k0, k1)
        self.p.label("L_nopf_" + tag)

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
            table = self.stage_table(c)
            while dw > 0:
                n = 16 if dw >= 16 else dw
                e({4: "s_load_dwordx4", 8: "s_load_dwordx8", 16: "s_load_dwordx16"}[n], s(dst, n), table, off)
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
            pend = []
            for j in range(1 << c):
                if split and j == (1 << c) // 2:
                    # second half of the twiddles goes where the previous stage's lived
                    self.butterflies(pend)
                    pend = []
                    load_stage(c, bufs[(c + 1) % 2], 1)
                    e("s_waitcnt", "lgkmcnt(0)")
                    cur = bufs[(c + 1) % 2] - 4 * j
                tw = tuple(s(cur + 4 * j + i) for i in range(4))
                blist = []
                for i in range(1 << b):
                    k0 = (j << (b + 1)) | i
                    k1 = k0 | (1 << b)
                    for col in range(self.C):
                        blist.append((self.X[col * RA + k0], self.X[col * RA + k1], tw, self.correct_flag(c)))
                if c == 0:
                    # loads were issued in the order of this list: pair i needs the first 2i+2
                    for i in range(0, len(blist), 2):
                        if self.persist:
                            # rows 16..31 were issued first; rows 0..15 are the youngest operations of the queue, in order
                            e("s_waitcnt", "vmcnt(%d)" % (self.RA // 2 - 2 - i))
                        else:
                            e("s_waitcnt", "vmcnt(%d)" % max(self.NX - 2 * (i + 2), 0))
                        # first-stage U operands: any 64-bit value is accepted
                        if self.fp:
                            if not (self.sub and self.fused):     # (the fused top stage has converted its operands)
                                self.zip_emit([(lambda ts, x=blist[i + d][k]: self.ops_ingest(ts, x)) for d in range(2) for k in range(2)])
                        else:
                            self.zip_emit([(lambda ts, x=blist[i + d][0]: self.ops_reduce_2q(ts, x)) for d in range(2)])
                        self.butterflies(blist[i:i + 2])
                else:
                    pend.extend(blist)
            if c > 0:
                self.butterflies(pend)
            else:
                self.stamp(1)                 # every column load has returned, stage 0 is done

    # ------------------------------------------------------------------ LDS phase
    # After pass A the resident half consists of 16 independent sub-transforms of 1024 coefficients
    # (index bits 9..0).  Wave w owns sub-transform w for ALL remaining stages: 64 lanes x 16
    # coefficients.  The only workgroup barriers are the ones around the column exchange; between
    # them the 16 waves run decoupled (LDS operations of one wave execute in order), so the LDS and
    # twiddle latencies of one wave hide behind the arithmetic of the others.
    def uniform_twiddle_loads(self, half):
        """scalar twiddles of the stages over bits 9..7: H = 2^(logn-10) + SPH*half + wave"""
        e, sc = self.e, self.SC
        buf = self.PB
        e("s_add_u32", sc[2], self.WAVE, (1 << (self.logn - 10)) + self.SPH * half)
        if self.sub:
            e("s_lshl_b32", sc[3], self.BLK1, self.logn - 10)
            e("s_add_u32", sc[2], sc[2], sc[3])
        e("s_lshl_b32", sc[3], sc[2], 4)
        e("s_load_dwordx4", s(buf, 4), self.TW, sc[3])
        e("s_lshl_b32", sc[3], sc[2], 5)
        e("s_load_dwordx8", s(buf + 8, 8), self.TW, sc[3])
        e("s_lshl_b32", sc[3], sc[2], 6)
        e("s_load_dwordx16", s(buf + 16, 16), self.TW, sc[3])

    def lds_write_columns(self, half):
        e = self.e
        a0, a1, a2 = self.A_[0], self.A_[1], self.A_[2]
        self.c("half %d -> LDS image (16 B of padding per 16 coefficients)" % half)
        if self.persist and half == 0:
            e("s_barrier")            # every wave has read the previous poly's last image out (its copy-out loads have returned)
        self.uniform_twiddle_loads(half)
        e("v_lshrrev_b32", a2, 4, self.TID)
        e("v_lshlrev_b32", a2, 4, a2)
        e("v_lshl_add_u32", a0, self.TID, 3, a2)              # slot(t) * 8
        # row kk of column c goes to block kk at slot(t + c*T) = slot(t) + 576*c
        e("v_add_u32", a1, 8 * 9216 if self.C == 1 else 4608, a0)
        for c in range(self.C):
            for kk in range(self.SPH):
                base, off = self.lds_col(c, kk, a0, a1)
                e("ds_write_b64", base, self.X[c * self.RA + self.SPH * half + kk], offset=off)
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_barrier")

    def radix8(self, Y, twf, stage0, hook=None):
        """three stages on 8 coefficients; hook(n) runs after the n-th twiddle has been consumed"""
        n = 0
        for c in range(3):
            b = 2 - c
            blist, marks = [], []
            for j in range(1 << c):
                tw = twf(c, j)
                for i in range(1 << b):
                    k0 = (j << (b + 1)) | i
                    blist.append((Y[k0], Y[k0 | (1 << b)], tw, self.correct_flag(stage0 + c)))
                marks.append(len(blist))
            self.butterflies(blist)
            for _ in marks:
                n += 1
                if hook:
                    hook(n)

    # ---- streaming of per-lane twiddles through the small pool of VGPR quads -----------------------------
    # `stream_begin(requests)` takes the ordered list of (key, issue_fn(slot_reg)); as many as there are free
    # quads are issued at once, the rest as quads are released.  vector-memory operations complete in issue
    # order, so "key has landed" == "at most (issued - 1 - position(key)) operations outstanding".
    def stream_begin(self, requests, base_outstanding=0):
        self.st_keys = [k for k, _ in requests]
        self.st_fn = dict(requests)
        self.st_pending = list(self.st_keys)
        self.st_slot = {}
        self.st_issued = 0
        self.st_other = base_outstanding      # younger foreign operations are not allowed while a stream is live
        if not hasattr(self, "st_free"):
            self.st_free = list(range(self.n_tw_slots))
        self.stream_pump()

    def stream_pump(self):
        while self.st_pending and self.st_free:
            k = self.st_pending.pop(0)
            r = self.st_free.pop(0)
            self.st_slot[k] = r
            self.st_fn[k](self.tw_slot(r))
            self.st_issued += 1

    def stream_wait(self, key, also="") :
        idx = self.st_keys.index(key)
        assert key in self.st_slot, "twiddle %r was never issued (pool too small for the access order)" % (key,)
        self.e("s_waitcnt", ("vmcnt(%d) " % (self.st_issued - 1 - idx) + also).strip())
        r = self.tw_slot(self.st_slot[key])
        return tuple(r.sub(i) for i in range(4))

    def stream_release(self, key):
        self.st_free.append(self.st_slot[key])
        self.stream_pump()

    def lane_twiddle_requests(self, half, g, tag):
        """the 7 twiddles of the stages over bits 6..4 for task g, in heap order; H*16 is kept in A_[3]
        H = 2^(logn-7) + 8*(SPH*half + wave) + (lane >> 4) + 4*g"""
        e = self.e
        a3 = self.A_[3]

        def addr(shift):
            def f():
                e("v_lshrrev_b32", a3, 4, self.LANE)
                e("s_lshl_b32", self.SC[5], self.WAVE, 3)
                e("s_add_u32", self.SC[5], self.SC[5], (1 << (self.logn - 7)) + 8 * self.SPH * half + 4 * g)
                if self.sub:
                    e("s_lshl_b32", self.SC[4], self.BLK1, self.logn - 7)
                    e("s_add_u32", self.SC[5], self.SC[5], self.SC[4])
                e("v_add_u32", a3, self.SC[5], a3)
                e("v_lshlrev_b32", a3, 4 + shift, a3)
            return f

        reqs = []
        for c in range(3):
            for j in range(1 << c):
                def issue(slot, c=c, j=j):
                    addr(c)()
                    e("global_load_dwordx4", slot, a3, self.TW, offset=16 * j)
                reqs.append(((tag, c, j), issue))
        return reqs

    def wave_lds_base(self):
        """SC[6] <- wave * 9216: byte offset of this wave's 1024-coefficient block in the padded image"""
        self.e("s_mul_i32", self.SC[6], self.WAVE, 9216)

    def lds_pass_uniform(self, half):
        """bits 9..7 (R = 3): lane l owns the columns l and l + 64 of its wave's block; twiddles wave-uniform."""
        e = self.e
        a0, a1 = self.A_[0], self.A_[1]
        stage0 = self.logn - 10
        self.c("stages over bits 9..7 (wave-local)")
        self.wave_lds_base()
        e("v_lshrrev_b32", a1, 4, self.LANE)
        e("v_lshlrev_b32", a1, 4, a1)
        e("v_lshl_add_u32", a0, self.LANE, 3, a1)             # slot(l) * 8
        e("v_add_u32", a0, self.SC[6], a0)
        YA = [v(2 * k, 2) for k in range(8)]
        YB = [v(16 + 2 * k, 2) for k in range(8)]
        for k in range(8):
            e("ds_read_b64", YA[k], a0, offset=k * 1152)
        for k in range(8):
            e("ds_read_b64", YB[k], a0, offset=k * 1152 + 576)    # column l + 64: 64*8 + 4*16 bytes further
        twf = lambda c, j: tuple(s(self.PB + (0, 8, 16)[c] + 4 * j + i) for i in range(4))
        # per-lane twiddles of the next stages start travelling now (no other vector-memory traffic is live)
        self.stream_begin(self.lane_twiddle_requests(half, 0, "a") + self.lane_twiddle_requests(half, 1, "b"))
        e("s_waitcnt", "lgkmcnt(8)")
        self.radix8(YA, twf, stage0)
        for k in range(8):
            e("ds_write_b64", a0, YA[k], offset=k * 1152)
        e("s_waitcnt", "lgkmcnt(8)")                              # task 1's reads (issued before the 8 writes)
        self.radix8(YB, twf, stage0)
        for k in range(8):
            e("ds_write_b64", a0, YB[k], offset=k * 1152 + 576)

    def lds_pass_lane(self, half):
        """bits 6..4 (R = 3): lane l owns rows (l >> 4) and (l >> 4) + 4, column l & 15; per-lane twiddles."""
        e = self.e
        a0, a2 = self.A_[0], self.A_[2]
        stage0 = self.logn - 7
        self.c("stages over bits 6..4 (wave-local)")
        e("v_lshrrev_b32", a2, 4, self.LANE)
        e("s_movk_i32", self.SC[4], 1152)                    # VOP3 takes no literal on gfx9
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
        for tag, Y, off, lg in (("a", YA, 0, "lgkmcnt(8)"), ("b", YB, 4608, "lgkmcnt(8)")):
            first = [True]

            def twf(c, j, tag=tag, lg=lg):
                tw = self.stream_wait((tag, c, j), lg if first[0] else "")
                first[0] = False
                return tw

            order = [(tag, c, j) for c in range(3) for j in range(1 << c)]
            self.radix8(Y, twf, stage0, hook=lambda n, order=order: self.stream_release(order[n - 1]))
            for k in range(8):
                e("ds_write_b64", a0, Y[k], offset=k * 144 + off)
        # the last four stages' twiddles (lane-transposed table) follow in the same stream discipline
        self.final_prefetch(half)

    def final_requests(self, half):
        """the 15 lane-transposed twiddles of the stages over bits 3..0, table row (2^c - 1 + j), in heap order"""
        e = self.e
        a2 = self.A_[2]

        def make(c, j):
            def issue(slot):
                e("v_lshlrev_b32", a2, 4, self.TID)           # block index half*1024 + 64*wave + lane = half*1024 + t
                row = (1 << c) - 1 + j
                off = half * self.SPH * 1024 + row * (self.NFULL // 16) * 16  # block index SPH*64*half + t, 16 B each
                e("s_add_u32", self.TWFR.lo(), self.TWF.lo(), off)
                e("s_addc_u32", self.TWFR.hi(), self.TWF.hi(), 0)
                e("global_load_dwordx4", slot, a2, self.TWFR)
            return issue

        return [(("f", c, j), make(c, j)) for c in range(4) for j in range(1 << c)]

    def final_prefetch(self, half):
        self.stream_begin(self.final_requests(half))

    def lds_pass_final(self, half):
        """bits 3..0 (R = 4): lane l owns the 16 contiguous coefficients of block l; lane-transposed twiddles."""
        e = self.e
        a0 = self.A_[0]
        stage0 = self.logn - 4
        self.c("stages over bits 3..0 (wave-local)")
        e("s_movk_i32", self.SC[4], 144)
        e("v_mul_lo_u32", a0, self.LANE, self.SC[4])
        e("v_add_u32", a0, self.SC[6], a0)
        Y = [v(2 * k, 2) for k in range(16)]
        for k in range(0, 16, 2):
            e("ds_read_b128", v(2 * k, 4), a0, offset=8 * k)
        e("s_waitcnt", "lgkmcnt(0)")
        for c in range(4):
            b = 3 - c
            # twiddles are consumed two at a time so that butterflies of different twiddles can interleave
            js = list(range(1 << c))
            for j0 in range(0, len(js), 2):
                group = js[j0:j0 + 2]
                blist = []
                for j in group:
                    tw = self.stream_wait(("f", c, j))
                    for i in range(1 << b):
                        k0 = (j << (b + 1)) | i
                        blist.append((Y[k0], Y[k0 | (1 << b)], tw, self.correct_flag(stage0 + c)))
                self.butterflies(blist)
                for j in group:
                    self.stream_release(("f", c, j))
        for k in range(0, 16, 2):
            e("ds_write_b128", a0, v(2 * k, 4), offset=8 * k)

    def copy_out(self, half):
        """wave w stores its 1024 finished coefficients (8 KiB contiguous): canonical reduction + 16-byte stores"""
        e = self.e
        a0, a1, a2 = self.A_[0], self.A_[1], self.A_[2]
        self.c("copy-out (wave-local)")
        e("v_lshrrev_b32", a1, 3, self.LANE)
        e("v_lshlrev_b32", a1, 4, a1)
        e("v_lshl_add_u32", a0, self.LANE, 4, a1)             # slot(2l) * 8
        e("v_add_u32", a0, self.SC[6], a0)
        e("v_lshlrev_b32", a2, 4, self.LANE)                  # l * 16
        # destination of this wave: dst + (half*16 + wave) * 8192
        e("s_lshl_b32", self.SC[5], self.WAVE, 13)
        e("s_add_u32", self.TMP.lo(), self.DST.lo(), self.SC[5])
        e("s_addc_u32", self.TMP.hi(), self.DST.hi(), 0)
        n = 8
        regs = [v(4 * i, 4) for i in range(n)]
        if self.epi:
            self.copy_out_epilogue(half, regs, a0, a2)
        else:
            for i in range(n):
                e("ds_read_b128", regs[i], a0, offset=i * 1152)
            for i in range(n):
                r = regs[i]
                e("s_waitcnt", "lgkmcnt(%d)" % (n - 1 - i))
                self.zip_emit([(lambda ts, x=r.sub(0, 2): self.ops_canon(ts, x)), (lambda ts, x=r.sub(2, 2): self.ops_canon(ts, x))])
                e("global_store_dwordx4", a2, r, self.TMP, hint="nt")
                e("s_add_u32", self.TMP.lo(), self.TMP.lo(), 1024)
                e("s_addc_u32", self.TMP.hi(), self.TMP.hi(), 0)
        if half + 1 < self.HALVES:
            e("s_add_u32", self.DST.lo(), self.DST.lo(), self.M * 8)
            e("s_addc_u32", self.DST.hi(), self.DST.hi(), 0)
            e("s_waitcnt", "lgkmcnt(0)")
            e("s_barrier")
        elif self.persist:
            # v0..v31 are free once the last store has been issued (a store reads its data when it issues)
            self.prefetch_rows(0, self.RA // 2, advance=False)
            e("s_add_u32", self.DST.lo(), self.DST.lo(), self.OUT_STEP.lo())
            e("s_addc_u32", self.DST.hi(), self.DST.hi(), self.OUT_STEP.hi())

    def ops_epilogue(self, ts, Y, X, P, EC):
        """Y (the transform's value, a lazy double) <- canonical ((x - Y) * c + plus) mod q; X, P: canonical 64-bit integers"""
        ops = []
        for R in (X, P):
            ops += [("v_or_b32", R.hi(), 0x43300000, R.hi()),
                    ("v_add_f64", R, R, Neg(self.MAGIC))]
        ops += [("v_add_f64", Y, X, Neg(Y))]
        ops += self.ops_modmul_fp(ts, Y, EC, dst=Y)
        ops += [("v_add_f64", Y, Y, P)]
        return ops + self.ops_canon_fp(ts, Y)

    def copy_out_epilogue(self, half, regs, a0, a2):
        """copy-out with the subtract-multiply-add epilogue: x and plus live at the output's offsets inside their own rows"""
        e = self.e
        sc = self.SC
        K = 36                                       # the uniform-twiddle buffer is idle here: scratch SGPRs
        KA, WX, WY = self.KARG, self.WGX, self.WGY
        if self.sub:
            KA, WX, WY = s(K + 16, 2), s(K + 18), s(K + 19)
            self.unpark(s(K + 16, 4))
        e("s_load_dwordx8", s(K, 8), KA, 128)                 # epi_x, epi_x_stride, epi_plus, epi_plus_stride
        e("s_load_dwordx2", s(K + 8, 2), KA, 160)             # epi_consts
        e("s_load_dwordx4", s(K + 12, 4), KA, 40)             # out_limb0, out_limb_step, mod0, mod_step
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_mul_i32", sc[0], WX, s(K + 13))
        e("s_add_u32", sc[0], sc[0], s(K + 12))               # output row
        e("s_mul_i32", sc[1], WX, s(K + 15))
        e("s_add_u32", sc[1], sc[1], s(K + 14))               # modulus index
        rows = []
        for base, stride, dst in ((s(K, 2), s(K + 2, 2), s(K + 20, 2)), (s(K + 4, 2), s(K + 6, 2), s(K + 22, 2))):
            e("s_mul_i32", self.TMP.lo(), WY, stride.lo())
            e("s_mul_hi_u32", self.TMP.hi(), WY, stride.lo())
            e("s_mul_i32", sc[3], WY, stride.hi())
            e("s_add_u32", self.TMP.hi(), self.TMP.hi(), sc[3])
            e("s_lshl_b32", sc[3], sc[0], self.logn + (1 if self.sub else 0))
            e("s_add_u32", self.TMP.lo(), self.TMP.lo(), sc[3])
            e("s_addc_u32", self.TMP.hi(), self.TMP.hi(), 0)
            e("s_lshl_b64", self.TMP, self.TMP, 3)
            e("s_add_u32", dst.lo(), base.lo(), self.TMP.lo())
            e("s_addc_u32", dst.hi(), base.hi(), self.TMP.hi())
            # this wave's 8 KiB: (half * SPH + wave) * 8192 bytes into the row (+ the sub-block's half of the limb)
            e("s_lshl_b32", sc[3], self.WAVE, 13)
            if self.sub:
                e("s_sub_u32", sc[5], self.BLK1, 1)
                e("s_lshl_b32", sc[5], sc[5], self.logn + 3)
                e("s_add_u32", sc[3], sc[3], sc[5])
            e("s_add_u32", sc[3], sc[3], half * self.SPH * 8192)
            e("s_add_u32", dst.lo(), dst.lo(), sc[3])
            e("s_addc_u32", dst.hi(), dst.hi(), 0)
            rows.append(dst)
        XR, PR = rows
        e("s_lshl_b32", sc[3], sc[1], 4)
        e("s_add_u32", self.TMP.lo(), s(K + 8), sc[3])
        e("s_addc_u32", self.TMP.hi(), s(K + 9), 0)
        e("s_load_dwordx4", s(K + 24, 4), self.TMP, 0)         # EpiLimb: c, c / q as doubles (FP64 limbs) / c and its Shoup companion
        e("s_waitcnt", "lgkmcnt(0)")                           # (scalar loads return out of order: no counting across them)
        EC = tuple(s(K + 24 + i) for i in range(4))
        # restore the store pointer (TMP was scratch): dst + wave * 8192
        e("s_lshl_b32", sc[5], self.WAVE, 13)
        e("s_add_u32", self.TMP.lo(), self.DST.lo(), sc[5])
        e("s_addc_u32", self.TMP.hi(), self.DST.hi(), 0)
        n = 8
        for i in range(n):
            e("ds_read_b128", regs[i], a0, offset=i * 1152)
        XQ = [v(self.tw_base + 8 * i, 4) for i in range(4)]
        PQ = [v(self.tw_base + 8 * i + 4, 4) for i in range(4)]
        # LR_GEN_EXPERIMENT=noepiload (measurement only, wrong results): the epilogue's operands come from nowhere -- what the launch gains is
        # everything their loads cost, bandwidth and exposed latency together (profiles/r04/epilogue_load_cost.txt)
        noload = os.environ.get("LR_GEN_EXPERIMENT") == "noepiload"
        for batch in range(2):
            for i in range(4):
                k = 4 * batch + i
                if noload:
                    for r in range(4):
                        e("v_mov_b32", v(self.tw_base + 8 * i + r), 0)
                        e("v_mov_b32", v(self.tw_base + 8 * i + 4 + r), 0)
                    continue
                e("global_load_dwordx4", XQ[i], a2, XR, offset=(k % 4) * 1024, hint="nt")
                # plus: default cache policy, not streaming like x: the rounding rescale's plus operand is ONE table row per limb shared
                # by every poly of the launch, and with nt each workgroup fetched it from memory again (4.7 GB per launch of 256 polys x
                # 15 limbs instead of 2.1, profiles/r03/pmc_rescale15.json; Rescale PN15QP880 0.97 -> 0.88 ms per batch of 128); the
                # per-poly plus rows of MulRelin measure the same either way
                e("global_load_dwordx4", PQ[i], a2, PR, offset=(k % 4) * 1024)
            for ptr in (XR, PR):
                e("s_add_u32", ptr.lo(), ptr.lo(), 4096)
                e("s_addc_u32", ptr.hi(), ptr.hi(), 0)
            for i in range(4):
                k = 4 * batch + i
                r = regs[k]
                # loads return in order: at most the 2*(3-i) younger loads of this batch may still be out (stores issued
                # in between can only make the wait longer)
                e("s_waitcnt", ("lgkmcnt(%d)" % (n - 1 - k)) if noload else "vmcnt(%d) lgkmcnt(%d)" % (2 * (3 - i), n - 1 - k))
                epi = self.ops_epilogue if self.fp else self.ops_epilogue_int
                self.zip_emit([(lambda ts, y=r.sub(0, 2), x=XQ[i].sub(0, 2), pp=PQ[i].sub(0, 2): epi(ts, y, x, pp, EC)),
                               (lambda ts, y=r.sub(2, 2), x=XQ[i].sub(2, 2), pp=PQ[i].sub(2, 2): epi(ts, y, x, pp, EC))])
            for i in range(4):
                k = 4 * batch + i
                e("global_store_dwordx4", a2, regs[k], self.TMP, hint="nt")
                e("s_add_u32", self.TMP.lo(), self.TMP.lo(), 1024)
                e("s_addc_u32", self.TMP.hi(), self.TMP.hi(), 0)

    # stamps of the timeline build, in order (tools/timeline.py names the intervals between them)
    STAMP_NAMES = ["start", "loads returned + stage 0", "pass A done"] + [
        "%s (half %d)" % (n, h) for h in range(2) for n in ("column exchange", "stages over bits 9..7", "stages over bits 6..4",
                                                             "last four stages", "copy-out issued")]

    def build(self):
        self.prologue()
        tag = "fp" if self.fp else "int"
        if self.persist:
            self.p.label("L_poly_" + tag)
            self.stamp(0)
        self.pass_a()
        self.stamp(2)
        for half in range(self.HALVES):
            self.lds_write_columns(half)
            if self.persist and half == 1:
                # v32..v63 have just gone to the LDS image: the next poly's rows 16..31 travel while this image is worked on
                self.prefetch_rows(self.RA // 2, self.RA, advance=True)
            self.stamp(3 + 5 * half)
            self.lds_pass_uniform(half)
            self.stamp(4 + 5 * half)
            self.lds_pass_lane(half)
            self.stamp(5 + 5 * half)
            self.lds_pass_final(half)
            self.stamp(6 + 5 * half)
            self.copy_out(half)
            self.stamp(7 + 5 * half)
        if self.persist:
            self.e("s_sub_u32", self.CNT, self.CNT, 1)
            self.e("s_cmp_lg_u32", self.CNT, 0)
            self.e("s_cbranch_scc1", "L_poly_" + tag)
        if self.profile:
            self.flush_stamps(3 + 5 * self.HALVES)     # (persistent kernels: the stamps of the workgroup's last poly)
        self.e("s_endpgm")
        return self.p


class Dual:
    """two bodies behind one prologue: FP64 for limbs whose FpLimb entry is set, the integer body otherwise"""

    def __init__(self, make):
        self.gf, self.gi = make(True), make(False)
        for k in ("logn", "T", "SPH", "A", "N", "sub", "WGX", "WGY", "epi", "fuse_last"):
            setattr(self, k, getattr(self.gf, k))
        self.p = None

    def build(self):
        pf, pi = self.gf.build(), self.gi.build()
        mf, mi = self.gf.mark, self.gi.mark
        assert mf is not None and mf == mi and repr(pf.ins[:mf]) == repr(pi.ins[:mi]), "the two bodies must share their prologue"
        body_labels = [a[0] for op, a, _ in pf.ins[mf:] + pi.ins[mi:] if op == "@"]
        assert len(body_labels) == len(set(body_labels))                          # the bodies' own labels must not collide
        p = Program()
        p.ins = list(pf.ins)
        p.label("INT_BODY")
        p.ins += pi.ins[mi:]
        self.p = p
        return p


def kernel_text(logn, name):
    g = Gen(logn)
    return kernel_text_for(g, name), g, g.p


def kernel_text_for(g, name):
    prog = g.build()
    lds_bytes = g.SPH * 9216
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
    .amdhsa_kernarg_size 176
    .amdhsa_user_sgpr_count 2
    .amdhsa_user_sgpr_kernarg_segment_ptr 1
    .amdhsa_system_sgpr_workgroup_id_x 1
    .amdhsa_system_sgpr_workgroup_id_y 1
    .amdhsa_system_sgpr_workgroup_id_z 1
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
        .size: 176
        .value_kind: by_value
    .group_segment_fixed_size: {lds}
    .kernarg_segment_align: 8
    .kernarg_segment_size: 176
    .max_flat_workgroup_size: {threads}
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
""".format(name=name, lds=lds_bytes, vgpr=128, accum=128, threads=g.T)
    return hdr + prog.text() + desc


FP_LIMIT = 1 << 46


if __name__ == "__main__":
    logn = int(sys.argv[1])
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    threads = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    # mode 3: FP64 body for the limbs below 2^46, the integer body of mode 2 for the others (every modulus below 2^57)
    def make(logn_, threads_, **kw):
        if mode == 5:   # "m5": the integer kernel of mode 1 (q <= 2^60) with the subtract-multiply-add epilogue
            return Gen(logn_, 1, threads_, epi=True, **kw)
        if mode == 3:
            return Dual(lambda fp: Gen(logn_, 2, threads_, fp=fp, dual=True, **kw))
        if mode == 4:   # mode 3 whose FP64 body ends in the subtract-multiply-add epilogue
            return Dual(lambda fp: Gen(logn_, 2, threads_, fp=fp, dual=True, epi=True, **kw))
        return Gen(logn_, mode, threads_, **kw)

    if logn == 16:      # the 2^15 sub-block kernels of N = 2^16: "s" with the fused top stage, "p" plain
        fused = not (len(sys.argv) > 5 and sys.argv[5] == "plain")
        open(sys.argv[2], "w").write(kernel_text_for(make(15, 1024, sub=True, fused=fused),
                                                     "lr_ntt_fwd16%s_m%d" % ("s" if fused else "p", mode)))
        sys.exit(0)
    if len(sys.argv) > 5 and sys.argv[5] == "halves":        # N = 2^15 as two 2^14 sub-blocks, plain (the top stage applied before): small launches
        assert logn == 15
        open(sys.argv[2], "w").write(kernel_text_for(make(14, 1024, sub=True, fused=False), "lr_ntt_fwd15h_m%d" % mode))
        sys.exit(0)
    if len(sys.argv) > 5 and sys.argv[5] == "timeline":      # diagnostics build with per-phase clock stamps (Options::timeline)
        open(sys.argv[2], "w").write(kernel_text_for(make(logn, threads, profile=True), "lr_ntt_fwd%d_m%dt" % (logn, mode)))
        sys.exit(0)
    if len(sys.argv) > 5 and sys.argv[5] in ("persist", "persist-timeline"):    # several polys per workgroup, next poly's loads prefetched
        tl = sys.argv[5] == "persist-timeline"
        open(sys.argv[2], "w").write(kernel_text_for(make(logn, threads, persist=True, profile=tl), "lr_ntt_fwd%dp_m%d%s" % (logn, mode, "t" if tl else "")))
        sys.exit(0)
    name = "lr_ntt_fwd%d%s_m%d" % (logn, "x" if threads < 1024 else "", mode)     # x: several workgroups per CU
    open(sys.argv[2], "w").write(kernel_text_for(make(logn, threads), name))

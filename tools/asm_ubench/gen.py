"""Instruction-level throughput probes for the assembly NTT kernels: each variant is a register-only loop of one
instruction pattern, 1024 threads per workgroup, one workgroup per CU.  Reports cycles per wave-instruction per SIMD.
    python gen.py outdir        # writes <variant>.s ; assemble + link like csrc/build.sh
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "lattigo-fhe-by-go_amd", "csrc", "asmgen"))
from gen_ntt import Gen, kernel_text_for  # noqa: E402
from gen_intt import GenInv  # noqa: E402
from isa import VCC, s, v  # noqa: E402

ITER = 256


class Probe(Gen):
    def __init__(self, variant):
        super().__init__(15, 1)
        self.variant = variant
        self.REM = s(23)

    def body(self):
        e, X = self.e, self.X
        tw = tuple(s(36 + i) for i in range(4))
        var = self.variant
        if var.startswith("cfg_"):
            # cfg_<ts0 carry>_<twiddle base>[_c]: interleaved butterflies with the given SGPR placement
            parts = var.split("_")
            c0, twb = parts[1], int(parts[2])
            self.ts[0].CY = VCC if c0 == "vcc" else s(int(c0), 2)
            tw = tuple(s(twb + i) for i in range(4))
            for i in range(4):
                pass
            corr = len(parts) > 3
            self.butterflies([(X[i], X[i + 16], tw, corr) for i in range(16)])
            return 16 * (18 if corr else 14)
        if var in ("bfly_nc", "bfly_c"):
            self.butterflies([(X[i], X[i + 16], tw, var == "bfly_c") for i in range(16)])
            return 16 * (14 if var == "bfly_nc" else 18)
        if var == "bfly_nc_2sgpr":    # both temp sets carry through SGPR pairs
            self.ts[0].CY = s(98, 2)
            self.butterflies([(X[i], X[i + 16], tw, False) for i in range(16)])
            return 16 * 14
        if var == "bfly_c_seq":
            for i in range(16):
                for op in self.ops_butterfly(self.ts[0], X[i], X[i + 16], tw, True):
                    e(*op)
            return 16 * 18
        if var == "fp_real":
            # the generator's own FP64 butterflies (gen_ntt.py: ops_butterfly_fp), 16 of them, two interleaved
            self.fp = True
            twf = tuple(s(36 + i) for i in range(4))
            self.butterflies([(X[i], X[i + 16], twf, False) for i in range(16)])
            self.fp = False
            return 16 * 8
        if var == "fp64_bfly":
            # the butterfly an FP64 path would use for moduli <= 50 bits: error-free product (mul + fma), quotient by
            # multiplication with 1/q and round-to-nearest, exact remainder by fma, then X = U + r, Y = U - r
            n = 0
            for rep in range(2):
                for i in range(0, 16, 2):
                    for d in range(2):          # two butterflies interleaved by hand
                        pass
                    A = [(X[i + d], X[i + d + 16], self.ts[d]) for d in range(2)]
                    seq = lambda U, V, ts: [
                        ("v_mul_f64", ts.Q, V, s(36, 2)),
                        ("v_fma_f64", ts.R, V, s(36, 2), "-" + repr(ts.Q)),
                        ("v_mul_f64", ts.C, ts.Q, s(38, 2)),
                        ("v_rndne_f64", ts.C, ts.C),
                        ("v_fma_f64", ts.Q, "-" + repr(ts.C), s(40, 2), ts.Q),
                        ("v_add_f64", ts.Q, ts.Q, ts.R),
                        ("v_add_f64", V, U, "-" + repr(ts.Q)),
                        ("v_add_f64", U, U, ts.Q)]
                    a, b = seq(*A[0]), seq(*A[1])
                    for k in range(len(a)):
                        e(*a[k])
                        e(*b[k])
                    n += 16
            return n
        if var.startswith("xchg_"):
            # 16 butterflies (one radix-2 stage over the 32 coefficients a thread holds) followed by one exchange of the 16
            # V-side coefficients between lanes, the way a stage boundary of the LDS passes moves data:
            #   xchg_none   no exchange (baseline)
            #   xchg_lds    through the LDS: 16 ds_write_b64 + 16 ds_read_b64 (what the kernels do; no VALU slot used)
            #   xchg_perm   two index bits (lane bits 5 and 4) by v_permlane32_swap / v_permlane16_swap: 2 x 16 VALU
            #   xchg_dpp    one index bit inside a row by DPP moves: 3 v_mov_b32_dpp per 32-bit register pair = 48 VALU
            # (north_star's "wavefront shuffles for the intra-warp stages": a transposition of k index bits costs
            # 16 / 16 / 48 ... VALU instructions per bit on top of the butterflies; the LDS round trip costs none)
            self.butterflies([(X[i], X[i + 16], tw, False) for i in range(16)])
            n = 16 * 14
            if var == "xchg_lds":
                e("v_lshlrev_b32", self.A_[0], 3, self.TID)
                e("v_add_u32", self.A_[1], 0x10000, self.A_[0])
                for i in range(16):
                    e("ds_write_b64", self.A_[i // 8], X[16 + i], offset=8192 * (i % 8))
                e("s_waitcnt", "lgkmcnt(0)")
                for i in range(16):
                    j = (i + 1) % 16
                    e("ds_read_b64", X[16 + i], self.A_[j // 8], offset=8192 * (j % 8))
                e("s_waitcnt", "lgkmcnt(0)")
            elif var == "xchg_perm":
                for i in range(0, 16, 2):
                    a, b = X[16 + i], X[16 + i + 1]
                    e("v_permlane32_swap_b32", a.lo(), b.lo())
                    e("v_permlane32_swap_b32", a.hi(), b.hi())
                for i in range(0, 16, 2):
                    a, b = X[16 + i], X[16 + (i + 2) % 16]
                    e("v_permlane16_swap_b32", a.lo(), b.lo())
                    e("v_permlane16_swap_b32", a.hi(), b.hi())
                n += 32
            elif var == "xchg_dpp":
                T = self.ts[0].T0
                for i in range(0, 16, 2):
                    for half in ("lo", "hi"):
                        a, b = getattr(X[16 + i], half)(), getattr(X[16 + i + 1], half)()
                        e("v_mov_b32_dpp", T, b, dpp="quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                        e("v_mov_b32_dpp", b, a, dpp="quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0x5")
                        e("v_mov_b32_dpp", a, T, dpp="quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xa")
                n += 48
            return n
        if var == "bfly_nc_seq":      # no interleaving of two butterflies
            for i in range(16):
                for op in self.ops_butterfly(self.ts[0], X[i], X[i + 16], tw, False):
                    e(*op)
            return 16 * 14
        if var == "inv_c":
            gi = self.inv
            self.zip_emit([(lambda ts, U=X[i], V=X[i + 16]: gi.ops_butterfly(ts, U, V, tw, True)) for i in range(16)])
            return 16 * 19
        n = 128
        for i in range(n):
            d = X[i % 32]
            a = X[(i + 7) % 32]
            if var == "mad_vsv":
                e("v_mad_u64_u32", d, self.JUNK, a.lo(), s(36), d)
            elif var == "mad_vvv":
                e("v_mad_u64_u32", d, self.JUNK, a.lo(), a.hi(), d)
            elif var == "mad_vs0":
                e("v_mad_u64_u32", d, self.JUNK, a.lo(), s(36), 0)
            elif var == "mulhi_vs":
                e("v_mul_hi_u32", d.lo(), a.hi(), s(36))
            elif var == "mulhi_vv":
                e("v_mul_hi_u32", d.lo(), a.hi(), a.lo())
            elif var == "mullo_vv":
                e("v_mul_lo_u32", d.lo(), a.hi(), a.lo())
            elif var == "lshl_add":
                e("v_lshl_add_u64", d, a, 0, s(36, 2))
            elif var == "lshl_add_vvv":
                e("v_lshl_add_u64", d, a, 0, d)
            elif var == "add_u32":
                e("v_add_u32", d.lo(), a.lo(), d.hi())
            elif var == "cndmask_vcc":
                e("v_cndmask_b32", d.lo(), a.lo(), a.hi(), VCC)
            elif var == "cndmask_sgpr":
                e("v_cndmask_b32", d.lo(), a.lo(), a.hi(), s(100, 2))
            elif var == "sub_co_pair":
                if i % 2 == 0:
                    e("v_sub_co_u32", d.lo(), VCC, a.lo(), d.lo())
                else:
                    e("v_subb_co_u32", d.hi(), VCC, a.hi(), d.hi(), VCC)
            elif var == "cmp":
                e("v_cmp_lt_u32", VCC if i % 2 else s(100, 2), a.hi(), d.hi())
            elif var == "mad_sdst_vcc":
                e("v_mad_u64_u32", d, VCC, a.lo(), s(36), d)
            elif var == "mad_sdst_alt":
                e("v_mad_u64_u32", d, s(40 + 2 * (i % 4), 2), a.lo(), s(36), d)
            elif var == "subb_sgpr":
                if i % 2 == 0:
                    e("v_sub_co_u32", d.lo(), s(100, 2), a.lo(), d.lo())
                else:
                    e("v_subb_co_u32", d.hi(), s(100, 2), a.hi(), d.hi(), s(100, 2))
            elif var == "cmp_vcc":
                e("v_cmp_lt_u32", VCC, a.hi(), d.hi())
            elif var == "cmp_sgpr":
                e("v_cmp_lt_u32", s(100, 2), a.hi(), d.hi())
            elif var == "cndmask_vcc2":
                if i == 0:
                    e("v_cmp_lt_u32", VCC, a.hi(), d.hi())
                    e("s_nop", 4)
                e("v_cndmask_b32", d.lo(), a.lo(), a.hi(), VCC)
            elif var == "cndmask_e64_vcc":
                e("v_cndmask_b32_e64", d.lo(), a.lo(), a.hi(), VCC)
            elif var == "mad_add_mix":
                if i % 2 == 0:
                    e("v_mad_u64_u32", d, self.JUNK, a.lo(), s(36), d)
                else:
                    e("v_add_u32", d.lo(), a.lo(), d.hi())
            elif var.startswith("madadd_"):
                # runs of k multiply-adds followed by k plain adds: do adjacent plain 32-bit instructions issue at their own (double) rate
                # inside a multiply stream?
                k = int(var.split("_")[1])
                if (i // k) % 2 == 0:
                    e("v_mad_u64_u32", d, self.JUNK, a.lo(), s(36), d)
                else:
                    e("v_add_u32", d.lo(), a.lo(), d.hi())
            elif var.startswith("madsub_"):
                # the same with the carry pair the butterfly ends in: k x (v_sub_co_u32 on distinct registers), then k x v_subb_co_u32
                k = int(var.split("_")[1])
                ph = (i // k) % 4
                if ph in (0, 1):
                    e("v_mad_u64_u32", d, self.JUNK, a.lo(), s(36), d)
                elif ph == 2:
                    e("v_sub_co_u32", d.lo(), VCC if i % 2 else s(100, 2), a.lo(), d.lo())
                else:
                    e("v_subb_co_u32", d.hi(), VCC if i % 2 else s(100, 2), a.hi(), d.hi(), VCC if i % 2 else s(100, 2))
            elif var == "mulhi_add_mix":
                if i % 2 == 0:
                    e("v_mul_hi_u32", d.lo(), a.hi(), s(36))
                else:
                    e("v_add_u32", d.lo(), a.lo(), d.hi())
            elif var == "xor_b32":
                e("v_xor_b32", d.lo(), a.lo(), d.hi())
            elif var == "and_or":
                e("v_and_or_b32", d.lo(), a.lo(), a.hi(), d.lo())
            elif var == "add3":
                e("v_add3_u32", d.lo(), a.lo(), a.hi(), d.lo())
            elif var == "lshl_add_u32":
                e("v_lshl_add_u32", d.lo(), a.lo(), 3, d.lo())
            elif var == "mov":
                e("v_mov_b32", d.lo(), a.hi())
            elif var == "fma_f64":
                e("v_fma_f64", d, a, a, d)
            elif var == "mul_f64":
                e("v_mul_f64", d, a, d)
            elif var == "add_f64":
                e("v_add_f64", d, a, d)
            elif var == "rndne_f64":
                e("v_rndne_f64", d, a)
            elif var == "floor_f64":
                e("v_floor_f64", d, a)
            elif var == "cvt_f64_u32":
                e("v_cvt_f64_u32", d, a.lo())
            elif var == "ldexp_f64":
                e("v_ldexp_f64", d, a, 32)
            elif var == "fma_rnd_mix":      # the FP64 butterfly's mix: seven multiply / add / fma class instructions per v_rndne_f64
                if i % 8 == 3:
                    e("v_rndne_f64", d, a)
                else:
                    e("v_fma_f64", d, a, a, d)
            elif var == "fma_f32":
                e("v_fma_f32", d.lo(), a.lo(), a.hi(), d.lo())
            else:
                raise ValueError(var)
        return n

    def build_mem(self):
        """burst = what one wave of the NTT kernel does at its start: 32 x dwordx2 (or 16 x dwordx4, 8 x dwordx4 twice...)
        covering its share of a 256 KiB limb, then s_waitcnt vmcnt(0); one workgroup per CU, ITER items per workgroup"""
        e = self.e
        var = self.variant
        e("s_load_dwordx2", self.DST, self.KARG, 16)
        e("s_load_dwordx2", self.SRC, self.KARG, 24)
        e("v_mov_b32", self.TID, v(0))
        e("v_and_b32", self.LANE, 63, self.TID)
        e("v_readfirstlane_b32", self.WAVE, self.TID)
        e("s_nop", 4)
        e("s_lshr_b32", self.WAVE, self.WAVE, 6)
        e("s_waitcnt", "lgkmcnt(0)")
        # item base: (workgroup id * ITER) * 256 KiB, wrapped into a 1 GiB buffer by the host-side size choice
        e("s_and_b32", self.SC[0], self.WGX, 255)
        e("s_lshl_b32", self.SC[0], self.SC[0], 24)            # 64 items of 256 KiB per workgroup slot, 4 GiB in all
        if "l2" in var:
            e("s_mov_b32", self.SC[0], 0)                       # every workgroup reads the same 256 KiB again and again: L2 hits
        e("s_add_u32", self.SRC.lo(), self.SRC.lo(), self.SC[0])
        e("s_addc_u32", self.SRC.hi(), self.SRC.hi(), 0)
        wide = "x4" in var
        if wide:
            e("v_lshlrev_b32", self.GOFF, 4, self.TID)          # t*16: rows of 16 KiB
        else:
            e("v_lshlrev_b32", self.GOFF, 3, self.TID)          # t*8: rows of 8 KiB
        e("s_movk_i32", self.REM, 64)
        self.p.label("L_top")
        e("s_mov_b64", self.TMP, self.SRC)
        n = 16 if wide else 32
        for k in range(n):
            if wide:
                e("global_load_dwordx4", v(4 * (k % 16), 4), self.GOFF, self.TMP, hint="nt")
            else:
                e("global_load_dwordx2", self.X[k], self.GOFF, self.TMP, hint="nt")
            e("s_add_u32", self.TMP.lo(), self.TMP.lo(), 16384 if wide else 8192)
            e("s_addc_u32", self.TMP.hi(), self.TMP.hi(), 0)
        e("s_waitcnt", "vmcnt(0)")
        if "work" in var:
            # stand-in for the compute phase: ~30 us of dependent VALU per item
            for i in range(2000):
                e("v_mad_u64_u32", self.X[i % 32], self.JUNK, self.X[(i + 7) % 32].lo(), s(36), self.X[i % 32])
        if "l2" not in var:
            e("s_add_u32", self.SRC.lo(), self.SRC.lo(), 262144)
            e("s_addc_u32", self.SRC.hi(), self.SRC.hi(), 0)
        e("s_sub_u32", self.REM, self.REM, 1)
        e("s_cmp_eq_u32", self.REM, 0)
        e("s_cbranch_scc0", "L_top")
        e("v_lshlrev_b32", self.GOFF, 3, self.TID)
        e("global_store_dwordx2", self.GOFF, self.X[0], self.DST)
        e("s_endpgm")
        self.count = 64
        if "l2" in var or "hbm" in var:
            self.count = 64 * n // ITER if (64 * n) % ITER == 0 else 64 * n     # energy mode: wave-instructions = count x ITER per wave
            self.count = 64 * n
            self.mem_energy = True
        return self.p

    def build(self):
        e = self.e
        if self.variant.startswith("mem_"):
            return self.build_mem()
        self.inv = GenInv(15, 1)
        self.inv.p, self.inv.e = self.p, self.e
        e("s_load_dwordx2", self.DST, self.KARG, 16)
        e("s_load_dwordx4", s(36, 4), self.KARG, 0)
        e("v_mov_b32", self.TID, v(0))
        for ts in self.ts:
            e("v_mov_b32", ts.Z1, 0)
            e("v_mov_b32", ts.Z3, 0)
        for k in range(32):
            e("v_lshlrev_b32", self.X[k].lo(), k % 7, self.TID)
            e("v_add_u32", self.X[k].hi(), 12345 + k, self.TID)
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_mov_b64", self.NQ, s(36, 2))
        e("s_mov_b64", self.Q4, s(38, 2))
        e("s_mov_b64", self.NQ8, s(36, 2))
        e("s_mov_b64", self.NQ2 if hasattr(self, "NQ2") else self.Qm, s(38, 2))
        e("s_mov_b64", VCC, 0)
        e("s_mov_b64", s(100, 2), 0)
        for b in (40, 64, 68, 92, 96):
            e("s_mov_b64", s(b, 2), s(36, 2))
            e("s_mov_b64", s(b + 2, 2), s(38, 2))
        e("s_movk_i32", self.REM, ITER)
        self.p.label("L_top")
        self.count = self.body()
        e("s_sub_u32", self.REM, self.REM, 1)
        e("s_cmp_eq_u32", self.REM, 0)
        e("s_cbranch_scc0", "L_top")
        # keep the result alive
        e("v_lshlrev_b32", self.GOFF, 3, self.TID)
        for k in range(1, 32):
            e("v_xor_b32", self.X[0].lo(), self.X[0].lo(), self.X[k].lo())
            e("v_xor_b32", self.X[0].hi(), self.X[0].hi(), self.X[k].hi())
        e("global_store_dwordx2", self.GOFF, self.X[0], self.DST)
        e("s_endpgm")
        return self.p


PAIRING = ["mad_vsv", "add_u32", "madadd_1", "madadd_2", "madadd_4", "madadd_8", "madsub_2", "madsub_4"]
ENERGY = ["mov", "xor_b32", "add_u32", "lshl_add", "lshl_add_vvv", "mullo_vv", "mulhi_vv", "mulhi_vs", "mad_vs0", "mad_vsv", "mad_vvv", "fma_f32", "fma_f64", "bfly_nc", "bfly_c", "fp64_bfly", "xchg_lds"]
MEMPOWER = ["mem_hbm_x4", "mem_l2_x4", "mem_hbm_x2", "mem_l2_x2"]
FP64 = ["fma_f64", "mul_f64", "add_f64", "rndne_f64", "floor_f64", "cvt_f64_u32", "ldexp_f64", "fma_rnd_mix", "fp_real", "bfly_nc", "add_u32"]
VARIANTS = ["mad_add_mix", "mulhi_add_mix", "xchg_none", "xchg_lds", "xchg_perm", "xchg_dpp", "mulhi_vs", "mad_vs0", "mad_vsv", "lshl_add", "add_u32", "sub_co_pair", "cndmask_vcc", "fma_f64", "bfly_nc", "bfly_c"]
_OLD2 = ["cfg_98_36", "cfg_32_36", "cfg_2_36", "cfg_32_68", "cfg_vcc_68", "cfg_98_92", "cfg_98_40", "cfg_34_64", "cfg_2_36_c", "cfg_98_36_c", "cfg_32_36_c"]
_OLD = ["bfly_nc", "bfly_c", "bfly_nc_seq", "bfly_c_seq", "bfly_nc_2sgpr", "inv_c", "mad_sdst_vcc", "mad_sdst_alt", "subb_sgpr", "cmp_vcc", "cmp_sgpr",
            "cndmask_vcc2", "cndmask_e64_vcc", "mad_add_mix", "mulhi_add_mix", "xor_b32", "and_or", "add3", "lshl_add_u32", "mov", "mad_vsv", "mad_vvv", "mad_vs0", "mulhi_vs", "mulhi_vv", "mullo_vv",
            "lshl_add", "lshl_add_vvv", "add_u32", "cndmask_vcc", "cndmask_sgpr", "sub_co_pair", "cmp", "fma_f64", "fma_f32"]

if __name__ == "__main__":
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    meta = []
    for var in ({"energy": ENERGY, "pairing": PAIRING, "fp64": FP64, "mempower": MEMPOWER}.get(sys.argv[2] if len(sys.argv) > 2 else "", VARIANTS)):
        g = Probe(var)
        text = kernel_text_for(g, "probe_" + var)
        open(os.path.join(out, var + ".s"), "w").write(text)
        meta.append("%s %d" % (var, g.count if getattr(g, "mem_energy", False) else g.count * ITER))
    open(os.path.join(out, "variants.txt"), "w").write("\n".join(meta) + "\n")

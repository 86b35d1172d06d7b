#!/usr/bin/env python3
"""Per-phase timeline of the 2^15 NTT kernels: where a wave's and a workgroup's life goes.

    python tools/timeline.py [fwd|inv] [qi60|ckks] [out.json]

Runs 256 polys x 16 limbs (the bench shape) on the stamped build of the kernel the context selects -- qi60: ring.DefaultParamsQi[15]
(integer body, lr_ntt_{fwd,inv}15_m1t); ckks: the first 16 moduli of DefaultParams[PN15QP880] (dual kernel lr_ntt_{fwd,inv}15_m3t: FP64
body on the limbs below 2^46, the 50-bit limb on the integer body; the summary separates the two) -- same instruction stream plus one
s_memtime per phase boundary and wave (asmgen/gen_ntt.py, gen_intt.py: profile=True) and summarises the stamps: per phase the median
over all waves of all workgroups, in shader clocks, next to the launch's wall time.  The stamps cost a few percent (each drains the
wave's LDS / scalar-memory counter): read the split, not the absolute sum.
"""
import json
import os
import sys

os.environ["LR_NTT_TIMELINE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lattigo-fhe-by-go_amd", "csrc", "asmgen"))

import numpy as np  # noqa: E402

import __graft_entry__ as g  # noqa: E402

FWD_PHASES = ["column loads + stage 0 (first loads return .. all returned)", "pass A, stages 1..4 (registers, scalar twiddles)"] + [
    "%s, half %d" % (n, h) for h in range(2) for n in ("column exchange through LDS (2 barriers)", "stages over bits 9..7",
                                                        "stages over bits 6..4", "last four stages", "copy-out: canonical reduction + stores issued")]
INV_PHASES = ["%s, half %d" % (n, h) for h in range(2) for n in (
    "copy-in: 16-byte loads returned, LDS image written", "stages over bits 0..3", "stages over bits 4..6", "stages over bits 7..9",
    "column exchange through LDS (barriers)")] + ["top five stages + fused last stage (registers, scalar twiddles)", "column stores issued"]

# static VALU instruction counts per wave (tests/asm_emulate.py: executed on the emulator; SQ_INSTS_VALU / SQ_WAVES agrees)
VALU_PER_WAVE = {("fwd", "int1"): 4370, ("fwd", "fp"): 2467, ("fwd", "int2"): 3922, ("inv", "int1"): 4899, ("inv", "fp"): 2548}


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    mset = sys.argv[2] if len(sys.argv) > 2 else "qi60"
    out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "timeline_%s15_%s.json" % (kind, mset))
    inverse = kind == "inv"
    pkg = g.load_package()
    ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
    N, moduli = params.DefaultParamsQi(15)
    if mset == "ckks":
        moduli = list(params.ckks_moduli("PN15QP880")[1][:16])
    L, B = len(moduli), 256
    ctx = ring.NewContextWithParams(N, moduli)
    base = sampling.uniform_poly(moduli, N, 2, seed=1)
    src, dst = ctx.NewPoly(B).set(np.concatenate([base] * (B // 2))), ctx.NewPoly(B)
    fn = (lambda: ctx.InvNTT(src, dst)) if inverse else (lambda: ctx.NTT(src, dst))
    for _ in range(20):
        fn()
    ctx.Sync()
    ctx.TimerStart()
    for _ in range(10):
        fn()
    ms = ctx.TimerStop() / 10
    kname = ctx.last_ntt_kernel()
    assert kname.endswith("t") and (kname.startswith("lr_ntt_%s15_m" % kind) or kname.startswith("lr_ntt_%s15p_m" % kind)), kname
    # the stamped build computes the same transform as the shipped kernel (a second context created without the switch)
    os.environ.pop("LR_NTT_TIMELINE")
    os.environ["LR_NTT_SPLIT15"] = "0"          # (a launch of two polys would otherwise run on the 2^14 sub-block kernels)
    plain = ring.NewContextWithParams(N, moduli)
    ref = plain.NewPoly(2)
    (plain.InvNTT if inverse else plain.NTT)(plain.NewPoly(2).set(base), ref)
    assert plain.last_ntt_kernel() == kname[:-1], plain.last_ntt_kernel()
    assert np.array_equal(dst.get().reshape(B, L, N)[:2], ref.get().reshape(2, L, N))
    names = INV_PHASES if inverse else FWD_PHASES
    ns = len(names) + 1
    raw = ctx.timeline().astype(np.int64)                           # [workgroup = poly * L + limb, wave, 16 stamps]
    if os.environ.get("LR_TIMELINE_RAW"):
        np.savez_compressed(os.environ["LR_TIMELINE_RAW"], stamps=raw.astype(np.uint32), moduli=np.array(moduli, dtype=np.uint64), launch_ms=ms)
    st_all = raw[:, :, :ns]
    limb_of = np.arange(st_all.shape[0]) % L
    fp_limb = np.array([int(q) < (1 << 46) for q in moduli])
    groups = {"all": np.ones(L, dtype=bool)}
    if kname.endswith("m3t") and fp_limb.any() and not fp_limb.all():
        groups = {"fp64 body (limbs below 2^46)": fp_limb, "integer body (the others)": ~fp_limb}
    elif kname.endswith("m3t"):
        groups = {"fp64 body (limbs below 2^46)": fp_limb}
    persist = int(os.environ.get("LR_NTT_PERSIST", "0") or 0) if "15p_" in kname else 0
    if persist:
        # persistent kernels: one stamp row per workgroup = chunk of polys (the stamps of its last poly), [chunk * L + limb]
        st_all = st_all[:(-(-B // persist)) * L]
        limb_of = np.arange(st_all.shape[0]) % L
    res = {"kernel": "%s (stamped build of %s), %d workgroups x 16 waves" % (kname, kname[:-1], st_all.shape[0]),
           "polys_per_workgroup": persist or 1,
           "moduli_bits": [int(q).bit_length() for q in moduli],
           "launch_ms": ms, "launch_ms_note": "with stamps; the shipped kernel's time is the bench line's",
           "clock_note": "s_memtime ticks = shader clocks; 256 CUs x 16 workgroups each; launch_ms x clock / 16 = clocks per workgroup slot",
           "bodies": {}}
    # the clock the waves ran at: shader-clock ticks (stamp 13 at the start of the kernel .. the last phase stamp) over the ticks of the
    # constant 100 MHz real-time counter between the same two points (stamps 14 and 15)
    raw = raw[:st_all.shape[0]]
    dt_shader = (raw[:, :, ns - 1] - raw[:, :, 13]) & 0xFFFFFFFF
    dt_real = (raw[:, :, 15] - raw[:, :, 14]) & 0xFFFFFFFF
    good = dt_real > 0
    res["in_kernel_clock_GHz_median"] = float(np.median(dt_shader[good] / dt_real[good])) * 0.1
    res["in_kernel_clock_note"] = "s_memtime delta / s_memrealtime delta x 100 MHz per wave over its whole life (prologue included), median over all waves"
    wg_life_all = ((st_all[:, :, ns - 1].max(axis=1) - st_all[:, :, 0].min(axis=1)) & 0xFFFFFFFF)
    res["effective_shader_clock_GHz"] = float(wg_life_all.sum()) / 256.0 / (ms * 1e-3) / 1e9
    res["effective_shader_clock_note"] = ("sum of the workgroups' lives in shader clocks / 256 CUs / launch wall time: a lower bound "
                                          "(dispatch gaps between workgroups are not counted); nominal 2.4 GHz")
    for gname, mask in groups.items():
        st = st_all[mask[limb_of]]
        d = (np.diff(st, axis=2)) & 0xFFFFFFFF                       # low-word differences, wrap-safe
        life = (st[:, :, ns - 1] - st[:, :, 0]) & 0xFFFFFFFF
        wg_life = ((st[:, :, ns - 1].max(axis=1) - st[:, :, 0].min(axis=1)) & 0xFFFFFFFF)
        # skew inside a workgroup: the first wave's end to the last wave's end (what the successor workgroup waits for), and the
        # spread of the waves' starts
        end_skew = (st[:, :, ns - 1].max(axis=1) - st[:, :, ns - 1].min(axis=1)) & 0xFFFFFFFF
        start_skew = (st[:, :, 0].max(axis=1) - st[:, :, 0].min(axis=1)) & 0xFFFFFFFF
        total = float(np.median(life))
        phases = []
        for i, name in enumerate(names):
            med = float(np.median(d[:, :, i]))
            phases.append({"phase": name, "clocks_median": med, "p10": float(np.percentile(d[:, :, i], 10)),
                           "p90": float(np.percentile(d[:, :, i], 90)), "share": med / total})
        body = "fp" if gname.startswith("fp64") else ("int2" if kname.endswith("m3t") and not inverse else "int1")
        valu = VALU_PER_WAVE[(kind, body)]
        grp = lambda keys: float(sum(p["clocks_median"] for p in phases if any(k in p["phase"] for k in keys)))
        if inverse:
            summary = {"copy-in (loads exposed at the start of each half)": grp(["copy-in"]),
                       "LDS-phase stages": grp(["stages over"]), "column exchanges": grp(["column exchange"]),
                       "top stages (registers)": grp(["top five"]), "column stores issued": grp(["column stores"])}
        else:
            summary = {"loads (exposed: nothing else runs in the workgroup)": phases[0]["clocks_median"], "pass A": phases[1]["clocks_median"],
                       "column exchanges": grp(["column exchange"]), "LDS-phase stages": grp(["stages over", "last four"]),
                       "copy-out (stores issued)": grp(["copy-out"])}
        res["bodies"][gname] = {
            "workgroups": int(st.shape[0]), "wave_life_clocks_median": total, "workgroup_life_clocks_median": float(np.median(wg_life)),
            "wave_start_skew_in_workgroup_median": float(np.median(start_skew)), "wave_end_skew_in_workgroup_median": float(np.median(end_skew)),
            "valu_instructions_per_wave": valu, "waves_per_simd": 4,
            "valu_issue_clocks_at_4_per_instruction": valu * 4 * 4,
            "valu_issue_share_of_workgroup_life_at_4_clocks": valu * 4 * 4 / float(np.median(wg_life)),
            "summary_clocks": summary, "phases": phases}
        print(gname, json.dumps(summary), "wave life", total, "wg life", float(np.median(wg_life)))
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    print("launch ms", ms, "clock GHz >=", res["effective_shader_clock_GHz"], "in-kernel clock GHz", res["in_kernel_clock_GHz_median"])


if __name__ == "__main__":
    main()

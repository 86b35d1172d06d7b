#!/usr/bin/env python3
"""Per-phase timeline of the headline kernel (forward NTT, N = 2^15, 60-bit limbs): where a workgroup's life goes.

    LR_NTT_TIMELINE=1 python tools/timeline.py [out.json]

Runs ring.DefaultParamsQi[15] x 256 polys (the bench shape) on the stamped build of lr_ntt_fwd15_m1 (same instruction stream
plus one s_memtime per phase boundary and wave; asmgen/gen_ntt.py, profile=True) and summarises the stamps: per phase the
median over all waves of all workgroups, in shader clocks, next to the launch's wall time.  The stamps cost a few percent
(each drains the wave's LDS / scalar-memory counter): read the split, not the absolute sum.
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

PHASES = ["column loads + stage 0 (first loads return .. all returned)", "pass A, stages 1..4 (registers, scalar twiddles)"] + [
    "%s, half %d" % (n, h) for h in range(2) for n in ("column exchange through LDS (2 barriers)", "stages over bits 9..7",
                                                        "stages over bits 6..4", "last four stages", "copy-out: canonical reduction + stores issued")]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "timeline_fwd15.json")
    pkg = g.load_package()
    ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
    N, moduli = params.DefaultParamsQi(15)
    L, B = len(moduli), 256
    ctx = ring.NewContextWithParams(N, moduli)
    base = sampling.uniform_poly(moduli, N, 2, seed=1)
    src, dst = ctx.NewPoly(B).set(np.concatenate([base] * (B // 2))), ctx.NewPoly(B)
    for _ in range(20):
        ctx.NTT(src, dst)
    ctx.Sync()
    ctx.TimerStart()
    for _ in range(10):
        ctx.NTT(src, dst)
    ms = ctx.TimerStop() / 10
    assert ctx.last_ntt_kernel() == "lr_ntt_fwd15_m1t", ctx.last_ntt_kernel()
    # the stamped build computes the same transform as the shipped kernel (a second context created without the switch)
    os.environ.pop("LR_NTT_TIMELINE")
    plain = ring.NewContextWithParams(N, moduli)
    ref = plain.NewPoly(2)
    plain.NTT(plain.NewPoly(2).set(base), ref)
    assert plain.last_ntt_kernel() == "lr_ntt_fwd15_m1"
    assert np.array_equal(dst.get().reshape(B, L, N)[:2], ref.get().reshape(2, L, N))
    st = ctx.timeline().astype(np.int64)[:, :, :13]                 # [workgroup, wave, stamp]
    d = (np.diff(st, axis=2)) & 0xFFFFFFFF                           # low-word differences, wrap-safe
    life = (st[:, :, 12] - st[:, :, 0]) & 0xFFFFFFFF
    wg_life = ((st[:, :, 12].max(axis=1) - st[:, :, 0].min(axis=1)) & 0xFFFFFFFF)
    # effective shader clock: per CU the 16 workgroups of the launch run back to back, so the sum of their lives (first
    # stamp to last stamp, in shader clocks) over the launch's wall time is the clock the CU ran at (minus dispatch gaps)
    clocks_per_cu = float(wg_life.sum()) / 256.0
    eff_ghz = clocks_per_cu / (ms * 1e-3) / 1e9
    phases = []
    total = float(np.median(life))
    for i, name in enumerate(PHASES):
        med = float(np.median(d[:, :, i]))
        phases.append({"phase": name, "clocks_median": med, "p10": float(np.percentile(d[:, :, i], 10)),
                       "p90": float(np.percentile(d[:, :, i], 90)), "share": med / total})
    group = lambda keys: float(sum(p["clocks_median"] for p in phases if any(k in p["phase"] for k in keys)))
    res = {
        "kernel": "lr_ntt_fwd15_m1t (stamped build of lr_ntt_fwd15_m1), %d workgroups x 16 waves" % st.shape[0],
        "launch_ms": ms, "launch_ms_note": "with stamps; the shipped kernel's time is the bench line's kernel_ms",
        "wave_life_clocks_median": total, "workgroup_life_clocks_median": float(np.median(wg_life)),
        "effective_shader_clock_GHz": eff_ghz,
        "effective_shader_clock_note": "sum of the workgroups' lives in shader clocks / 256 CUs / launch wall time: a lower bound "
                                       "(dispatch gaps between workgroups are not counted); nominal 2.4 GHz",
        "valu_instructions_per_wave": 4338, "waves_per_simd": 4,
        "valu_issue_share_of_workgroup_life": 4338 * 4 * 4.3 / float(np.median(wg_life)),
        "valu_note": "4338 VALU instructions per wave x 4 waves per SIMD x ~4.3 clocks per instruction (tools/asm_ubench, "
                     "interleaved butterflies) against the workgroup's life: the kernel is bound by vector-instruction issue at the clock "
                     "the chip sustains under this load, not by the exposed memory phases",
        "summary_clocks": {"loads (exposed: nothing else runs in the workgroup)": phases[0]["clocks_median"],
                           "pass A": phases[1]["clocks_median"],
                           "column exchanges": group(["column exchange"]),
                           "LDS-phase stages": group(["stages over", "last four"]),
                           "copy-out (stores issued)": group(["copy-out"])},
        "phases": phases,
        "clock_note": "s_memtime ticks = shader clocks; 256 CUs x 16 workgroups each; launch_ms x clock / 16 = clocks per workgroup slot",
    }
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res["summary_clocks"]), "wave life", total, "launch ms", ms)


if __name__ == "__main__":
    main()

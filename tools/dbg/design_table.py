"""Regenerates the table of DESIGN.md section 6.2 (every benchmarked pipeline entry point) from a bench line:
    python tools/dbg/design_table.py profiles/r04/bench.json          (rewrites the text between the legs-table markers of DESIGN.md)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROWS = [("moddown_ntt", "ModDownNTT (`ModDownSplitedNTTPQ`, in place), R15 16 + 16 limbs", "`ring/ring_benchmark_test.go:341`"),
        ("moddown", "ModDown (`ModDownSplitedPQ`, in place), R15", "`:335`"),
        ("div_floor_ntt", "DivFloorByLastModulusNTT, R15", "`:365`"),
        ("div_floor", "DivFloorByLastModulus, R15", "`:354`"),
        ("div_round", "DivRoundByLastModulus, R15", "`:376`"),
        ("ew_mform", "MForm, R15 (in place, like every row of this family)", "`:140`"),
        ("ew_inv_mform", "InvMForm", "`:146`"),
        ("ew_mulcoeffs_barrett", "MulCoeffs (Barrett)", "`:197`"),
        ("ew_mulcoeffs_barrett_constant", "MulCoeffsConstant", "`:203`"),
        ("ew_mulcoeffs_montgomery_constant", "MulCoeffsMontgomeryConstant (MulCoeffsMontgomery itself: §6.1)", "`:215`"),
        ("ew_add", "Add", "`:231`"),
        ("ew_add_nomod", "AddNoMod", "`:237`"),
        ("ew_sub", "Sub", "`:253`"),
        ("ew_sub_nomod", "SubNoMod", "`:259`"),
        ("ew_neg", "Neg", "`:274`"),
        ("ew_mulscalar", "MulScalar (uint64)", "`:296`"),
        ("ew_mulscalar_bigint", "MulScalarBigint", "`:302`"),
        ("marshal", "Poly.MarshalBinary, one 4.2 MB R15 poly per call (host bytes out: PCIe-inclusive)", "`:52`"),
        ("ckks_rescale", "CKKS Rescale (both components), PN15QP880", "`ckks/ckks_benchmarks_test.go:152`"),
        ("ckks_mul", "CKKS Mul (no key, degree-2 result)", "`:166`"),
        ("ckks_square", "CKKS Square (no key)", "`:172`"),
        ("ckks_add", "CKKS Add (both components, in place)", "`:134`"),
        ("ckks_add_const", "CKKS AddScalar (`AddConst`, first component)", "`:140`"),
        ("ckks_mult_by_const", "CKKS MulScalar (`MultByConst`, both components)", "`:146`"),
        ("ckks_relinearize", "CKKS Relin (switchKeysInPlace + 2 AddLvl)", "`:178`"),
        ("ckks_rotate", "CKKS Rotate (RotateColumns by 1)", "`:190`"),
        ("ckks_conjugate", "CKKS Conjugate", "`:184`"),
        ("ckks_rotate_hoisted", "CKKS RotateHoisted, 8 rotations of each of 32 ciphertexts", "`:199-240`"),
        ("ckks_encrypt_pk", "CKKS Encrypt (pk; everything after the sampling)", "`:79`"),
        ("ckks_decrypt", "CKKS Decrypt (degree 1)", "`:103`"),
        ("marshal_ingest", "Poly.UnmarshalBinary, one 4.7 MB component per call (host bytes in: PCIe-inclusive)", "`ring/ring_object.go:252`"),
        ("bfv_mul", "**BFV Mul, PN14QP438** (BASELINE config 4)", "`bfv/bfv_benchmark_test.go:133`"),
        ("bfv_square", "BFV Square (operand lifted once)", "`:139`"),
        ("bfv_add", "BFV Add (both components, in place)", "`:121`"),
        ("bfv_mulscalar", "BFV MulScalar (both components)", "`:127`"),
        ("bfv_relinearize", "BFV Relin", "`:145`"),
        ("bfv_rotate_rows", "BFV RotateRows", "`:151`"),
        ("bfv_rotate_columns", "BFV RotateCols by 1", "`:157`"),
        ("simple_scaler", "SimpleScaler.Scale, PN14QP438 → t", "`ring/ring_scaling.go:275`")]


def table(d):
    P = d["pipelines"]
    out = ["| entry point | reference benchmark | GPU, one call over the resident batch | **fraction of the 8 TB/s HBM roofline** on the algorithmic bytes; fabric traffic ÷ algorithmic; "
           "vector issue at the sustained clock; nearer roof | CPU oracle, 16 threads / 1 thread (units/s) | heaviest kernels of the call (µs under the profiler) |", "|---|---|---|---|---|---|"]
    for name, title, ref in ROWS:
        o = d["bfv_mul"] if name == "bfv_mul" else P[name]
        r, c = o["roofline"], o["cpu_baseline"]
        ks = ", ".join("`%s` ×%d %.0f" % (x["kernel"].replace("void lr::", "").replace("lr::", "").split("(")[0][:36], x["launches"], x["us_under_the_profiler"]) for x in r.get("kernels", [])[:3])
        rate = ("%.1f k" % (o["value"] / 1e3)) if o["value"] < 1e6 else ("%.2f M" % (o["value"] / 1e6))
        out.append("| %s | %s | %.3f ms per %d: %s %s | **%.3f**; %.2f ×; %.2f; %s | %.0f / %.0f | %s |" % (
            title, ref, o["ms_per_batch"], r["units_per_call"], rate, o["unit"], r["frac"], r["traffic_source"]["ratio_to_algorithmic"], r["valu"]["issue_frac_sustained"],
            r["bound"], c["value"], c["one_thread_value"], ks))
    return "\n".join(out)


if __name__ == "__main__":
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    b, e = "<!-- legs-table:begin -->", "<!-- legs-table:end -->"
    i0, i1 = s.index(b) + len(b), s.index(e)
    open(path, "w").write(s[:i0] + "\n" + table(d) + "\n" + s[i1:])
    print("DESIGN.md section 6.2 rewritten from", sys.argv[1])

#!/usr/bin/env python3
"""Checklist of the reference `ring` package's exported identifiers against the Go shim in go/ring.

    python tools/go_surface.py [--reference /root/reference] [--check]

Reads the exported functions, methods, types and variables of every non-test file of <reference>/ring (names only: the
equivalent of `grep '^func ' ring/*.go`), looks each one up in go/ring/*.go, and writes go/ring/SURFACE.md.  --check exits
non-zero when an identifier of a REPLACED file is neither defined in the shim nor listed in OMITTED below, or when SURFACE.md
is stale.  The reference tree exists only in the build container; the committed SURFACE.md is the artefact a reviewer reads.
"""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the shim replaces these files of the reference package; the others stay as they are upstream (go/ring/cgo.go, file map)
REPLACED = ["modular_reduction.go", "ntt.go", "ring.go", "ring_basis_extension.go", "ring_context.go", "ring_galois.go",
            "ring_object.go", "ring_scaling.go"]
KEPT = {
    "int.go": "big.Int helpers; no Context involved",
    "utils.go": "ModExp, IsPrime, GenerateNTTPrimes, PowerOf2: host number theory (the library has its own C++ restatement for its tables)",
    "params.go": "modulus tables Qi60 / Pi60 / DefaultParams",
    "prng.go": "CRPGenerator (keyed PRNG for common reference polynomials); reads Context.N / Modulus / mask through UniformPoly-style loops",
    "float128.go": "double-double arithmetic used by the reference's own SimpleScaler; the shim's SimpleScaler is on the device",
    "sampler.go": "uniform sampler; reads Context.mask",
    "gaussianSampler.go": "KYSampler / Gaussian samplers; write Coeffs on the host",
    "ternarySampler.go": "ternary samplers; read Context.matrixTernary / matrixTernaryMontgomery, call Context.NTT",
}
# exported identifiers of replaced files that the shim deliberately does not provide
OMITTED = {
    "Context.NTTBarrett": "benchmark-only (\"For benchmark purposes only\", ring/ntt.go:141; no caller in bfv/ ckks/ dbfv/ dckks/): Barrett products with the Montgomery-form table, so its outputs are not a transform any caller could use",
    "Context.InvNTTBarrett": "benchmark-only alternative (ring/ntt.go:141)",
    "ButterflyBarrett": "benchmark-only (ring/ntt.go:155)",
    "InvButterflyBarrett": "benchmark-only",
    "NTTBarrett": "benchmark-only",
    "InvNTTBarrett": "benchmark-only",
}
# where a covered identifier does its work
HOST = {"MForm", "MFormConstant", "InvMForm", "InvMFormConstant", "MRedParams", "MRed", "MRedConstant", "BRedParams", "BRedAdd",
        "BRedAddConstant", "BRed", "BRedConstant", "CRed", "Butterfly", "InvButterfly", "GenGaloisParams", "PermuteNTT",
        "PermuteNTTWithIndex", "NewPoly", "NewPolyUniform", "NewContext", "NewDecomposer", "Decomposer.Xalpha", "WriteCoeffsTo",
        "DecodeCoeffs", "DecodeCoeffsNew", "Context.SetParameters", "Context.MarshalBinary", "Context.UnmarshalBinary",
        "Context.AllowsNTT", "Context.GetBredParams", "Context.GetMredParams", "Context.GetPsi", "Context.GetPsiInv",
        "Context.GetNttPsi", "Context.GetNttPsiInv", "Context.GetNttNInv", "Context.NewPoly", "Context.NewPolyLvl",
        "Context.SetCoefficientsInt64", "Context.SetCoefficientsUint64", "Context.SetCoefficientsString",
        "Context.SetCoefficientsBigint", "Context.SetCoefficientsBigintLvl", "Context.PolyToString", "Context.PolyToBigint",
        "Context.Mod", "Context.AND", "Context.OR", "Context.XOR", "Context.MulPolyNaive", "Context.MulPolyNaiveMontgomery",
        "Context.MulByVectorMontgomery", "Context.MulByVectorMontgomeryAndAddNoMod", "Context.BitReverse", "Poly.GetDegree",
        "Poly.GetLenModuli", "Poly.CopyNew", "Poly.Copy", "Poly.SetCoefficients", "Poly.GetCoefficients", "Poly.WriteTo",
        "Poly.WriteCoeffs", "Poly.GetDataLen", "Poly.DecodePolyNew"}

DECL = re.compile(r"^func (?:\((?:[A-Za-z_]\w*) \*?([A-Za-z_]\w*)\) )?([A-Z]\w*)\s*\(|^type ([A-Z]\w*)\b|^(?:var|const) ([A-Z]\w*)\b", re.M)


def exported(path):
    out = []
    for m in DECL.finditer(open(path).read()):
        recv, fn, typ, var = m.groups()
        if fn:
            out.append((recv + "." if recv else "") + fn)
        elif typ:
            out.append("type " + typ)
        elif var:
            out.append("var " + var)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    ref = os.path.join(args.reference, "ring")
    if not os.path.isdir(ref):
        print("reference tree not present; nothing to do")
        return 0
    shim_dir = os.path.join(ROOT, "go", "ring")
    shim = {}
    for f in sorted(os.listdir(shim_dir)):
        if f.endswith(".go"):
            for name in exported(os.path.join(shim_dir, f)):
                shim.setdefault(name, f)
    lines = ["# go/ring against the reference `ring` package (v1.3.1)", "",
             "Generated by `tools/go_surface.py` from the names `grep '^func \\|^type \\|^var ' ring/*.go` yields in the reference tree.",
             "`device`: forwarded to the C ABI; `host`: Go code inside the shim (no kernel involved); `omitted`: not provided, with",
             "the reason; `upstream file`: the identifier lives in a file the shim does not replace (see go/ring/cgo.go).", ""]
    missing = []
    counts = {"device": 0, "host": 0, "omitted": 0, "upstream file": 0}
    for f in sorted(os.listdir(ref)):
        if not f.endswith(".go") or f.endswith("_test.go"):
            continue
        names = exported(os.path.join(ref, f))
        if f in KEPT:
            lines += ["## %s -- kept from upstream (%s)" % (f, KEPT[f]), "", ", ".join("`%s`" % n for n in names), ""]
            counts["upstream file"] += len(names)
            continue
        assert f in REPLACED, f
        lines += ["## %s -- replaced" % f, "", "| identifier | status | where |", "|---|---|---|"]
        for n in names:
            if n in shim:
                kind = "host" if n in HOST else "device"
                if n.startswith("type ") or n.startswith("var "):
                    kind = "host"
                counts[kind] += 1
                lines.append("| `%s` | %s | go/ring/%s |" % (n, kind, shim[n]))
            elif n in OMITTED:
                counts["omitted"] += 1
                lines.append("| `%s` | omitted | %s |" % (n, OMITTED[n]))
            else:
                missing.append(f + ": " + n)
                lines.append("| `%s` | **MISSING** | |" % n)
        lines.append("")
    extra = sorted(n for n in shim if not any(n in exported(os.path.join(ref, f)) for f in os.listdir(ref) if f.endswith(".go") and not f.endswith("_test.go")))
    lines += ["## additions of the shim (no counterpart in the reference)", "", ", ".join("`%s` (%s)" % (n, shim[n]) for n in extra), "",
              "Totals: %d device, %d host, %d omitted, %d in upstream files; missing: %d." %
              (counts["device"], counts["host"], counts["omitted"], counts["upstream file"], len(missing)), ""]
    text = "\n".join(lines)
    target = os.path.join(shim_dir, "SURFACE.md")
    if args.check:
        stale = not os.path.exists(target) or open(target).read() != text
        for m in missing:
            print("MISSING", m)
        if stale:
            print("SURFACE.md is stale: run tools/go_surface.py")
        return 1 if (missing or stale) else 0
    open(target, "w").write(text)
    print("wrote %s; missing: %s" % (target, missing))
    return 0


if __name__ == "__main__":
    sys.exit(main())

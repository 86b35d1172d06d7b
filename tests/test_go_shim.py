"""The Go shim (go/ring) cannot be compiled here (no Go toolchain in the image).  These checks keep it honest without a
compiler: every C symbol it calls is declared in include/lattigo_ring.h with the same number of arguments, its braces and
parentheses balance, and the exported-identifier checklist (go/ring/SURFACE.md) is complete and current."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

GO_DIR = os.path.join(ROOT, "go", "ring")
OVERLAYS = {"ckks": os.path.join(ROOT, "go", "ckks", "evaluator_device.go"), "bfv": os.path.join(ROOT, "go", "bfv", "evaluator_device.go")}


def _go_sources():
    return {f: open(os.path.join(GO_DIR, f)).read() for f in sorted(os.listdir(GO_DIR)) if f.endswith(".go")}


def _exported_methods(text, receiver):
    """names of the methods declared on `receiver` (pointer or value) in Go source text"""
    return set(re.findall(r"func \(\w+ \*?%s\) (\w+)\(" % receiver, text))


def _strip(text):
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)
    text = re.sub(r"'(?:\\.|[^'\\])'", "''", text)
    return re.sub(r"`[^`]*`", "``", text)


def _header_arity():
    text = open(os.path.join(ROOT, "include", "lattigo_ring.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(lr_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def _split_args(s):
    depth, cur, out = 0, "", []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def test_delimiters_balance():
    for name, text in _go_sources().items():
        t = _strip(text)
        for a, b in ("{}", "()", "[]"):
            assert t.count(a) == t.count(b), (name, a, t.count(a), t.count(b))
        assert t.lstrip().startswith("package ring"), name


def test_every_c_call_matches_the_header():
    arity = _header_arity()
    seen = set()
    for name, text in _go_sources().items():
        t = _strip(text)
        for m in re.finditer(r"\bC\.(lr_[a-z0-9_]+)\s*\(", t):
            sym = m.group(1)
            assert sym in arity, (name, sym, "not declared in include/lattigo_ring.h")
            # the call's argument list
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(t[i], 0)
                i += 1
            nargs = len(_split_args(t[m.end():i - 1]))
            assert nargs == arity[sym], (name, sym, nargs, arity[sym])
            seen.add(sym)
    assert len(seen) >= 55            # the shim reaches (nearly) the whole ABI
    unused = sorted(set(arity) - seen - {"lr_build_info", "lr_device_count", "lr_timer_start", "lr_timer_stop", "lr_context_info",
                                         "lr_poly_info", "lr_poly_wrap", "lr_poly_wrap_strided", "lr_poly_upload_dense",
                                         "lr_poly_download_dense", "lr_context_ntt_variants", "lr_context_last_ntt_kernel",
                                         "lr_ntt_limb", "lr_intt_limb", "lr_bext_get_table", "lr_simple_scaler_tables", "lr_context_timeline", "lr_selftest_division",
                                         # the pointer-array forms serve C / C++ / Python callers; a go 1.13 cgo caller may not store Go pointers in a C
                                         # array and uses the per-limb forms (lr_poly_upload_limb, lr_poly_download_limb, lr_ntt_host_limb)
                                         "lr_poly_upload", "lr_poly_download", "lr_ntt_host", "lr_intt_host",
                                         # the shim creates its handles through the *_create_ex forms (DefaultOptions; nil = the defaults, which is what these do)
                                         "lr_context_create", "lr_ckks_plan_create", "lr_bfv_plan_create",
                                         # a Go Poly is one polynomial (its own lr_poly): the shim's GatherTo is lr_poly_copy_peer per poly +
                                         # lr_context_wait_peer_copies; the one-call block form serves batched callers (C++, Python)
                                         "lr_gather_blocks"})
    assert not unused, unused


def test_status_constants_exist_in_header():
    text = open(os.path.join(ROOT, "include", "lattigo_ring.h")).read()
    for name, src in _go_sources().items():
        for m in re.finditer(r"\bC\.(LR_[A-Z0-9_]+)\b", src):
            assert re.search(r"\b%s\b" % m.group(1), text), (name, m.group(1))


@pytest.mark.skipif(not os.path.isdir("/root/reference/ring"), reason="the reference tree exists only in the build container")
def test_surface_checklist_is_complete_and_current():
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "go_surface.py"), "--check"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert rc.returncode == 0, rc.stdout.decode()
    text = open(os.path.join(GO_DIR, "SURFACE.md")).read()
    assert "MISSING" not in text
    # the package-level identifiers the evaluators call (ckks/evaluator.go, bfv/evaluator.go)
    for ident in ("MRed", "CRed", "MForm", "NTT", "InvNTT", "PermuteNTTWithIndex", "GenGaloisParams", "NewPoly", "NewContext",
                  "Context.SetParameters", "Context.GenNTTParams", "NewDecomposer", "NewFastBasisExtender", "BRedParams", "MRedParams"):
        assert "| `%s` |" % ident in text, ident


def test_go_1_13_language_level():
    """the reference module declares go 1.13 (/go.mod:3): no runtime.Pinner (1.21), no unsafe.Slice (1.17), no generics, no `any`"""
    texts = dict(_go_sources())
    for name, path in OVERLAYS.items():
        texts[name + "/evaluator_device.go"] = open(path).read()
    for name, text in texts.items():
        t = _strip(text)
        assert "runtime.Pinner" not in t and "unsafe.Slice" not in t and "unsafe.String" not in t, name
        assert not re.search(r"func \w+\[", t), (name, "type parameters")
        assert not re.search(r"\bany\b", t), name


def test_evaluator_overlays_use_only_what_the_shim_exports():
    """go/ckks/evaluator_device.go and go/bfv/evaluator_device.go: delimiters balance, they are overlays of the upstream packages, every
    method they call on ring.CkksPlan / ring.BfvPlan / ring.Poly exists in go/ring with the same number of arguments, and the methods
    the review listed are re-pointed"""
    ring_text = "\n".join(_go_sources().values())
    plan_methods = {"CkksPlan": _exported_methods(ring_text, "CkksPlan"), "BfvPlan": _exported_methods(ring_text, "BfvPlan"), "Poly": _exported_methods(ring_text, "Poly")}
    assert {"MulRelin", "Rescale", "SwitchKeysInPlace", "PermuteNTT", "RotateHoisted", "SwitchingKeyImage", "BfvSwitchKeys", "BfvRelinearize"} <= plan_methods["CkksPlan"]
    assert {"Pin", "HostView", "HostWritten", "Sync", "Unpin"} <= plan_methods["Poly"]

    def arity_of(receiver, method):
        m = re.search(r"func \(\w+ \*?%s\) %s\(([^)]*)\)" % (receiver, method), ring_text)
        assert m, (receiver, method)
        groups = [g for g in _split_args(m.group(1)) if g.strip()]
        n = 0
        pending = 0
        for g in groups:                       # "a, b *Poly" declares two parameters: names without a type wait for the next typed group
            parts = g.strip().split()
            pending += 1
            if len(parts) >= 2:
                n += pending
                pending = 0
        return n + pending

    for pkg, path in OVERLAYS.items():
        text = open(path).read()
        t = _strip(text)
        for a, b in ("{}", "()", "[]"):
            assert t.count(a) == t.count(b), (pkg, a)
        assert t.lstrip().startswith("package " + pkg)
        assert '"github.com/ldsec/lattigo/ring"' in text
        for field, receiver in (("plan", "CkksPlan"), ("ks", "CkksPlan"), ("mul", "BfvPlan")):
            calls = list(re.finditer(r"\.%s\.(\w+)\(" % field, t))
            for m in calls:
                meth = m.group(1)
                assert meth in plan_methods[receiver], (pkg, field, meth)
                i, depth = m.end(), 1
                while depth:
                    depth += {"(": 1, ")": -1}.get(t[i], 0)
                    i += 1
                assert len(_split_args(t[m.end():i - 1])) == arity_of(receiver, meth), (pkg, meth)
        for m in re.finditer(r"\bp\.(HostView|HostWritten|Pin)\(", t):
            assert m.group(1) in plan_methods["Poly"]
        # replacement bodies: methods on the upstream receiver type itself (Go has no virtual dispatch; an embedding wrapper would leave
        # upstream callers such as Power / EvaluatePoly / the ...New wrappers on the upstream bodies)
        assert "type deviceEvaluator" not in t and re.search(r"func \(\w+ \*evaluator\) dev\(\)", t)
    ck = open(OVERLAYS["ckks"]).read()
    for meth in ("MulRelin", "Relinearize", "SwitchKeys", "Rescale", "switchKeysInPlace", "permuteNTT", "RotateHoisted"):
        assert re.search(r"func \(eval \*evaluator\) %s\(" % meth, ck), meth
        assert re.search(r"delete\s+%s\b" % meth, ck), (meth, "missing from the patch list in the header")
    for meth in ("AddConst", "MultByConstAndAdd", "MultByConst", "MultByi", "DivByi", "decomposeAndSplitNTT", "switchKeyHoisted"):
        assert meth in ck, meth
    bf = open(OVERLAYS["bfv"]).read()
    for meth in ("Mul", "switchKeys", "relinearize"):
        assert re.search(r"func \(evaluator \*evaluator\) %s\(" % meth, bf), meth
        assert re.search(r"delete\s+%s\b" % meth, bf), meth


@pytest.mark.skipif(not os.path.isdir("/root/reference/ckks"), reason="the reference tree exists only in the build container")
def test_replacement_bodies_keep_the_upstream_signatures():
    """every method the overlays define on *evaluator under an upstream name has the upstream parameter list, so upstream callers
    (polynomial_evaluation.go, the ...New wrappers, rotateColumnsPow2) compile against it unchanged"""
    def sigs(text):
        out = {}
        for m in re.finditer(r"func \(\w+ \*evaluator\) (\w+)\(([^)]*)\)([^{]*)\{", text):
            params = re.sub(r"\s+", " ", m.group(2)).strip()
            types = [re.sub(r"^\w+ ", "", g.strip()) if " " in g.strip() else None for g in _split_args(params)] if params else []
            # names without a type take the type of the next typed parameter
            for i in range(len(types) - 2, -1, -1):
                if types[i] is None:
                    types[i] = types[i + 1]
            out[m.group(1)] = (types, re.sub(r"\s+", " ", m.group(3)).strip())
        return out
    for pkg in ("ckks", "bfv"):
        up = sigs(open(os.path.join("/root/reference", pkg, "evaluator.go")).read())
        mine = sigs(open(OVERLAYS[pkg]).read())
        helpers = {"dev", "keyImage", "resident", "hostLoop", "galoisElement", "halfScalar", "batcher", "ReleaseDevice"}
        for name, (types, ret) in mine.items():
            if name in helpers:
                assert name not in up, (pkg, name, "helper collides with an upstream method")
                continue
            assert name in up, (pkg, name, "not an upstream method")
            assert types == up[name][0], (pkg, name, types, up[name][0])
            strip_names = lambda r: re.sub(r"\b\w+ (\*?\w)", r"\1", r)
            assert strip_names(ret) == strip_names(up[name][1]), (pkg, name, ret, up[name][1])


@pytest.mark.skipif(not os.path.isdir("/root/reference/ckks"), reason="the reference tree exists only in the build container")
def test_overlays_name_upstream_identifiers_that_exist():
    """every unexported upstream identifier the overlays lean on (fields, helpers) exists in the reference's package sources"""
    for pkg, idents in (("ckks", ["getElemAndCheckBinary", "ckksContext", "contextQ", "contextP", "evakeyRotColLeft", "baseconverter",
                                  "permuteNTTLeftIndex", "DropLevel", "logN", "poolQ[", "Resize(", "DivScale("]),
                        ("bfv", ["getElemAndCheckBinary", "bfvContext", "contextQMul", "keyswitchpool", "SetValue(", "evakey.evakey",
                                 "baseconverterQ1P", "tensorAndRescale("])):
        src = "\n".join(open(os.path.join("/root/reference", pkg, f)).read() for f in os.listdir(os.path.join("/root/reference", pkg))
                        if f.endswith(".go") and not f.endswith("_test.go"))
        overlay = open(OVERLAYS[pkg]).read()
        for ident in idents:
            assert ident in overlay, (pkg, ident, "no longer used by the overlay")
            assert ident in src, (pkg, ident, "not in the reference package")

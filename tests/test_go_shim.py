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


def _go_sources():
    return {f: open(os.path.join(GO_DIR, f)).read() for f in sorted(os.listdir(GO_DIR)) if f.endswith(".go")}


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
                                         "lr_ntt_limb", "lr_intt_limb", "lr_bext_get_table", "lr_simple_scaler_tables", "lr_context_timeline", "lr_selftest_division"})
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

"""ctypes front-end for the CPU oracle (oracle/lr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  Parity status:
pinned by the reference's golden NTT vectors (tests/golden/ring_test_data) and the
big-integer identities of ring/ring_test.go (tests/test_oracle_*.py).

Polynomials are numpy uint64 arrays of shape [limbs, N] (limb-major), the dense
image of Go's ``Poly.Coeffs [][]uint64`` (ring/ring_object.go:11-13).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblr_oracle.so")

u64 = C.c_uint64
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    """Compile liblr_oracle.so with gcc (a few seconds)."""
    src = os.path.join(_HERE, "lr_oracle.c")
    hdr = os.path.join(_HERE, "lr_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liblr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _CContext(C.Structure):
    _fields_ = [("N", u64), ("L", C.c_int), ("q", u64p), ("mask", u64p), ("bred", u64p), ("mred", u64p),
                ("rescale", u64p), ("psi_mont", u64p), ("psi_inv_mont", u64p), ("ntt_psi", u64p),
                ("ntt_psi_inv", u64p), ("n_inv", u64p)]


class _CModup(C.Structure):
    _fields_ = [("nQ", C.c_int), ("nP", C.c_int), ("Q", u64p), ("P", u64p), ("qib_mont", u64p),
                ("qispj_mont", u64p), ("qpj_inv", u64p), ("bredQ", u64p), ("bredP", u64p),
                ("mredQ", u64p), ("mredP", u64p)]


class _CBext(C.Structure):
    _fields_ = [("cQ", C.c_void_p), ("cP", C.c_void_p), ("qp", C.POINTER(_CModup)), ("pq", C.POINTER(_CModup)),
                ("moddown_pq", u64p), ("moddown_qp", u64p), ("poolQ", u64p), ("poolP", u64p)]


_lib = None
_NATIVE_DIR = os.path.join(_HERE, "_native")
_NATIVE_PATH = os.path.join(_NATIVE_DIR, "liblr_oracle_native.so")
native_loaded = False


def use_native():
    """Switch this process to a build of the SAME source with ``-O2 -march=native``, compiled on the machine it runs on
    (bench.py's cpu_baseline on the GPU box).  The portable build (oracle/Makefile, no -march) is what
    travels and what the tests use.  Returns True if the native build is in use."""
    global _lib, native_loaded
    if native_loaded:
        return True
    try:
        os.makedirs(_NATIVE_DIR, exist_ok=True)
        src = os.path.join(_HERE, "lr_oracle.c")
        subprocess.check_call(["gcc", "-O2", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
                               "-shared", "-o", _NATIVE_PATH, src, "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _lib = _bind(C.CDLL(_NATIVE_PATH))
        native_loaded = True
    except Exception:
        native_loaded = False
    return native_loaded


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


def _bind(L):
    vp = C.c_void_p
    i = C.c_int

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("oc_mred_params", u64, u64)
    sig("oc_bred_params", None, u64, u64p)
    sig("oc_mform", u64, u64, u64, u64p)
    sig("oc_mform_constant", u64, u64, u64, u64p)
    sig("oc_inv_mform", u64, u64, u64, u64)
    sig("oc_mred", u64, u64, u64, u64, u64)
    sig("oc_mred_constant", u64, u64, u64, u64, u64)
    sig("oc_bred_add", u64, u64, u64, u64p)
    sig("oc_bred_add_constant", u64, u64, u64, u64p)
    sig("oc_bred", u64, u64, u64, u64, u64p)
    sig("oc_bred_constant", u64, u64, u64, u64, u64p)
    sig("oc_cred", u64, u64, u64)
    sig("oc_power_of_2", u64, u64, u64, u64, u64)
    sig("oc_mod_exp", u64, u64, u64, u64)
    sig("oc_is_prime", i, u64)
    sig("oc_primitive_root", u64, u64)
    sig("oc_get_factors", i, u64, u64p, i)
    sig("oc_generate_ntt_primes", i, u64, u64, u64, u64p)
    sig("oc_bit_reverse64", u64, u64, u64)
    sig("oc_context_new", i, u64, u64p, i, C.POINTER(C.POINTER(_CContext)))
    sig("oc_context_free", None, vp)
    sig("oc_ntt_limb", None, vp, vp, u64, vp, u64, u64, u64p)
    sig("oc_intt_limb", None, vp, vp, u64, vp, u64, u64, u64)
    sig("oc_ntt_lvl", None, vp, i, vp, vp)
    sig("oc_intt_lvl", None, vp, i, vp, vp)
    sig("oc_ewise", None, vp, i, i, vp, vp, vp, vp)
    sig("oc_modup_params_new", C.POINTER(_CModup), u64p, i, u64p, i)
    sig("oc_modup_params_free", None, vp)
    sig("oc_modup_exact", None, vp, vp, i, vp, i, u64)
    sig("oc_bext_new", C.POINTER(_CBext), vp, vp)
    sig("oc_bext_free", None, vp)
    sig("oc_modup_split_qp", None, vp, i, vp, vp)
    sig("oc_modup_split_pq", None, vp, i, vp, vp)
    sig("oc_moddown_ntt_pq", None, vp, i, vp, vp)
    sig("oc_moddown_split_ntt_pq", None, vp, i, vp, vp, vp)
    sig("oc_moddown_pq", None, vp, i, vp, vp)
    sig("oc_moddown_split_pq", None, vp, i, vp, vp, vp)
    sig("oc_moddown_split_qp", None, vp, i, i, vp, vp, vp)
    sig("oc_decomposer_new", vp, u64p, i, u64p, i)
    sig("oc_decomposer_free", None, vp)
    sig("oc_decompose", None, vp, i, i, vp, vp, u64)
    sig("oc_decompose_and_split", None, vp, i, i, vp, vp, vp, u64)
    for n in ("oc_div_floor_by_last_modulus_ntt", "oc_div_floor_by_last_modulus",
              "oc_div_round_by_last_modulus_ntt", "oc_div_round_by_last_modulus"):
        sig(n, None, vp, vp, i)
    sig("oc_div_floor_by_last_modulus_many", None, vp, vp, i, i, i)
    sig("oc_div_round_by_last_modulus_many", None, vp, vp, i, i, i)
    sig("oc_ckks_plan_new", vp, vp, vp)
    sig("oc_ckks_plan_free", None, vp)
    sig("oc_ckks_switch_keys", None, vp, i, vp, vp, vp, vp)
    sig("oc_ckks_mulrelin", None, vp, i, vp, vp, vp, vp)
    sig("oc_half_scalar_op", None, vp, i, i, vp, vp, vp, vp)
    sig("oc_bfv_switch_keys", None, vp, vp, vp, vp, vp)
    sig("oc_bfv_relinearize", None, vp, vp, vp, vp)
    sig("oc_bfv_permute", None, vp, vp, C.c_uint64, vp, vp)
    sig("oc_ckks_permute_ntt", None, vp, i, vp, u64, vp, vp)
    sig("oc_ckks_mul_norelin", None, vp, i, vp, vp, i, vp)
    sig("oc_ckks_mul_plain", None, vp, i, vp, vp, vp)
    sig("oc_ckks_encrypt_pk", None, vp, vp, i, vp, vp, vp, vp, vp, vp, vp)
    sig("oc_ckks_decrypt", None, vp, i, vp, i, vp, vp)
    sig("oc_ckks_rotate_hoisted", None, vp, i, vp, i, vp, vp, vp)
    sig("oc_bfv_mul", None, vp, u64, vp, vp, vp, vp, vp)
    sig("oc_bfv_square", None, vp, u64, vp, vp, vp, vp)
    sig("oc_permute_ntt_index", None, u64, u64, u64, vp)
    sig("oc_permute_ntt", None, vp, u64, vp, i, u64)
    sig("oc_permute_ntt_with_index", None, vp, vp, vp, i, u64)
    sig("oc_permute", None, vp, vp, u64, vp)
    sig("oc_mult_by_monomial", None, vp, vp, u64, vp)
    sig("oc_shift", i, vp, vp, u64, vp)
    sig("oc_rotate", None, vp, vp, u64)
    dp = C.POINTER(C.c_double)
    sig("oc_f128_set_uint53", None, u64, dp)
    sig("oc_f128_set_uint64", None, u64, dp)
    sig("oc_f128_to_uint53", u64, dp)
    sig("oc_f128_to_uint64", u64, dp)
    sig("oc_f128_add", None, dp, dp, dp)
    sig("oc_f128_mul", None, dp, dp, dp)
    sig("oc_f128_div", None, dp, dp, dp)
    sig("oc_simple_scaler_new", None, vp, u64, u64p, dp, u64p)
    sig("oc_simple_scale", None, vp, u64, u64p, dp, u64p, vp, vp, i)
    return L


# -- enum oc_ewise_op (keep in sync with lr_oracle.h) --------------------------
EWISE_OPS = ["ADD", "ADD_NOMOD", "SUB", "SUB_NOMOD", "NEG", "REDUCE", "MUL_COEFFS", "MUL_COEFFS_AND_ADD",
             "MUL_COEFFS_AND_ADD_NOMOD", "MUL_COEFFS_CONSTANT", "MUL_MONT", "MUL_MONT_AND_ADD",
             "MUL_MONT_AND_ADD_NOMOD", "MUL_MONT_CONSTANT_AND_ADD_NOMOD", "MUL_MONT_AND_SUB",
             "MUL_MONT_AND_SUB_NOMOD", "MUL_MONT_CONSTANT", "MFORM", "INV_MFORM", "MUL_SCALAR",
             "MUL_SCALAR_LIMBS", "ADD_SCALAR_LIMBS", "SUB_SCALAR_LIMBS", "COPY", "MUL_BY_POW2"]
OP = {name: k for k, name in enumerate(EWISE_OPS)}


def _arr(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u64arr(vals):
    vals = [int(v) for v in vals]
    return (u64 * len(vals))(*vals)


# -- scalar helpers --------------------------------------------------------------
def bred_params(q):
    u = (u64 * 2)()
    lib().oc_bred_params(q, u)
    return [u[0], u[1]]


def mred_params(q):
    return lib().oc_mred_params(q)


def mform(a, q):
    return lib().oc_mform(a, q, _u64arr(bred_params(q)))


def inv_mform(a, q):
    return lib().oc_inv_mform(a, q, mred_params(q))


def mred(x, y, q):
    return lib().oc_mred(x, y, q, mred_params(q))


def bred(x, y, q):
    return lib().oc_bred(x, y, q, _u64arr(bred_params(q)))


def bred_add(x, q):
    return lib().oc_bred_add(x, q, _u64arr(bred_params(q)))


def generate_ntt_primes(logQ, logN, levels):
    out = (u64 * levels)()
    n = lib().oc_generate_ntt_primes(logQ, logN, levels, out)
    assert n == levels
    return [int(v) for v in out]


class Context:
    """ring.Context (ring/ring_context.go:18-51) built by NewContextWithParams (:60)."""

    def __init__(self, N, moduli):
        self.N = int(N)
        self.moduli = [int(m) for m in moduli]
        self.L = len(self.moduli)
        pp = C.POINTER(_CContext)()
        rc = lib().oc_context_new(self.N, _u64arr(self.moduli), self.L, C.byref(pp))
        if rc == 1:
            raise ValueError("warning : provided modulus does not allow NTT")
        if rc == 2:
            raise ValueError("invalid ring degree (must be a power of 2)")
        self._pp = pp
        self.h = C.cast(pp, C.c_void_p)
        c = pp.contents
        L, N_ = self.L, self.N
        self.bred = np.ctypeslib.as_array(c.bred, shape=(L, 2)).copy()
        self.mred = np.ctypeslib.as_array(c.mred, shape=(L,)).copy()
        self.mask = np.ctypeslib.as_array(c.mask, shape=(L,)).copy()
        self.rescale = np.ctypeslib.as_array(c.rescale, shape=(L, L)).copy()
        self.psi_mont = np.ctypeslib.as_array(c.psi_mont, shape=(L,)).copy()
        self.psi_inv_mont = np.ctypeslib.as_array(c.psi_inv_mont, shape=(L,)).copy()
        self.n_inv = np.ctypeslib.as_array(c.n_inv, shape=(L,)).copy()
        self.ntt_psi = np.ctypeslib.as_array(c.ntt_psi, shape=(L, N_)).copy()
        self.ntt_psi_inv = np.ctypeslib.as_array(c.ntt_psi_inv, shape=(L, N_)).copy()

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().oc_context_free(self.h)
                self.h = None
        except Exception:
            pass

    def new_poly(self, limbs=None):
        return np.zeros((self.L if limbs is None else limbs, self.N), dtype=np.uint64)

    # NTTLvl / InvNTTLvl (ring/ntt.go:11,25); level=None -> all limbs (:4,:18)
    def ntt(self, p, level=None):
        p = _arr(p)
        level = p.shape[0] - 1 if level is None else level
        out = p.copy()
        lib().oc_ntt_lvl(self.h, level, _ptr(p), _ptr(out))
        return out

    def intt(self, p, level=None):
        p = _arr(p)
        level = p.shape[0] - 1 if level is None else level
        out = p.copy()
        lib().oc_intt_lvl(self.h, level, _ptr(p), _ptr(out))
        return out

    def ewise(self, op, a, b=None, out=None, scalars=None, level=None):
        """out <- op(a, b[, out]) on limbs 0..level; returns out (a fresh copy unless given)."""
        a = _arr(a)
        level = a.shape[0] - 1 if level is None else level
        b = None if b is None else _arr(b)
        out = np.zeros_like(a) if out is None else _arr(out).copy()
        sc = None if scalars is None else np.ascontiguousarray(scalars, dtype=np.uint64)
        lib().oc_ewise(self.h, OP[op] if isinstance(op, str) else op, level, _ptr(a), _ptr(b), _ptr(out), _ptr(sc))
        return out

    def half_scalar_op(self, op, p, lo, hi, out=None):
        """element loops of ckks.Evaluator's constant methods (ckks/evaluator.go:429-828): op 0 CRed(x+s), 1 MRed(x,s), 2 CRed(out+MRed(x,s))"""
        p = _arr(p)
        res = np.zeros_like(p) if out is None else _arr(out).copy()
        lo_, hi_ = _u64arr(lo), _u64arr(hi)
        lib().oc_half_scalar_op(self.h, op, p.shape[0] - 1, _ptr(p), lo_, hi_, _ptr(res))
        return res

    def permute_ntt(self, p, gen):           # ring.PermuteNTT (ring/ring_galois.go:55)
        p = _arr(p)
        out = np.zeros_like(p)
        lib().oc_permute_ntt(_ptr(p), int(gen), _ptr(out), p.shape[0], self.N)
        return out

    def permute_ntt_index(self, gen, power):  # ring.PermuteNTTIndex (:29)
        idx = np.zeros(self.N, dtype=np.uint64)
        lib().oc_permute_ntt_index(int(gen), int(power), self.N, _ptr(idx))
        return idx

    def permute_ntt_with_index(self, p, index):   # ring.PermuteNTTWithIndex (:89)
        p, index = _arr(p), _arr(index)
        out = np.zeros_like(p)
        lib().oc_permute_ntt_with_index(_ptr(p), _ptr(index), _ptr(out), p.shape[0], self.N)
        return out

    def permute(self, p, gen):               # Context.Permute (:106)
        p = _arr(p)
        out = np.zeros_like(p)
        lib().oc_permute(self.h, _ptr(p), int(gen), _ptr(out))
        return out

    def mult_by_monomial(self, p, deg):      # Context.MultByMonomial (ring/ring.go:663)
        p = _arr(p)
        out = np.zeros_like(p)
        lib().oc_mult_by_monomial(self.h, _ptr(p), int(deg), _ptr(out))
        return out

    def shift(self, p, n):                   # Context.Shift (ring/ring.go:575); None where the reference panics
        p = _arr(p)
        out = np.zeros_like(p)
        return out if lib().oc_shift(self.h, _ptr(p), int(n), _ptr(out)) == 0 else None

    def rotate(self, p, n):                  # Context.Rotate (ring/ring.go:775): the new contents of p1
        out = _arr(p).copy()
        lib().oc_rotate(self.h, _ptr(out), int(n))
        return out

    def rescale_op(self, name, p, nb=None, ntt=False):
        p = _arr(p).copy()
        f = getattr(lib(), name)
        if nb is None:
            f(self.h, _ptr(p), p.shape[0])
            return p[:-1].copy()
        f(self.h, _ptr(p), p.shape[0], nb, 1 if ntt else 0)
        return p[:p.shape[0] - nb].copy()


class ModupParams:
    def __init__(self, Q, P):
        self.p = lib().oc_modup_params_new(_u64arr(Q), len(Q), _u64arr(P), len(P))
        c = self.p.contents
        nQ, nP = len(Q), len(P)
        self.qib_mont = np.ctypeslib.as_array(c.qib_mont, shape=(nQ,)).copy()
        self.qispj_mont = np.ctypeslib.as_array(c.qispj_mont, shape=(nQ, nP)).copy()
        self.qpj_inv = np.ctypeslib.as_array(c.qpj_inv, shape=(nP, nQ + 1)).copy()

    def modup_exact(self, p_in, n_out):
        p_in = _arr(p_in)
        out = np.zeros((n_out, p_in.shape[1]), dtype=np.uint64)
        lib().oc_modup_exact(C.cast(self.p, C.c_void_p), _ptr(p_in), p_in.shape[0], _ptr(out), n_out, p_in.shape[1])
        return out

    def __del__(self):
        try:
            lib().oc_modup_params_free(C.cast(self.p, C.c_void_p))
        except Exception:
            pass


class Float128:
    """ring.Float128 (ring/float128.go): a pair of float64"""

    def __init__(self, hi=0.0, lo=0.0):
        self.v = (C.c_double * 2)(hi, lo)

    @staticmethod
    def SetUint53(i):
        f = Float128()
        lib().oc_f128_set_uint53(int(i), f.v)
        return f

    @staticmethod
    def SetUint64(i):
        f = Float128()
        lib().oc_f128_set_uint64(int(i), f.v)
        return f

    def _bin(self, name, other):
        f = Float128()
        getattr(lib(), name)(self.v, other.v, f.v)
        return f

    def Add(self, o): return self._bin("oc_f128_add", o)
    def Mul(self, o): return self._bin("oc_f128_mul", o)
    def Div(self, o): return self._bin("oc_f128_div", o)
    def ToUint64(self): return int(lib().oc_f128_to_uint64(self.v))
    def ToUint53(self): return int(lib().oc_f128_to_uint53(self.v))
    def pair(self): return (float(self.v[0]), float(self.v[1]))


class SimpleScaler:
    """ring.SimpleScaler (ring/ring_scaling.go:168-300)"""

    def __init__(self, t, ctx):
        self.t, self.ctx = int(t), ctx
        self.wi = np.zeros(ctx.L, dtype=np.uint64)
        self.ti = np.zeros((ctx.L, 2), dtype=np.float64)
        self.params = np.zeros(2, dtype=np.uint64)
        lib().oc_simple_scaler_new(ctx.h, self.t, self.wi.ctypes.data_as(u64p), self.ti.ctypes.data_as(C.POINTER(C.c_double)),
                                   self.params.ctypes.data_as(u64p))

    def scale(self, p1, limbs_out=1):
        p1 = _arr(p1)
        out = np.zeros((limbs_out, self.ctx.N), dtype=np.uint64)
        lib().oc_simple_scale(self.ctx.h, self.t, self.wi.ctypes.data_as(u64p), self.ti.ctypes.data_as(C.POINTER(C.c_double)),
                              self.params.ctypes.data_as(u64p), _ptr(p1), _ptr(out), limbs_out)
        return out


class BasisExtender:
    """ring.FastBasisExtender (ring/ring_basis_extension.go:9-74)."""

    def __init__(self, ctxQ, ctxP):
        self.cQ, self.cP = ctxQ, ctxP
        self._p = lib().oc_bext_new(ctxQ.h, ctxP.h)
        self.h = C.cast(self._p, C.c_void_p)
        c = self._p.contents
        self.moddown_params_pq = np.ctypeslib.as_array(c.moddown_pq, shape=(ctxQ.L,)).copy()
        self.moddown_params_qp = np.ctypeslib.as_array(c.moddown_qp, shape=(ctxP.L,)).copy()

    def __del__(self):
        try:
            lib().oc_bext_free(self.h)
        except Exception:
            pass

    def modup_split_qp(self, level, p1):
        p1 = _arr(p1)
        out = np.zeros((self.cP.L, self.cQ.N), dtype=np.uint64)
        lib().oc_modup_split_qp(self.h, level, _ptr(p1), _ptr(out))
        return out

    def modup_split_pq(self, level, p1):
        p1 = _arr(p1)
        out = np.zeros((self.cQ.L, self.cQ.N), dtype=np.uint64)
        lib().oc_modup_split_pq(self.h, level, _ptr(p1), _ptr(out))
        return out

    def moddown_ntt_pq(self, level, p1):
        p1 = _arr(p1).copy()
        out = np.zeros((level + 1, self.cQ.N), dtype=np.uint64)
        lib().oc_moddown_ntt_pq(self.h, level, _ptr(p1), _ptr(out))
        return out

    def moddown_split_ntt_pq(self, level, p1Q, p1P):
        p1Q, p1P = _arr(p1Q), _arr(p1P).copy()
        out = np.zeros((level + 1, self.cQ.N), dtype=np.uint64)
        lib().oc_moddown_split_ntt_pq(self.h, level, _ptr(p1Q), _ptr(p1P), _ptr(out))
        return out

    def moddown_pq(self, level, p1):
        p1 = _arr(p1)
        out = np.zeros((level + 1, self.cQ.N), dtype=np.uint64)
        lib().oc_moddown_pq(self.h, level, _ptr(p1), _ptr(out))
        return out

    def moddown_split_pq(self, level, p1Q, p1P):
        p1Q, p1P = _arr(p1Q), _arr(p1P)
        out = np.zeros((level + 1, self.cQ.N), dtype=np.uint64)
        lib().oc_moddown_split_pq(self.h, level, _ptr(p1Q), _ptr(p1P), _ptr(out))
        return out

    def moddown_split_qp(self, levelQ, levelP, p1Q, p1P):
        p1Q, p1P = _arr(p1Q), _arr(p1P)
        out = np.zeros((levelP + 1, self.cQ.N), dtype=np.uint64)
        lib().oc_moddown_split_qp(self.h, levelQ, levelP, _ptr(p1Q), _ptr(p1P), _ptr(out))
        return out


class BfvPlan:
    """The ring-level call sequence of bfv.Evaluator.Mul (bfv/evaluator.go:278-467)."""

    def __init__(self, ctxQ, ctxQMul, t):
        self.cQ, self.cM, self.t = ctxQ, ctxQMul, int(t)
        self.bext = BasisExtender(ctxQ, ctxQMul)            # baseconverterQ1Q2, bfv/evaluator.go:97
        P = 1
        for m in ctxQMul.moduli:
            P *= m
        self.p_half = P >> 1                                 # :100

    def mul(self, ct0, ct1):
        ct0, ct1 = _arr(ct0), _arr(ct1)
        out = np.zeros((3, self.cQ.L, self.cQ.N), dtype=np.uint64)
        pq = np.array([self.p_half % m for m in self.cQ.moduli], dtype=np.uint64)
        pm = np.array([self.p_half % m for m in self.cM.moduli], dtype=np.uint64)
        lib().oc_bfv_mul(self.bext.h, self.t, _ptr(pq), _ptr(pm), _ptr(ct0), _ptr(ct1), _ptr(out))
        return out

    def square(self, ct0):
        """evaluator.Mul(ct, ct, .) = tensorAndRescale's squaring case (bfv/evaluator.go:306,334-349)"""
        ct0 = _arr(ct0)
        out = np.zeros((3, self.cQ.L, self.cQ.N), dtype=np.uint64)
        pq = np.array([self.p_half % m for m in self.cQ.moduli], dtype=np.uint64)
        pm = np.array([self.p_half % m for m in self.cM.moduli], dtype=np.uint64)
        lib().oc_bfv_square(self.bext.h, self.t, _ptr(pq), _ptr(pm), _ptr(ct0), _ptr(out))
        return out


class Decomposer:
    """ring.Decomposer (ring/ring_basis_extension.go:398-472)."""

    def __init__(self, Q, P):
        self.Q, self.P = list(Q), list(P)
        self.h = lib().oc_decomposer_new(_u64arr(Q), len(Q), _u64arr(P), len(P))
        self.alpha = len(P)
        self.beta = -(-len(Q) // len(P))

    def __del__(self):
        try:
            lib().oc_decomposer_free(self.h)
        except Exception:
            pass

    def decompose(self, level, crt, p0):
        p0 = _arr(p0)
        N = p0.shape[1]
        out = np.zeros((level + 1 + len(self.P), N), dtype=np.uint64)
        lib().oc_decompose(self.h, level, crt, _ptr(p0), _ptr(out), N)
        return out

    def decompose_and_split(self, level, crt, p0):
        p0 = _arr(p0)
        N = p0.shape[1]
        outQ = np.zeros((level + 1, N), dtype=np.uint64)
        outP = np.zeros((len(self.P), N), dtype=np.uint64)
        lib().oc_decompose_and_split(self.h, level, crt, _ptr(p0), _ptr(outQ), _ptr(outP), N)
        return outQ, outP


class CkksPlan:
    """The ring-level call sequence of ckks.Evaluator.MulRelin (ckks/evaluator.go:1016,1475,1561)."""

    def __init__(self, ctxQ, ctxP):
        self.cQ, self.cP = ctxQ, ctxP
        self.h = lib().oc_ckks_plan_new(ctxQ.h, ctxP.h)

    def __del__(self):
        try:
            lib().oc_ckks_plan_free(self.h)
        except Exception:
            pass

    def switch_keys(self, level, cx, evk):
        cx, evk = _arr(cx), _arr(evk)
        p0 = np.zeros((level + 1, self.cQ.N), dtype=np.uint64)
        p1 = np.zeros_like(p0)
        lib().oc_ckks_switch_keys(self.h, level, _ptr(cx), _ptr(evk), _ptr(p0), _ptr(p1))
        return p0, p1

    def bfv_switch_keys(self, cx, evk):
        """bfv.evaluator.switchKeys (bfv/evaluator.go:736): cx [|Q|, N] coefficient domain -> (p0, p1) coefficient domain"""
        cx, evk = _arr(cx), _arr(evk)
        p0 = np.zeros((self.cQ.L, self.cQ.N), dtype=np.uint64)
        p1 = np.zeros_like(p0)
        lib().oc_bfv_switch_keys(self.h, _ptr(cx), _ptr(evk), _ptr(p0), _ptr(p1))
        return p0, p1

    def bfv_relinearize(self, ct, evk):
        """bfv.evaluator.Relinearize of a degree-2 ciphertext ct [3, |Q|, N] -> [2, |Q|, N]"""
        ct, evk = _arr(ct), _arr(evk)
        out = np.zeros((2, self.cQ.L, self.cQ.N), dtype=np.uint64)
        lib().oc_bfv_relinearize(self.h, _ptr(ct), _ptr(evk), _ptr(out))
        return out

    def bfv_permute(self, ct, gen, evk):
        """bfv.evaluator.permute (bfv/evaluator.go:711): RotateRows / RotateColumns with the key of that rotation; ct [2, |Q|, N]"""
        ct, evk = _arr(ct), _arr(evk)
        out = np.zeros((2, self.cQ.L, self.cQ.N), dtype=np.uint64)
        lib().oc_bfv_permute(self.h, _ptr(ct), C.c_uint64(int(gen)), _ptr(evk), _ptr(out))
        return out

    def permute_ntt(self, level, ct, gen, evk):
        ct, evk = _arr(ct), _arr(evk)
        out = np.zeros((2, level + 1, self.cQ.N), dtype=np.uint64)
        lib().oc_ckks_permute_ntt(self.h, level, _ptr(ct), C.c_uint64(int(gen)), _ptr(evk), _ptr(out))
        return out

    def rotate_hoisted(self, level, ct, gens, evks):
        ct = _arr(ct)
        evks = [_arr(e) for e in evks]
        n = len(gens)
        out = np.zeros((n, 2, level + 1, self.cQ.N), dtype=np.uint64)
        g = (C.c_uint64 * n)(*[int(x) for x in gens])
        ptrs = (C.c_void_p * n)(*[e.ctypes.data for e in evks])
        lib().oc_ckks_rotate_hoisted(self.h, level, _ptr(ct), n, g, ptrs, _ptr(out))
        return out

    def mul_norelin(self, level, ct0, ct1, squaring=False):
        """MulRelin with evakey == nil (ckks/evaluator.go:1038-1111): degree-2 output [3, level+1, N]"""
        ct0, ct1 = _arr(ct0), _arr(ct1)
        out = np.zeros((3, level + 1, self.cQ.N), dtype=np.uint64)
        lib().oc_ckks_mul_norelin(self.h, level, _ptr(ct0), _ptr(ct1), 1 if squaring else 0, _ptr(out))
        return out

    def mul_plain(self, level, pt, ct):
        """MulRelin, plaintext x ciphertext (ckks/evaluator.go:1113-1131)"""
        pt, ct = _arr(pt), _arr(ct)
        out = np.zeros((2, level + 1, self.cQ.N), dtype=np.uint64)
        lib().oc_ckks_mul_plain(self.h, level, _ptr(pt), _ptr(ct), _ptr(out))
        return out

    def encrypt_pk(self, ctxQP, level, u, pk0, pk1, e0, e1, pt):
        """pkEncryptor.encrypt after the sampling (ckks/encryptor.go:205-234)"""
        u, pk0, pk1, e0, e1, pt = (_arr(x) for x in (u, pk0, pk1, e0, e1, pt))
        out = np.zeros((2, level + 1, self.cQ.N), dtype=np.uint64)
        lib().oc_ckks_encrypt_pk(self.h, ctxQP.h, level, _ptr(u), _ptr(pk0), _ptr(pk1), _ptr(e0), _ptr(e1), _ptr(pt), _ptr(out))
        return out

    def decrypt(self, level, ct, sk):
        """decryptor.Decrypt (ckks/decryptor.go:53-78); ct = [degree+1, level+1, N]"""
        ct, sk = _arr(ct), _arr(sk)
        out = np.zeros((level + 1, self.cQ.N), dtype=np.uint64)
        lib().oc_ckks_decrypt(self.h, level, _ptr(ct), ct.shape[0] - 1, _ptr(sk), _ptr(out))
        return out

    def mulrelin(self, level, ct0, ct1, evk):
        ct0, ct1, evk = _arr(ct0), _arr(ct1), _arr(evk)
        out = np.zeros((2, level + 1, self.cQ.N), dtype=np.uint64)
        lib().oc_ckks_mulrelin(self.h, level, _ptr(ct0), _ptr(ct1), _ptr(evk), _ptr(out))
        return out

"""Host-side mirror of Lattigo's ``ring`` package over the C ABI (include/lattigo_ring.h).

Same exported names, argument order and meaning as the reference (github.com/ldsec/lattigo/ring,
v1.3.1) so that code and tests written against ``ring.Context`` read the same:

    ctx = ring.NewContextWithParams(N, moduli)      # ring/ring_context.go:60
    p = ctx.NewPoly()                               # ring/ring_context.go:288
    ctx.NTT(p, p)                                   # ring/ntt.go:4
    ctx.MulCoeffsMontgomery(a, b, c)                # ring/ring.go:221

Differences forced by the device boundary: a ``Poly`` is a *batch* of polynomials resident in
HBM ((poly, limb, coeff)-major uint64); ``Poly.set``/``Poly.get`` move numpy arrays of shape
[batch, limbs, N] (or [limbs, N] when batch == 1).  Errors the reference reports by panicking
are raised as ``LatticeRingError``.
"""
import ctypes as C

import numpy as np

from . import _native as nat
from ._native import LatticeRingError, Options, check, lib

# lr_ewise_op (include/lattigo_ring.h)
OPS = ["ADD", "ADD_NOMOD", "SUB", "SUB_NOMOD", "NEG", "REDUCE", "MUL_COEFFS", "MUL_COEFFS_AND_ADD",
       "MUL_COEFFS_AND_ADD_NOMOD", "MUL_COEFFS_CONSTANT", "MUL_MONT", "MUL_MONT_AND_ADD",
       "MUL_MONT_AND_ADD_NOMOD", "MUL_MONT_CONSTANT_AND_ADD_NOMOD", "MUL_MONT_AND_SUB",
       "MUL_MONT_AND_SUB_NOMOD", "MUL_MONT_CONSTANT", "MFORM", "INV_MFORM", "MUL_SCALAR",
       "MUL_SCALAR_LIMBS", "ADD_SCALAR_LIMBS", "SUB_SCALAR_LIMBS", "COPY", "MUL_BY_POW2"]
OP = {n: i for i, n in enumerate(OPS)}

TAB = {"MODULUS": 0, "BRED": 1, "MRED": 2, "PSI_MONT": 3, "PSI_INV_MONT": 4, "NTT_PSI": 5, "NTT_PSI_INV": 6,
       "NTT_N_INV": 7, "RESCALE": 8, "MASK": 9}


def _u64(vals):
    vals = [int(v) for v in vals]
    return (C.c_uint64 * len(vals))(*vals)


class Poly:
    """ring.Poly (ring/ring_object.go:11-13), as a device-resident batch."""

    def __init__(self, ctx, limbs, batch=1, _handle=None):
        self.ctx = ctx
        self.N = ctx.N
        self.alloc_limbs = limbs
        self.batch = batch
        if _handle is None:
            h = C.c_void_p()
            check(lib().lr_poly_alloc(ctx.h, limbs, batch, C.byref(h)))
            _handle = h
        self.h = _handle

    @classmethod
    def wrap(cls, ctx, device_ptr, limbs, batch):
        """A Poly over caller-owned device memory (e.g. a torch int64 tensor's data_ptr()), no copy (lr_poly_wrap)."""
        h = C.c_void_p()
        check(lib().lr_poly_wrap(ctx.h, C.c_void_p(device_ptr), limbs, batch, C.byref(h)))
        return cls(ctx, limbs, batch, _handle=h)

    @classmethod
    def wrap_strided(cls, ctx, device_ptr, limbs, batch, stride_limbs):
        """the same with `stride_limbs` limbs between consecutive polys (lr_poly_wrap_strided)"""
        h = C.c_void_p()
        check(lib().lr_poly_wrap_strided(ctx.h, C.c_void_p(device_ptr), limbs, batch, stride_limbs * ctx.N, C.byref(h)))
        return cls(ctx, limbs, batch, _handle=h)

    # --- reference accessors -----------------------------------------------------------
    def GetLenModuli(self):  # ring/ring_object.go:55
        n = C.c_int()
        check(lib().lr_poly_info(self.h, None, C.byref(n), None, None))
        return n.value

    def GetDegree(self):  # ring/ring_object.go:50
        return self.N

    def Zero(self):  # ring/ring_object.go:60
        check(lib().lr_poly_zero(self.h))

    @property
    def limbs(self):
        return self.GetLenModuli()

    @property
    def device_ptr(self):
        p = C.c_void_p()
        check(lib().lr_poly_info(self.h, None, None, None, C.byref(p)))
        return p.value

    # --- data movement -----------------------------------------------------------------
    def set(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.uint64)
        limbs = self.limbs
        if a.ndim == 2:
            a = a[None]
        if a.shape != (self.batch, limbs, self.N):
            raise LatticeRingError(3, "expected shape %s, got %s" % ((self.batch, limbs, self.N), a.shape))
        check(lib().lr_poly_upload_dense(self.h, a.ctypes.data_as(C.c_void_p), a.size))
        return self

    def get(self):
        limbs = self.limbs
        out = np.empty((self.batch, limbs, self.N), dtype=np.uint64)
        check(lib().lr_poly_download_dense(self.h, out.ctypes.data_as(C.c_void_p), out.size))
        return out[0] if self.batch == 1 else out

    def UnmarshalBinary(self, data, batch_index=0):
        """Poly.UnmarshalBinary (ring/ring_object.go:252): the big-endian image goes to the device as it is."""
        data = bytes(data)
        check(lib().lr_poly_unmarshal(self.h, batch_index, data, len(data)))
        return self

    def MarshalBinary(self, batch_index=0):
        """Poly.MarshalBinary (ring/ring_object.go:222)."""
        buf = (C.c_uint8 * (2 + 8 * self.limbs * self.N))()
        n = C.c_size_t(0)
        check(lib().lr_poly_marshal(self.h, batch_index, buf, len(buf), C.byref(n)))
        return bytes(buf[:n.value])

    def set_limb_slices(self, batch_index, limb_arrays):
        """Go boundary form: one independent array per limb (``[][]uint64``)."""
        arrs = [np.ascontiguousarray(x, dtype=np.uint64) for x in limb_arrays]
        ptrs = (nat.u64p * len(arrs))(*[x.ctypes.data_as(nat.u64p) for x in arrs])
        check(lib().lr_poly_upload(self.h, batch_index, ptrs, len(arrs)))

    def get_limb_slices(self, batch_index, limbs=None):
        limbs = self.limbs if limbs is None else limbs
        arrs = [np.empty(self.N, dtype=np.uint64) for _ in range(limbs)]
        ptrs = (nat.u64p * limbs)(*[x.ctypes.data_as(nat.u64p) for x in arrs])
        check(lib().lr_poly_download(self.h, batch_index, ptrs, limbs))
        return arrs

    def CopyNew(self):  # ring/ring_object.go:67
        p = Poly(self.ctx, self.alloc_limbs, self.batch)
        check(lib().lr_poly_set_limbs(p.h, self.limbs))
        self.ctx._ew("COPY", self.limbs - 1, self, None, p)
        return p

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().lr_poly_free(self.h)
                self.h = None
        except Exception:
            pass


class Context:
    """ring.Context (ring/ring_context.go:18-51)."""

    def __init__(self, N, Moduli, device=0, options=None):
        self.N = int(N)
        self.Modulus = [int(m) for m in Moduli]
        self.device = device
        h = C.c_void_p()
        if options is None:
            check(lib().lr_context_create(self.N, _u64(self.Modulus), len(self.Modulus), device, C.byref(h)))
        else:
            check(lib().lr_context_create_ex(self.N, _u64(self.Modulus), len(self.Modulus), device, C.byref(options), C.byref(h)))
        self.h = h

    def GetOptions(self):
        """the lr_options this context ended up with (after the test-only LR_* override), as an Options"""
        o = Options()
        check(lib().lr_context_get_options(self.h, C.byref(o)))
        return o

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().lr_context_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # --- getters (ring/ring_context.go:253-285) ----------------------------------------------
    def _table(self, name, shape):
        out = np.empty(shape, dtype=np.uint64)
        check(lib().lr_context_get_table(self.h, TAB[name], out.ctypes.data_as(nat.u64p), out.size))
        return out

    def GetBredParams(self):
        return self._table("BRED", (len(self.Modulus), 2))

    def GetMredParams(self):
        return self._table("MRED", (len(self.Modulus),))

    def GetPsi(self):
        return self._table("PSI_MONT", (len(self.Modulus),))

    def GetPsiInv(self):
        return self._table("PSI_INV_MONT", (len(self.Modulus),))

    def GetNttPsi(self):
        return self._table("NTT_PSI", (len(self.Modulus), self.N))

    def GetNttPsiInv(self):
        return self._table("NTT_PSI_INV", (len(self.Modulus), self.N))

    def GetNttNInv(self):
        return self._table("NTT_N_INV", (len(self.Modulus),))

    def GetRescaleParams(self):
        return self._table("RESCALE", (len(self.Modulus), len(self.Modulus)))

    def Sync(self):
        check(lib().lr_context_sync(self.h))

    def SetStream(self, stream_ptr):
        """an externally owned hipStream_t handle, or None for the library's own stream.  Handle 0 is HIP's legacy default stream,
        which the C ABI cannot express (NULL there selects the library's stream, and that one does not synchronise with the legacy
        stream): it is refused here instead of silently running somewhere else than the caller thinks."""
        if stream_ptr is None:
            check(lib().lr_context_set_stream(self.h, C.c_void_p(None)))
            return
        if int(stream_ptr) == 0:
            raise LatticeRingError(1, "SetStream(0): the legacy default stream cannot be selected; create an explicit stream "
                                      "(torch.cuda.Stream()) and pass its handle, or None for the library's own stream")
        check(lib().lr_context_set_stream(self.h, C.c_void_p(int(stream_ptr))))

    # --- allocation (ring/ring_context.go:288,300) -------------------------------------------
    def NewPoly(self, batch=1):
        return Poly(self, len(self.Modulus), batch)

    def NewPolyLvl(self, level, batch=1):
        return Poly(self, level + 1, batch)

    # --- NTT (ring/ntt.go:4-29) ---------------------------------------------------------------
    def NTT(self, p1, p2):
        check(lib().lr_ntt(self.h, len(self.Modulus) - 1, p1.h, p2.h))

    def NTTLvl(self, level, p1, p2):
        check(lib().lr_ntt(self.h, level, p1.h, p2.h))

    def InvNTT(self, p1, p2):
        check(lib().lr_intt(self.h, len(self.Modulus) - 1, p1.h, p2.h))

    def InvNTTLvl(self, level, p1, p2):
        check(lib().lr_intt(self.h, level, p1.h, p2.h))

    def NTTLimb(self, mod_index, p1, limb1, p2, limb2):  # package-level ring.NTT (ring/ntt.go:53)
        check(lib().lr_ntt_limb(self.h, mod_index, p1.h, limb1, p2.h, limb2))

    def InvNTTLimb(self, mod_index, p1, limb1, p2, limb2):  # ring.InvNTT (ring/ntt.go:89)
        check(lib().lr_intt_limb(self.h, mod_index, p1.h, limb1, p2.h, limb2))

    def NTTHost(self, limb_arrays):
        """Literal Go call shape: per-limb host slices in, per-limb host slices out."""
        ins = [np.ascontiguousarray(x, dtype=np.uint64) for x in limb_arrays]
        outs = [np.empty(self.N, dtype=np.uint64) for _ in ins]
        pi = (nat.u64p * len(ins))(*[x.ctypes.data_as(nat.u64p) for x in ins])
        po = (nat.u64p * len(ins))(*[x.ctypes.data_as(nat.u64p) for x in outs])
        check(lib().lr_ntt_host(self.h, len(ins) - 1, pi, po))
        return outs

    def InvNTTHost(self, limb_arrays):
        ins = [np.ascontiguousarray(x, dtype=np.uint64) for x in limb_arrays]
        outs = [np.empty(self.N, dtype=np.uint64) for _ in ins]
        pi = (nat.u64p * len(ins))(*[x.ctypes.data_as(nat.u64p) for x in ins])
        po = (nat.u64p * len(ins))(*[x.ctypes.data_as(nat.u64p) for x in outs])
        check(lib().lr_intt_host(self.h, len(ins) - 1, pi, po))
        return outs

    # --- coefficient-wise family (ring/ring.go) ------------------------------------------------
    def _ew(self, op, level, a, b, out, scalars=None):
        sc = None if scalars is None else _u64(scalars)
        check(lib().lr_ewise(self.h, OP[op], level, a.h, None if b is None else b.h, out.h, sc))

    def _full(self):
        return len(self.Modulus) - 1

    def Add(self, p1, p2, p3): self._ew("ADD", self._full(), p1, p2, p3)
    def AddLvl(self, level, p1, p2, p3): self._ew("ADD", level, p1, p2, p3)
    def AddNoMod(self, p1, p2, p3): self._ew("ADD_NOMOD", self._full(), p1, p2, p3)
    def AddNoModLvl(self, level, p1, p2, p3): self._ew("ADD_NOMOD", level, p1, p2, p3)
    def Sub(self, p1, p2, p3): self._ew("SUB", self._full(), p1, p2, p3)
    def SubLvl(self, level, p1, p2, p3): self._ew("SUB", level, p1, p2, p3)
    def SubNoMod(self, p1, p2, p3): self._ew("SUB_NOMOD", self._full(), p1, p2, p3)
    def SubNoModLvl(self, level, p1, p2, p3): self._ew("SUB_NOMOD", level, p1, p2, p3)
    def Neg(self, p1, p2): self._ew("NEG", self._full(), p1, None, p2)
    def NegLvl(self, level, p1, p2): self._ew("NEG", level, p1, None, p2)
    def Reduce(self, p1, p2): self._ew("REDUCE", self._full(), p1, None, p2)
    def ReduceLvl(self, level, p1, p2): self._ew("REDUCE", level, p1, None, p2)
    def MulCoeffs(self, p1, p2, p3): self._ew("MUL_COEFFS", self._full(), p1, p2, p3)
    def MulCoeffsAndAdd(self, p1, p2, p3): self._ew("MUL_COEFFS_AND_ADD", self._full(), p1, p2, p3)
    def MulCoeffsAndAddNoMod(self, p1, p2, p3): self._ew("MUL_COEFFS_AND_ADD_NOMOD", self._full(), p1, p2, p3)
    def MulCoeffsConstant(self, p1, p2, p3): self._ew("MUL_COEFFS_CONSTANT", self._full(), p1, p2, p3)
    def MulCoeffsMontgomery(self, p1, p2, p3): self._ew("MUL_MONT", self._full(), p1, p2, p3)
    def MulCoeffsMontgomeryLvl(self, level, p1, p2, p3): self._ew("MUL_MONT", level, p1, p2, p3)
    def MulCoeffsMontgomeryAndAdd(self, p1, p2, p3): self._ew("MUL_MONT_AND_ADD", self._full(), p1, p2, p3)
    def MulCoeffsMontgomeryAndAddLvl(self, level, p1, p2, p3): self._ew("MUL_MONT_AND_ADD", level, p1, p2, p3)
    def MulCoeffsMontgomeryAndAddNoMod(self, p1, p2, p3): self._ew("MUL_MONT_AND_ADD_NOMOD", self._full(), p1, p2, p3)
    def MulCoeffsMontgomeryAndAddNoModLvl(self, level, p1, p2, p3): self._ew("MUL_MONT_AND_ADD_NOMOD", level, p1, p2, p3)
    def MulCoeffsMontgomeryConstantAndAddNoModLvl(self, level, p1, p2, p3):
        self._ew("MUL_MONT_CONSTANT_AND_ADD_NOMOD", level, p1, p2, p3)
    def MulCoeffsMontgomeryAndSub(self, p1, p2, p3): self._ew("MUL_MONT_AND_SUB", self._full(), p1, p2, p3)
    def MulCoeffsMontgomeryAndSubNoMod(self, p1, p2, p3): self._ew("MUL_MONT_AND_SUB_NOMOD", self._full(), p1, p2, p3)
    def MulCoeffsMontgomeryConstant(self, p1, p2, p3): self._ew("MUL_MONT_CONSTANT", self._full(), p1, p2, p3)
    def MForm(self, p1, p2): self._ew("MFORM", self._full(), p1, None, p2)
    def MFormLvl(self, level, p1, p2): self._ew("MFORM", level, p1, None, p2)
    def InvMForm(self, p1, p2): self._ew("INV_MFORM", self._full(), p1, None, p2)
    def MulScalar(self, p1, scalar, p2): self._ew("MUL_SCALAR", self._full(), p1, None, p2, [scalar])
    def MulScalarLvl(self, level, p1, scalar, p2): self._ew("MUL_SCALAR", level, p1, None, p2, [scalar])
    def Copy(self, p0, p1): self._ew("COPY", self._full(), p0, None, p1)
    def CopyLvl(self, level, p0, p1): self._ew("COPY", level, p0, None, p1)
    def MulByPow2(self, p1, pow2, p2): self._ew("MUL_BY_POW2", self._full(), p1, None, p2, [pow2])
    def MulByPow2Lvl(self, level, p1, pow2, p2): self._ew("MUL_BY_POW2", level, p1, None, p2, [pow2])

    def MulScalarBigint(self, p1, scalar, p2):  # ring/ring.go:541
        self._ew("MUL_SCALAR_LIMBS", self._full(), p1, None, p2, [int(scalar) % q for q in self.Modulus])

    def MulScalarBigintLvl(self, level, p1, scalar, p2):  # ring/ring.go:557
        self._ew("MUL_SCALAR_LIMBS", level, p1, None, p2, [int(scalar) % q for q in self.Modulus[:level + 1]])

    def AddScalarBigint(self, p1, scalar, p2):
        # the reference writes into p1, not p2 (ring/ring.go:482: p1tmp, p2tmp := p1.Coeffs[i], p1.Coeffs[i])
        self._ew("ADD_SCALAR_LIMBS", self._full(), p1, None, p1, [int(scalar) % q for q in self.Modulus])

    def SubScalarBigint(self, p1, scalar, p2):  # ring/ring.go:500, same quirk
        self._ew("SUB_SCALAR_LIMBS", self._full(), p1, None, p1, [int(scalar) % q for q in self.Modulus])

    def MulPolyMontgomery(self, p1, p2, p3):  # ring/ring.go:370
        a, b = self.NewPoly(p1.batch), self.NewPoly(p1.batch)
        self.NTT(p1, a)
        self.NTT(p2, b)
        self.MulCoeffsMontgomery(a, b, p3)
        self.InvNTT(p3, p3)

    # --- Galois automorphisms (ring/ring_galois.go) ------------------------------------------------
    def Permute(self, polIn, gen, polOut):  # :106
        check(lib().lr_permute(self.h, polIn.h, int(gen), polOut.h))

    def ntt_variants(self):
        """(forward, inverse) assembly variants this context selects (diagnostics)"""
        f, i = C.c_int(), C.c_int()
        check(lib().lr_context_ntt_variants(self.h, C.byref(f), C.byref(i)))
        return f.value, i.value

    def last_ntt_kernel(self):
        """name of the kernel the last NTT / InvNTT launch of this context dispatched (diagnostics)"""
        buf = C.create_string_buffer(64)
        check(lib().lr_context_last_ntt_kernel(self.h, buf, len(buf)))
        return buf.value.decode()

    def selftest_division(self, samples, seed=1):
        """diagnostics: quotients of the extension's constant-divisor division that differ from IEEE division (must be 0)"""
        n = C.c_uint64(0)
        check(lib().lr_selftest_division(self.h, samples, seed, C.byref(n)))
        return int(n.value)

    def timeline(self):
        """clock stamps of the last launch of the stamped diagnostics kernel (context created under LR_NTT_TIMELINE=1):
        uint32 array [workgroups, 16 waves, 16 stamps]"""
        n = C.c_size_t(0)
        check(lib().lr_context_timeline(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            check(lib().lr_context_timeline(self.h, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)))
        return out.reshape(-1, 16, 16)

    def HalfScalarOp(self, op, level, p1, lo, hi, p2):
        """the element loops of ckks.Evaluator's AddConst / MultByConst / MultByConstAndAdd / MultByi / DivByi (ckks/evaluator.go:429-828):
        op "ADD" CRed(x + s), "MRED" MRed(x, s), "MRED_ADD" CRed(out + MRed(x, s)); s = lo[i] below N/2, hi[i] above"""
        lo = np.ascontiguousarray(lo, dtype=np.uint64)
        hi = np.ascontiguousarray(hi, dtype=np.uint64)
        if lo.size < level + 1 or hi.size < level + 1:
            raise LatticeRingError(3, "half-vector scalars: need level+1 words each")
        check(lib().lr_half_scalar_op(self.h, {"ADD": 0, "MRED": 1, "MRED_ADD": 2}[op], level, p1.h, lo.ctypes.data_as(C.c_void_p),
                                      hi.ctypes.data_as(C.c_void_p), p2.h))

    def MultByMonomial(self, p1, monomialDeg, p2):  # ring/ring.go:663
        check(lib().lr_mult_by_monomial(self.h, p1.h, int(monomialDeg), p2.h))

    def Shift(self, p1, n, p2):  # ring/ring.go:575
        check(lib().lr_shift(self.h, p1.h, int(n), p2.h))

    def Rotate(self, p1, n, p2=None):  # ring/ring.go:775 -- the reference writes into p1 and ignores p2 (:791)
        check(lib().lr_rotate(self.h, p1.h, int(n)))

    def PermuteNTTLvl(self, level, polIn, gen, polOut):  # package-level PermuteNTT (:55) on limbs 0..level
        check(lib().lr_permute_ntt(self.h, level, polIn.h, int(gen), polOut.h))

    # --- RNS rescale (ring/ring_scaling.go:9-164) ----------------------------------------------
    def DivFloorByLastModulusNTT(self, p0): check(lib().lr_div_floor_by_last_modulus_ntt(self.h, p0.h))
    def DivFloorByLastModulus(self, p0): check(lib().lr_div_floor_by_last_modulus(self.h, p0.h))
    def DivRoundByLastModulusNTT(self, p0): check(lib().lr_div_round_by_last_modulus_ntt(self.h, p0.h))
    def DivRoundByLastModulus(self, p0): check(lib().lr_div_round_by_last_modulus(self.h, p0.h))
    def DivFloorByLastModulusMany(self, p0, nb): check(lib().lr_div_floor_by_last_modulus_many(self.h, p0.h, nb, 0))
    def DivFloorByLastModulusManyNTT(self, p0, nb): check(lib().lr_div_floor_by_last_modulus_many(self.h, p0.h, nb, 1))
    def DivRoundByLastModulusMany(self, p0, nb): check(lib().lr_div_round_by_last_modulus_many(self.h, p0.h, nb, 0))
    def DivRoundByLastModulusManyNTT(self, p0, nb): check(lib().lr_div_round_by_last_modulus_many(self.h, p0.h, nb, 1))

    # --- multi-device (one process, one thread per device): SURVEY 8(e) ---------------------------
    def CopyPeer(self, dst, dst_index, src_ctx, src, src_index, count):
        """polys [src_index, +count) of src (on src_ctx's device) -> slots [dst_index, ...) of dst (on this context's device), on the copy
        stream of that device pair, behind what src_ctx has enqueued so far (lr_poly_copy_peer); asynchronous"""
        check(lib().lr_poly_copy_peer(self.h, dst.h, dst_index, src_ctx.h, src.h, src_index, count))

    def WaitPeerCopies(self):
        """this context's stream waits (on the device) for every peer copy into its device enqueued so far"""
        check(lib().lr_context_wait_peer_copies(self.h))

    def GatherBlocks(self, dst, blocks):
        """blocks: [(src_ctx, src_poly, count)] -> dst, one behind the other in block order, then WaitPeerCopies (lr_gather_blocks)"""
        n = len(blocks)
        ctxs = (C.c_void_p * n)(*[b[0].h.value for b in blocks])
        polys = (C.c_void_p * n)(*[b[1].h.value for b in blocks])
        counts = (C.c_int * n)(*[int(b[2]) for b in blocks])
        check(lib().lr_gather_blocks(self.h, dst.h, ctxs, polys, counts, n))

    # --- measurement -------------------------------------------------------------------------
    def TimerStart(self):
        check(lib().lr_timer_start(self.h))

    def TimerStop(self):
        ms = C.c_float()
        check(lib().lr_timer_stop(self.h, C.byref(ms)))
        return ms.value


def GenGaloisParams(n, gen):
    """ring.GenGaloisParams (ring/ring_galois.go:9): powers of gen modulo 2n"""
    out, mask = [1], (n << 1) - 1
    for _ in range(1, n >> 1):
        out.append((out[-1] * gen) & mask)
    return out


def PermuteNTTIndex(gen, power, N):
    """ring.PermuteNTTIndex (ring/ring_galois.go:29)"""
    idx = np.empty(N, dtype=np.uint64)
    check(lib().lr_permute_ntt_index(int(gen), int(power), int(N), idx.ctypes.data_as(nat.u64p)))
    return idx


def PermuteNTT(context, polIn, gen, polOut):
    """ring.PermuteNTT (ring/ring_galois.go:55): all limbs of polIn"""
    context.PermuteNTTLvl(polIn.limbs - 1, polIn, gen, polOut)


def NewContextWithParams(N, Moduli, device=0, options=None):
    """ring.NewContextWithParams (ring/ring_context.go:60).  Raises LatticeRingError
    LR_ERR_NOT_NTT_FRIENDLY where the reference returns its error value.  options: an Options (lr_options) or None for the defaults."""
    return Context(N, Moduli, device, options)


class FastBasisExtender:
    """ring.FastBasisExtender (ring/ring_basis_extension.go:9-74)."""

    def __init__(self, contextQ, contextP):
        self.contextQ, self.contextP = contextQ, contextP
        h = C.c_void_p()
        check(lib().lr_bext_create(contextQ.h, contextP.h, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().lr_bext_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def ModDownParamsPQ(self):
        out = np.empty(len(self.contextQ.Modulus), dtype=np.uint64)
        check(lib().lr_bext_get_table(self.h, 0, out.ctypes.data_as(nat.u64p), out.size))
        return out

    def ModDownParamsQP(self):
        out = np.empty(len(self.contextP.Modulus), dtype=np.uint64)
        check(lib().lr_bext_get_table(self.h, 1, out.ctypes.data_as(nat.u64p), out.size))
        return out

    def ModUpSplitQP(self, level, p1, p2): check(lib().lr_modup_split_qp(self.h, level, p1.h, p2.h))
    def ModUpSplitPQ(self, level, p1, p2): check(lib().lr_modup_split_pq(self.h, level, p1.h, p2.h))
    def ModDownNTTPQ(self, level, p1, p2): check(lib().lr_moddown_ntt_pq(self.h, level, p1.h, p2.h))
    def ModDownSplitedNTTPQ(self, level, p1Q, p1P, p2): check(lib().lr_moddown_split_ntt_pq(self.h, level, p1Q.h, p1P.h, p2.h))
    def ModDownPQ(self, level, p1, p2): check(lib().lr_moddown_pq(self.h, level, p1.h, p2.h))
    def ModDownSplitedPQ(self, level, p1Q, p1P, p2): check(lib().lr_moddown_split_pq(self.h, level, p1Q.h, p1P.h, p2.h))
    def ModDownSplitedQP(self, levelQ, levelP, p1Q, p1P, p2):
        check(lib().lr_moddown_split_qp(self.h, levelQ, levelP, p1Q.h, p1P.h, p2.h))


def NewFastBasisExtender(contextQ, contextP):
    return FastBasisExtender(contextQ, contextP)


class Decomposer:
    """ring.Decomposer (ring/ring_basis_extension.go:398-472).  The reference constructor takes the
    modulus lists; here the two contexts (which carry them) so the handle knows its device."""

    def __init__(self, contextQ, contextP):
        self.contextQ, self.contextP = contextQ, contextP
        h = C.c_void_p()
        check(lib().lr_decomposer_create(contextQ.h, contextP.h, C.byref(h)))
        self.h = h
        nQ, nP = len(contextQ.Modulus), len(contextP.Modulus)
        self.alpha = nP
        self.beta = -(-nQ // nP)
        self.xalpha = [nP] * self.beta
        if nQ % nP:
            self.xalpha[-1] = nQ % nP

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().lr_decomposer_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def Xalpha(self):
        return list(self.xalpha)

    def Decompose(self, level, crtDecompLevel, p0, p1):
        check(lib().lr_decompose(self.h, level, crtDecompLevel, p0.h, p1.h))

    def DecomposeAndSplit(self, level, crtDecompLevel, p0, p1Q, p1P):
        check(lib().lr_decompose_and_split(self.h, level, crtDecompLevel, p0.h, p1Q.h, p1P.h))


def NewDecomposer(contextQ, contextP):
    return Decomposer(contextQ, contextP)


class SimpleScaler:
    """ring.SimpleScaler (ring/ring_scaling.go:168-300): reconstruct a polynomial of `context`, scale it by t/Q and
    return it modulo t."""

    def __init__(self, t, context):
        self.t, self.context = int(t), context
        h = C.c_void_p()
        check(lib().lr_simple_scaler_create(context.h, self.t, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().lr_simple_scaler_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def tables(self):
        """(wi[L], ti[L][2]) as computed on the host by NewSimpleScaler"""
        n = len(self.context.Modulus)
        wi = np.zeros(n, dtype=np.uint64)
        ti = np.zeros((n, 2), dtype=np.float64)
        check(lib().lr_simple_scaler_tables(self.h, wi.ctypes.data_as(C.POINTER(C.c_uint64)), ti.ctypes.data_as(C.POINTER(C.c_double)), n))
        return wi, ti

    def Scale(self, p1, p2):  # :275
        check(lib().lr_simple_scale(self.h, p1.h, p2.h))


def NewSimpleScaler(t, context):  # ring/ring_scaling.go:186
    return SimpleScaler(t, context)


class CkksPlan:
    """What ckks.NewEvaluator builds around the ring (ckks/evaluator.go:81-112) plus the
    MulRelin / switchKeysInPlace / Rescale call sequences (:1016, :1475, :933), device-resident."""

    def __init__(self, contextQ, contextP, max_batch=1, options=None):
        self.contextQ, self.contextP = contextQ, contextP
        h = C.c_void_p()
        if options is None:
            check(lib().lr_ckks_plan_create(contextQ.h, contextP.h, max_batch, C.byref(h)))
        else:
            check(lib().lr_ckks_plan_create_ex(contextQ.h, contextP.h, max_batch, C.byref(options), C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().lr_ckks_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def Stats(self):
        """diagnostics: forks (independent launches of a small batch side by side), grouped digit extensions (lr_ckks_plan_stats)"""
        f, g = C.c_uint64(), C.c_uint64()
        check(lib().lr_ckks_plan_stats(self.h, C.byref(f), C.byref(g)))
        return {"forks": f.value, "grouped_extensions": g.value}

    def NewSwitchingKey(self, batch=1):
        """Storage for SwitchingKey.evakey (ckks/keygen.go:68-70): beta x 2 polys over Q||P."""
        nQ, nP = len(self.contextQ.Modulus), len(self.contextP.Modulus)
        beta = -(-nQ // nP)
        return Poly(self.contextQ, nQ + nP, 2 * beta)

    def SwitchKeysInPlace(self, level, cx, evakey, p0, p1):
        check(lib().lr_ckks_switch_keys(self.h, level, cx.h, evakey.h, p0.h, p1.h))

    # bfv.NewEvaluator builds the same objects for its key switch (decomposer, baseconverterQ1P, pools: bfv/evaluator.go:100-112):
    # one plan over (contextQ, contextP) serves both schemes
    def BfvSwitchKeys(self, cx, evakey, p0, p1):
        """bfv.evaluator.switchKeys (bfv/evaluator.go:736): coefficient domain in and out, all of Q"""
        check(lib().lr_bfv_switch_keys(self.h, cx.h, evakey.h, p0.h, p1.h))

    def BfvRelinearize(self, ct, evakey, ctOut):
        """bfv.evaluator.Relinearize (bfv/evaluator.go:512) of a degree-2 ciphertext: ct = (c0, c1, c2), ctOut = (out0, out1)"""
        check(lib().lr_bfv_relinearize(self.h, ct[0].h, ct[1].h, ct[2].h, evakey.h, ctOut[0].h, ctOut[1].h))

    def BfvPermute(self, ct0, generator, switchKey, ctOut):
        """bfv.evaluator.permute (bfv/evaluator.go:711): the body of RotateRows (generator = galElRotRow) and of RotateColumns with
        the key of that rotation (galElRotColLeft[k]); ct0, ctOut: pairs of Poly over Q, coefficient domain; ctOut may be ct0"""
        check(lib().lr_bfv_rotate(self.h, ct0[0].h, ct0[1].h, C.c_uint64(int(generator)), switchKey.h, ctOut[0].h, ctOut[1].h))

    def BfvRotateColumnsPow2(self, ct0, generator, k, pow2_keys, ctOut):
        """bfv.evaluator.rotateColumnsPow2 (bfv/evaluator.go:636-662): rotation by k as the chain of the power-of-two rotations in
        its binary expansion; pow2_keys: {2^i: SwitchingKey image}; generator = GaloisGen (left) or its inverse mod 2N (right)"""
        mask = (self.contextQ.N << 1) - 1
        if ctOut[0] is not ct0[0]:
            self.contextQ.Copy(ct0[0], ctOut[0])                     # :647-648
            self.contextQ.Copy(ct0[1], ctOut[1])
        idx = 1
        while k > 0:
            if k & 1:
                self.BfvPermute(ctOut, generator, pow2_keys[idx], ctOut)   # :655
            generator = (generator * generator) & mask                     # :658-659
            idx <<= 1
            k >>= 1

    def MulRelin(self, level, ct0, ct1, evakey, ctOut):
        """evaluator.MulRelin (ckks/evaluator.go:1016).  ct0, ct1: tuples of Poly -- (value[0], value[1]) for a ciphertext,
        (value[0],) for a plaintext; ctOut: pair, or triple when evakey is None and both operands are ciphertexts
        (degree-2 result, :1061-1066)."""
        if len(ct0) + len(ct1) == 3:                                     # plaintext x ciphertext, :1113-1131
            pt, ct = (ct0, ct1) if len(ct0) == 1 else (ct1, ct0)
            check(lib().lr_ckks_mul_plain(self.h, level, pt[0].h, ct[0].h, ct[1].h, ctOut[0].h, ctOut[1].h))
        elif evakey is None:
            check(lib().lr_ckks_mul_norelin(self.h, level, ct0[0].h, ct0[1].h, ct1[0].h, ct1[1].h,
                                            ctOut[0].h, ctOut[1].h, ctOut[2].h))
        else:
            check(lib().lr_ckks_mulrelin(self.h, level, ct0[0].h, ct0[1].h, ct1[0].h, ct1[1].h, evakey.h,
                                         ctOut[0].h, ctOut[1].h))

    def Rescale(self, ct):
        check(lib().lr_ckks_rescale(self.h, ct[0].h, ct[1].h))

    def EncryptPk(self, level, u, pk, e, plaintext, ctOut):
        """pkEncryptor.encrypt, the branch through the special primes (ckks/encryptor.go:205-234), after the sampling:
        u = SampleTernaryMontgomeryNTT over QP, e = the two Gaussian samples as residues over QP (coefficient domain,
        what SampleAndAdd adds, ring/gaussianSampler.go:254-274), pk = (pk0, pk1) over QP, plaintext over Q (NTT)."""
        check(lib().lr_ckks_encrypt_pk(self.h, level, u.h, pk[0].h, pk[1].h, e[0].h, e[1].h, plaintext.h, ctOut[0].h, ctOut[1].h))

    def Decrypt(self, level, ct, sk, ptOut):
        """decryptor.Decrypt (ckks/decryptor.go:53-78): Horner evaluation at the secret key (Montgomery NTT form)."""
        arr = (C.c_void_p * len(ct))(*[c.h.value for c in ct])
        check(lib().lr_ckks_decrypt(self.h, level, arr, len(ct) - 1, sk.h, ptOut.h))

    def PermuteNTT(self, level, ct0, gen, rotkey, ctOut):
        """evaluator.permuteNTT (ckks/evaluator.go:1448): RotateColumns with the key of that rotation, or Conjugate.
        gen: the Galois element whose PermuteNTTIndex the reference stores next to the key."""
        check(lib().lr_ckks_rotate(self.h, level, ct0[0].h, ct0[1].h, int(gen), rotkey.h, ctOut[0].h, ctOut[1].h))

    def RotateColumnsPow2(self, level, ct0, k, pow2_keys, ctOut):
        """evaluator.rotateColumnsPow2 (ckks/evaluator.go:1408-1430): rotation by k as the chain of the power-of-two
        rotations in its binary expansion.  pow2_keys: {2^i: (Galois element, SwitchingKey image)}."""
        if ctOut[0] is not ct0[0]:
            self.contextQ.CopyLvl(level, ct0[0], ctOut[0])     # :1417-1418
            self.contextQ.CopyLvl(level, ct0[1], ctOut[1])
        idx = 1
        while k > 0:
            if k & 1:
                gen, key = pow2_keys[idx]
                self.PermuteNTT(level, ctOut, gen, key, ctOut)  # :1424
            idx <<= 1
            k >>= 1

    def RotateHoisted(self, level, ct0, gens, rotkeys, ctOuts):
        """evaluator.RotateHoisted (ckks/evaluator.go:1252): ctOuts[r] = rotation of ct0 by the Galois element gens[r]."""
        n = len(gens)
        g = (C.c_uint64 * n)(*[int(x) for x in gens])
        keys = (C.c_void_p * n)(*[k.h.value for k in rotkeys])
        o0 = (C.c_void_p * n)(*[o[0].h.value for o in ctOuts])
        o1 = (C.c_void_p * n)(*[o[1].h.value for o in ctOuts])
        check(lib().lr_ckks_rotate_hoisted(self.h, level, ct0[0].h, ct0[1].h, n, g, keys, o0, o1))


class CkksBatcher:
    """Merges the MulRelin calls of concurrent evaluators -- the reference's model is one evaluator per goroutine, one ciphertext
    per call (examples/dbfv/psi/psi.go:215-233) -- into batched launches (lr_ckks_batcher_* in include/lattigo_ring.h).
    One lane = one CkksPlan over its own pair of contexts on its own stream; the batcher builds them."""

    def __init__(self, N, Q, P, max_batch=64, lanes=2, device=0):
        self.lanes = []
        for i in range(lanes):
            cq, cp = Context(N, Q, device=device), Context(N, P, device=device)
            self.lanes.append((cq, cp, CkksPlan(cq, cp, max_batch)))
        arr = (C.c_void_p * lanes)(*[ln[2].h for ln in self.lanes])
        h = C.c_void_p()
        check(lib().lr_ckks_batcher_create(arr, lanes, C.byref(h)))
        self.h = h
        self.max_batch = max_batch

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().lr_ckks_batcher_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def NewSwitchingKey(self):
        """the key image every calling evaluator passes (one handle: calls share a batch only over the same key)"""
        return self.lanes[0][2].NewSwitchingKey()

    def MulRelin(self, level, ct0, ct1, evakey, ctOut):
        """evaluator.MulRelin (ckks/evaluator.go:1016) of two ciphertexts with key; blocks until this call's result is complete.
        Call it from as many host threads as there are evaluators (ctypes releases the GIL for the call)."""
        check(lib().lr_ckks_batcher_mulrelin(self.h, level, ct0[0].h, ct0[1].h, ct1[0].h, ct1[1].h, evakey.h, ctOut[0].h, ctOut[1].h))

    def PermuteNTT(self, level, ct0, gen, rotkey, ctOut):
        """evaluator.permuteNTT (ckks/evaluator.go:1448) = RotateColumns with the key of that rotation / Conjugate; blocks until this
        call's result is complete; calls with the same (level, gen, key) in flight together share a launch"""
        check(lib().lr_ckks_batcher_rotate(self.h, level, ct0[0].h, ct0[1].h, C.c_uint64(int(gen)), rotkey.h, ctOut[0].h, ctOut[1].h))

    def Stats(self):
        b, p, l = C.c_uint64(), C.c_uint64(), C.c_int()
        check(lib().lr_ckks_batcher_stats(self.h, C.byref(b), C.byref(p), C.byref(l)))
        return {"batches": b.value, "products": p.value, "largest": l.value}


class BfvPlan:
    """What bfv.NewEvaluator builds around the ring for Mul (bfv/evaluator.go:89-112) and the tensorAndRescale
    call sequence (:278-464) for two degree-1 ciphertexts, device-resident."""

    def __init__(self, contextQ, contextQMul, t, max_batch=1, options=None):
        self.contextQ, self.contextQMul, self.t = contextQ, contextQMul, int(t)
        h = C.c_void_p()
        if options is None:
            check(lib().lr_bfv_plan_create(contextQ.h, contextQMul.h, self.t, max_batch, C.byref(h)))
        else:
            check(lib().lr_bfv_plan_create_ex(contextQ.h, contextQMul.h, self.t, max_batch, C.byref(options), C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().lr_bfv_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def Mul(self, ct0, ct1, ctOut):
        """ct0, ct1: pairs of Poly over Q (coefficient domain); ctOut: triple (degree 2)."""
        check(lib().lr_bfv_mul(self.h, ct0[0].h, ct0[1].h, ct1[0].h, ct1[1].h, ctOut[0].h, ctOut[1].h, ctOut[2].h))

    def NewRelinearizer(self, contextP, max_batch=1):
        """the key-switch half of bfv.NewEvaluator (decomposer, baseconverterQ1P, keyswitchpool: bfv/evaluator.go:100-112) over
        (contextQ, contextP): a CkksPlan, whose BfvRelinearize / BfvSwitchKeys / NewSwitchingKey serve Evaluator.Relinearize"""
        return CkksPlan(self.contextQ, contextP, max_batch)


class BfvBatcher:
    """Merges the Mul and Relinearize calls of concurrent BFV evaluators -- the reference's own pooled workload: every task of
    examples/dbfv/psi/psi.go:215-233 calls evaluator.Mul and evaluator.Relinearize on one ciphertext pair -- into batched launches
    (lr_bfv_batcher_* in include/lattigo_ring.h).  One lane = a BfvPlan over (contextQ, contextQMul) and the key-switch plan over
    (contextQ, contextP) on their own stream; the batcher builds them."""

    def __init__(self, N, Q, P, QMul, t, max_batch=64, lanes=2, device=0):
        self.lanes = []
        for i in range(lanes):
            cq, cp, cm = Context(N, Q, device=device), Context(N, P, device=device), Context(N, QMul, device=device)
            self.lanes.append((cq, cp, cm, BfvPlan(cq, cm, t, max_batch), CkksPlan(cq, cp, max_batch)))
        muls = (C.c_void_p * lanes)(*[ln[3].h for ln in self.lanes])
        kss = (C.c_void_p * lanes)(*[ln[4].h for ln in self.lanes])
        h = C.c_void_p()
        check(lib().lr_bfv_batcher_create(muls, kss, lanes, C.byref(h)))
        self.h = h
        self.max_batch = max_batch

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().lr_bfv_batcher_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def NewSwitchingKey(self):
        """the relinearisation key image every calling evaluator passes (one handle: calls share a batch only over the same key)"""
        return self.lanes[0][4].NewSwitchingKey()

    def Mul(self, ct0, ct1, ctOut):
        """evaluator.Mul (bfv/evaluator.go:467) of two degree-1 ciphertexts -> degree 2; blocks until this call's result is complete"""
        check(lib().lr_bfv_batcher_mul(self.h, ct0[0].h, ct0[1].h, ct1[0].h, ct1[1].h, ctOut[0].h, ctOut[1].h, ctOut[2].h))

    def Relinearize(self, ct, evakey, ctOut):
        """evaluator.Relinearize (bfv/evaluator.go:512) of a degree-2 ciphertext"""
        check(lib().lr_bfv_batcher_relinearize(self.h, ct[0].h, ct[1].h, ct[2].h, evakey.h, ctOut[0].h, ctOut[1].h))

    def Stats(self):
        b, p, l = C.c_uint64(), C.c_uint64(), C.c_int()
        check(lib().lr_bfv_batcher_stats(self.h, C.byref(b), C.byref(p), C.byref(l)))
        return {"batches": b.value, "products": p.value, "largest": l.value}

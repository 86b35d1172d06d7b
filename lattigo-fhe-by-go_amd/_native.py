"""ctypes binding of include/lattigo_ring.h (liblattigo_ring_hip.so).

There is no CPU fallback: if the shared library is missing or no HIP device is visible,
every arithmetic entry point raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"``
or ``lattigo-fhe-by-go_amd/csrc/build.sh``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblattigo_ring_hip.so")

u64 = C.c_uint64
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p
i32 = C.c_int

STATUS = {0: "LR_OK", 1: "LR_ERR_INVALID_DEGREE", 2: "LR_ERR_NOT_NTT_FRIENDLY", 3: "LR_ERR_SHAPE",
          4: "LR_ERR_ARG", 5: "LR_ERR_HIP", 6: "LR_ERR_UNSUPPORTED", 7: "LR_ERR_NOMEM", 8: "LR_ERR_INTERNAL"}

class Options(C.Structure):
    """lr_options (include/lattigo_ring.h), field for field.  Options() is lr_options_init's image: every switch at its default."""
    _fields_ = [("struct_size", C.c_uint32), ("version", C.c_uint32)] + [(n, C.c_int32) for n in (
        "no_asm", "no_fp", "ntt_mode", "asm_variant", "asm14_1024", "no_wide14_small", "wide14_max_items", "ntt_split15", "split15_max_workgroups",
        "no_invfuse", "no_grid_padding", "ntt_stagger", "ntt_persist", "ntt_timeline", "no_epilogue", "no_int_epilogue", "rescale_unfused",
        "rescale_unpaired", "pair_max_workgroups", "ext_narrow", "ext_ieee_div", "no_ext_chunks", "no_staging", "no_exttop", "no_invtop",
        "no_ext_group", "keymac_narrow", "no_pair", "no_fork", "fork_below_workgroups", "bfv_no_ext_epilogue", "bfv_no_gather")] + [("bfv_gather_below", C.c_int64)]

    def __init__(self, **fields):
        super().__init__()
        check(lib().lr_options_init(C.byref(self)))
        for k, v in fields.items():
            if k not in dict(self._fields_):
                raise AttributeError("lr_options has no field %r" % k)
            setattr(self, k, int(v))

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol declared in include/lattigo_ring.h: name -> argtypes
SYMBOLS = {
    "lr_last_error_string": [],
    "lr_device_count": [C.POINTER(i32)],
    "lr_build_info": [],
    "lr_options_init": [vp],
    "lr_context_get_options": [vp, vp],
    "lr_context_create": [u64, u64p, i32, i32, C.POINTER(vp)],
    "lr_context_create_ex": [u64, u64p, i32, i32, vp, C.POINTER(vp)],
    "lr_context_destroy": [vp],
    "lr_context_ntt_variants": [vp, C.POINTER(i32), C.POINTER(i32)],
    "lr_context_set_stream": [vp, vp],
    "lr_context_sync": [vp],
    "lr_context_info": [vp, u64p, C.POINTER(i32), C.POINTER(i32)],
    "lr_context_get_table": [vp, i32, u64p, C.c_size_t],
    "lr_poly_alloc": [vp, i32, i32, C.POINTER(vp)],
    "lr_poly_wrap": [vp, vp, i32, i32, C.POINTER(vp)],
    "lr_poly_wrap_strided": [vp, vp, i32, i32, C.c_longlong, C.POINTER(vp)],
    "lr_poly_free": [vp],
    "lr_poly_info": [vp, u64p, C.POINTER(i32), C.POINTER(i32), C.POINTER(vp)],
    "lr_poly_upload": [vp, i32, C.POINTER(u64p), i32],
    "lr_poly_download": [vp, i32, C.POINTER(u64p), i32],
    "lr_poly_upload_limb": [vp, i32, i32, vp],
    "lr_poly_download_limb": [vp, i32, i32, vp],
    "lr_ntt_host_limb": [vp, i32, i32, vp, vp],
    "lr_poly_unmarshal": [vp, i32, C.c_char_p, C.c_size_t],
    "lr_poly_marshal": [vp, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)],
    "lr_poly_upload_dense": [vp, vp, C.c_size_t],
    "lr_poly_download_dense": [vp, vp, C.c_size_t],
    "lr_poly_zero": [vp],
    "lr_poly_set_limbs": [vp, i32],
    "lr_ntt": [vp, i32, vp, vp],
    "lr_intt": [vp, i32, vp, vp],
    "lr_ntt_limb": [vp, i32, vp, i32, vp, i32],
    "lr_intt_limb": [vp, i32, vp, i32, vp, i32],
    "lr_ntt_host": [vp, i32, C.POINTER(u64p), C.POINTER(u64p)],
    "lr_intt_host": [vp, i32, C.POINTER(u64p), C.POINTER(u64p)],
    "lr_ewise": [vp, i32, i32, vp, vp, vp, u64p],
    "lr_half_scalar_op": [vp, i32, i32, vp, vp, vp, vp],
    "lr_permute_ntt": [vp, i32, vp, u64, vp],
    "lr_permute_ntt_index": [u64, u64, u64, u64p],
    "lr_permute": [vp, vp, u64, vp],
    "lr_mult_by_monomial": [vp, vp, u64, vp],
    "lr_shift": [vp, vp, u64, vp],
    "lr_rotate": [vp, vp, u64],
    "lr_simple_scaler_create": [vp, u64, C.POINTER(vp)],
    "lr_simple_scaler_destroy": [vp],
    "lr_simple_scaler_tables": [vp, u64p, C.POINTER(C.c_double), i32],
    "lr_simple_scale": [vp, vp, vp],
    "lr_bext_create": [vp, vp, C.POINTER(vp)],
    "lr_bext_destroy": [vp],
    "lr_modup_split_qp": [vp, i32, vp, vp],
    "lr_modup_split_pq": [vp, i32, vp, vp],
    "lr_moddown_ntt_pq": [vp, i32, vp, vp],
    "lr_moddown_split_ntt_pq": [vp, i32, vp, vp, vp],
    "lr_moddown_pq": [vp, i32, vp, vp],
    "lr_moddown_split_pq": [vp, i32, vp, vp, vp],
    "lr_moddown_split_qp": [vp, i32, i32, vp, vp, vp],
    "lr_bext_get_table": [vp, i32, u64p, C.c_size_t],
    "lr_decomposer_create": [vp, vp, C.POINTER(vp)],
    "lr_decomposer_destroy": [vp],
    "lr_decompose": [vp, i32, i32, vp, vp],
    "lr_decompose_and_split": [vp, i32, i32, vp, vp, vp],
    "lr_div_floor_by_last_modulus_ntt": [vp, vp],
    "lr_div_floor_by_last_modulus": [vp, vp],
    "lr_div_round_by_last_modulus_ntt": [vp, vp],
    "lr_div_round_by_last_modulus": [vp, vp],
    "lr_div_floor_by_last_modulus_many": [vp, vp, i32, i32],
    "lr_div_round_by_last_modulus_many": [vp, vp, i32, i32],
    "lr_ckks_plan_create": [vp, vp, i32, C.POINTER(vp)],
    "lr_ckks_plan_create_ex": [vp, vp, i32, vp, C.POINTER(vp)],
    "lr_ckks_plan_destroy": [vp],
    "lr_ckks_plan_stats": [vp, vp, vp],
    "lr_ckks_switch_keys": [vp, i32, vp, vp, vp, vp],
    "lr_bfv_switch_keys": [vp, vp, vp, vp, vp],
    "lr_bfv_relinearize": [vp, vp, vp, vp, vp, vp, vp],
    "lr_bfv_rotate": [vp, vp, vp, u64, vp, vp, vp],
    "lr_ckks_mulrelin": [vp, i32, vp, vp, vp, vp, vp, vp, vp],
    "lr_ckks_batcher_create": [vp, i32, vp],
    "lr_ckks_batcher_destroy": [vp],
    "lr_ckks_batcher_mulrelin": [vp, i32, vp, vp, vp, vp, vp, vp, vp],
    "lr_ckks_batcher_stats": [vp, vp, vp, vp],
    "lr_ckks_batcher_rotate": [vp, i32, vp, vp, C.c_uint64, vp, vp, vp],
    "lr_ckks_rescale": [vp, vp, vp],
    "lr_ckks_mul_norelin": [vp, i32, vp, vp, vp, vp, vp, vp, vp],
    "lr_ckks_mul_plain": [vp, i32, vp, vp, vp, vp, vp],
    "lr_ckks_encrypt_pk": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "lr_ckks_decrypt": [vp, i32, C.POINTER(vp), i32, vp, vp],
    "lr_context_last_ntt_kernel": [vp, C.c_char_p, C.c_size_t],
    "lr_context_timeline": [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)],
    "lr_selftest_division": [vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)],
    "lr_ckks_rotate": [vp, i32, vp, vp, u64, vp, vp, vp],
    "lr_ckks_rotate_hoisted": [vp, i32, vp, vp, i32, u64p, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)],
    "lr_bfv_plan_create": [vp, vp, u64, i32, C.POINTER(vp)],
    "lr_bfv_plan_create_ex": [vp, vp, u64, i32, vp, C.POINTER(vp)],
    "lr_bfv_plan_destroy": [vp],
    "lr_bfv_mul": [vp, vp, vp, vp, vp, vp, vp, vp],
    "lr_bfv_batcher_create": [vp, vp, i32, vp],
    "lr_bfv_batcher_destroy": [vp],
    "lr_bfv_batcher_mul": [vp, vp, vp, vp, vp, vp, vp, vp],
    "lr_bfv_batcher_relinearize": [vp, vp, vp, vp, vp, vp, vp],
    "lr_bfv_batcher_stats": [vp, vp, vp, vp],
    "lr_poly_copy_peer": [vp, vp, i32, vp, vp, i32, i32],
    "lr_context_wait_peer_copies": [vp],
    "lr_gather_blocks": [vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), i32],
    "lr_timer_start": [vp],
    "lr_timer_stop": [vp, C.POINTER(C.c_float)],
}

_lib = None


class LatticeRingError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (STATUS.get(code, code), msg))
        self.code = code


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("liblattigo_ring_hip.so is not built (%s); there is no CPU fallback -- run "
                           "__graft_entry__.build()" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, args in SYMBOLS.items():
        f = getattr(L, name)  # AttributeError if the header and the library disagree
        f.argtypes = args
        f.restype = C.c_char_p if name in ("lr_last_error_string", "lr_build_info") else i32
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = lib().lr_last_error_string()
        raise LatticeRingError(rc, msg.decode() if msg else "")
    return rc


def device_count():
    n = i32(0)
    try:
        check(lib().lr_device_count(C.byref(n)))
    except LatticeRingError:
        return 0
    return n.value

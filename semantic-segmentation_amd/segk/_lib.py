"""ctypes binding of libsegk.so (the C ABI declared in include/segk.h).

There is no fallback: if the shared library is missing or a call fails, the
caller gets an exception.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C semantic-segmentation_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsegk.so")

F32, BF16 = 0, 1
PM, NCHW = 0, 1
MAX_SEG = 4

c_int, c_float, c_void_p, c_ll = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_longlong


class ConvSeg(ctypes.Structure):
    _fields_ = [("inp", c_void_p), ("w", c_void_p), ("in_ld", c_int), ("Ck", c_int), ("R", c_int), ("dil", c_int),
                ("w_row_stride", c_ll), ("w_tap_stride", c_ll)]


class ConvArgs(ctypes.Structure):
    _fields_ = [("N", c_int), ("H", c_int), ("W", c_int), ("dtype", c_int), ("Cout", c_int), ("nseg", c_int),
                ("seg", ConvSeg * MAX_SEG), ("out", c_void_p), ("out_ld", c_int), ("out_dtype", c_int),
                ("bias", c_void_p), ("img_bias", c_void_p), ("stat_partials", c_void_p), ("batch", c_int),
                ("w_batch", c_ll), ("out_batch", c_ll), ("dil_step", c_int), ("addend", c_void_p),
                ("addend_ld", c_int), ("addend_bits", c_void_p), ("addend_bits_ld", c_int),
                ("bst_rows", c_void_p), ("bst_x", c_void_p), ("bst_x_ld", c_int), ("bst_mean", c_void_p),
                ("bst_inv_std", c_void_p), ("bst_scale", c_void_p), ("bst_shift", c_void_p), ("bst_bits", c_void_p),
                ("bst_bits_ld", c_int), ("bst_relu", c_int)]


class WgradArgs(ctypes.Structure):
    _fields_ = [("N", c_int), ("H", c_int), ("W", c_int), ("dtype", c_int), ("inp", c_void_p), ("in_ld", c_int),
                ("Cin", c_int), ("dout", c_void_p), ("dout_ld", c_int), ("Cout", c_int), ("R", c_int),
                ("dil", c_int), ("splits", c_int), ("slabs", c_void_p), ("tile", c_int)]


WGRAD_GROUP_MAX = 8


class WgradGroupArgs(ctypes.Structure):
    _fields_ = [("conv", WgradArgs), ("groups", c_int), ("inp", c_void_p * WGRAD_GROUP_MAX),
                ("dout", c_void_p * WGRAD_GROUP_MAX)]


class GemmTerm(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("a_batch", c_ll), ("lda", c_int), ("B", c_void_p), ("b_batch", c_ll), ("ldb", c_int),
                ("K", c_int)]


class GemmArgs(ctypes.Structure):
    _fields_ = [("batch", c_int), ("M", c_int), ("N", c_int), ("dtype", c_int), ("nterms", c_int),
                ("term", GemmTerm * MAX_SEG), ("C", c_void_p), ("c_batch", c_ll), ("ldc", c_int), ("c_dtype", c_int)]


class GemmTnArgs(ctypes.Structure):
    _fields_ = [("batch", c_int), ("M", c_int), ("N", c_int), ("K", c_int), ("dtype", c_int), ("A", c_void_p),
                ("a_batch", c_ll), ("lda", c_int), ("B", c_void_p), ("b_batch", c_ll), ("ldb", c_int),
                ("splits", c_int), ("slabs", c_void_p)]


# name -> argument types (every function returns int unless listed in _RESTYPES)
SIGNATURES = {
    "segk_version": [],
    "segk_last_error": [],
    "segk_device_count": [],
    "segk_nchw_to_pm": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "segk_pm_to_nchw": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "segk_pack_weight": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "segk_pack_weight_strided": [c_void_p, c_int, c_ll, c_ll, c_ll, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "segk_pack_weight_blocks": [c_int, c_int, c_int, c_ll, c_int],
    "segk_pack_weights": [c_void_p, c_int, c_int, c_void_p],
    "segk_conv_fwd": [ctypes.POINTER(ConvArgs), c_void_p],
    "segk_conv_stat_rows": [ctypes.POINTER(ConvArgs)],
    "segk_conv_bstat_rows": [ctypes.POINTER(ConvArgs)],
    "segk_conv_wgrad": [ctypes.POINTER(WgradArgs), c_void_p],
    "segk_wgrad_finish": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p],
    "segk_wgrad_finish_strided": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_ll, c_ll, c_ll, c_int, c_int,
                                  c_void_p],
    "segk_conv_wgrad_group": [ctypes.POINTER(WgradGroupArgs), c_void_p],
    "segk_wgrad_finish_group": [c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p), c_int, c_ll, c_ll,
                                c_ll, c_int, c_int, c_void_p],
    "segk_gemm_nt": [ctypes.POINTER(GemmArgs), c_void_p],
    "segk_gemm_tn": [ctypes.POINTER(GemmTnArgs), c_void_p],
    "segk_gemm_tn_finish": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_ll, c_int, c_int, c_int, c_void_p],
    "segk_softmax_rows": [c_void_p, c_ll, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "segk_softmax_rows_bwd": [c_void_p, c_int, c_int, c_void_p, c_int, c_ll, c_int, c_float, c_void_p, c_int, c_int,
                              c_int, c_void_p],
    "segk_transpose": [c_void_p, c_int, c_int, c_int, c_int, c_ll, c_int, c_void_p, c_ll, c_int, c_int, c_void_p],
    "segk_lincomb": [c_void_p, c_int, c_int, c_ll, c_int, c_void_p, c_int, c_float, c_void_p, c_void_p, c_int, c_float,
                     c_void_p, c_void_p, c_int, c_float, c_void_p, c_int, c_void_p],
    "segk_dot": [c_void_p, c_int, c_void_p, c_int, c_int, c_ll, c_int, c_void_p, c_void_p, c_void_p],
    "segk_dot_partial_rows": [c_ll],
    "segk_bn_stats": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "segk_bn_stat_rows": [c_int, c_int, c_int, c_int],
    "segk_bn_reduce": [c_void_p, c_int, c_int, ctypes.c_double, c_void_p, c_void_p, c_void_p],
    "segk_bn_coeffs": [c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p,
                       c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "segk_bn_reduce_coeffs": [c_void_p, c_int, c_int, ctypes.c_double, c_void_p, c_void_p, c_float, c_float, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "segk_bn_coeffs_eval": [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_int, c_void_p],
    "segk_bn_apply": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                      c_void_p, c_int, c_void_p, c_float, c_void_p, c_int, c_void_p, c_int, c_void_p],
    "segk_bn_bwd_stats": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                          c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p,
                          c_void_p],
    "segk_bn_bwd_apply": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                          c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float,
                          c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p],
    "segk_pool_sum": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p],
    "segk_upsample_fwd": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                          c_int, c_void_p],
    "segk_upsample_bwd": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int,
                          c_int, c_int, c_void_p, c_void_p],
    "segk_upsample_bwd_workspace": [c_int, c_int, c_int, c_int, c_int],
    "segk_ce_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                    c_void_p],
    "segk_ce_partial_rows": [c_ll],
    "segk_ce_fwd_static": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "segk_ce_fwd_static_max_pixels": [],
    "segk_ce_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                    c_void_p],
    "segk_argmax": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "segk_confmat": [c_void_p, c_int, c_void_p, c_ll, c_int, c_void_p, c_void_p],
    "segk_upsample_ce_fwd": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_void_p, c_void_p, c_void_p, c_void_p],
    "segk_upsample_ce_bwd": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                             c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "segk_upsample_argmax": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "segk_gather_f32": [c_void_p, c_int, c_void_p],
    "segk_comm_unique_id": [c_void_p],
    "segk_comm_init": [c_void_p, c_int, c_int],
    "segk_comm_destroy": [],
    "segk_allreduce_f32": [c_void_p, c_ll, c_void_p],
    "segk_allreduce_f64": [c_void_p, c_ll, c_void_p],
    "segk_allreduce_i64": [c_void_p, c_ll, c_void_p],
    "segk_mailbox_create": [c_int, c_void_p],
    "segk_mailbox_connect": [c_int, c_int, c_void_p, c_int],
    "segk_mailbox_allreduce_f64": [c_void_p, c_ll, c_void_p],
    "segk_mailbox_allreduce_f32": [c_void_p, c_ll, c_void_p],
    "segk_mailbox_status": [],
    "segk_maxpool3x3s2_fwd": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p],
    "segk_maxpool3x3s2_bwd": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p],
    "segk_mailbox_set_timeout": [c_int],
    "segk_dropout_mask": [c_void_p, c_ll, c_float, c_void_p, c_void_p],
    "segk_mailbox_destroy": [],
    "segk_sgd_step": [c_void_p, c_int, c_float, c_float, c_float, c_void_p],
    "segk_sgd_step_dev": [c_void_p, c_int, c_void_p, c_void_p],
}
_RESTYPES = {"segk_last_error": ctypes.c_char_p, "segk_upsample_bwd_workspace": c_ll,
             "segk_ce_fwd_static_max_pixels": c_ll, "segk_mailbox_status": c_ll}

_lib = None


class SegkError(RuntimeError):
    pass


def load():
    """Load libsegk.so once and attach the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SegkError("libsegk.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `make -C semantic-segmentation_amd/csrc`" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an entry point; a negative status becomes SegkError with the library's message."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc < 0:
        msg = lib.segk_last_error()
        raise SegkError("%s failed (%d): %s" % (name, rc, msg.decode("utf-8", "replace") if msg else ""))
    return rc

"""ctypes binding of liblstm_ctc_hip.so (C ABI: include/lstm_ctc_hip.h).

The library is built in-tree by ``make -C lstm_ctc_amd/csrc`` (see ``__graft_entry__.build``).
Missing library ⇒ ImportError-like RuntimeError: the product path never falls back to CPU code.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblstm_ctc_hip.so")

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
c_u32 = ctypes.c_uint32

# name -> (restype, argtypes); mirrors include/lstm_ctc_hip.h one to one
SIGNATURES = {
    "lc_last_error": (ctypes.c_char_p, []),
    "lc_version": (c_int, []),
    "lc_set_option": (c_int, [ctypes.c_char_p, ctypes.c_long]),
    "lc_get_option": (c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)]),
    "lc_ctc_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "lc_ctc_loss": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                            c_void_p, c_void_p, c_size_t, c_void_p]),
    "lc_ctc_greedy": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lc_edit_distance_host": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lc_gemm_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "lc_gemm_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int,
                            c_float, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lc_gemm_bf16": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int,
                             c_float, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lc_bn_workspace_bytes": (c_size_t, [c_int]),
    "lc_bn_moments": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lc_bn_apply": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                            c_int, c_void_p]),
    "lc_bn_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int,
                          c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lc_bn_update_moving": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "lc_cast_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "lc_gemm_bf16_nt": (c_int, [c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_int,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "lc_gemm_bf16_tn": (c_int, [c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_int,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "lc_gemm_f32_nt2": (c_int, [c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                c_float, c_void_p, c_int, c_void_p, c_void_p]),
    "lc_gemm_bf16_nt2": (c_int, [c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                 c_float, c_void_p, c_int, c_void_p, c_void_p]),
    "lc_split_bf16x3": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "lc_gemm_bf16x3_nt": (c_int, [c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_int,
                                  c_void_p, c_void_p]),
    "lc_gemm_bf16x3_tn_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "lc_gemm_bf16x3_tn_workspace_bytes_ld": (ctypes.c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "lc_gemm_bf16x3_tn": (c_int, [c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_int,
                                  c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    "lc_gemm_next_epilogue": (c_int, [c_void_p]),
    "lc_gemm_bf16_nn": (c_int, [c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_int,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "lc_debug_set_lstm_stamps": (None, [c_void_p]),
    "lc_debug_set_ctc_stamps": (None, [c_void_p]),
    "lc_debug_spin": (c_int, [c_int, c_int, c_int, c_void_p]),
    "lc_length_mask": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "lc_lstm_fwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "lc_lstm_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "lc_lstm_fwd_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "lc_lstm_fwd_x3": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "lc_lstm_bwd_x3": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "lc_lstm_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "lc_lstm_bwd_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "lc_lstm_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "lc_dropout_scale": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_u32, c_u32, c_void_p, c_int, c_int,
                                 c_void_p]),
    "lc_dropout_scale_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_u32, c_u32, c_void_p, c_int, c_int,
                                      c_void_p, c_int, c_void_p]),
    "lc_moe_combine_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_u32, c_void_p,
                                   c_void_p, c_void_p]),
    "lc_moe_combine_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_u32,
                                   c_void_p, c_void_p]),
    "lc_optimizer_workspace_bytes": (c_size_t, [c_size_t]),
    "lc_optimizer_step": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_float, c_float, c_int, c_float,
                                  c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lc_colsum_workspace_bytes": (c_size_t, [c_int]),
    "lc_colsum": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "lc_debug_last_lstm_schedule": (c_int, []),
    "lc_debug_gemm_whole_round_row_tiles": (c_int, [c_int, c_int, c_int]),
    "lc_transpose": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "lc_label_smoothing": (c_int, [c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "lc_tfrecord_inspect": (c_int, [c_void_p, c_size_t, c_int, c_void_p]),
    "lc_tfrecord_decode": (c_int, [c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_void_p, c_size_t, ctypes.c_int64,
                                   c_void_p, ctypes.c_int64]),
    "lc_crc32c": (c_u32, [c_void_p, c_size_t]),
    "lc_batch_open": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "lc_batch_decode": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_size_t, ctypes.c_int64, c_void_p,
                                c_size_t, ctypes.c_int64, ctypes.c_int64, c_int]),
    "lc_batch_close": (None, [c_void_p]),
    "lc_posteriors": (c_int, [c_void_p, c_int, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p]),
}


class GemmEpilogue(ctypes.Structure):
    """lc_gemm_epilogue_t (include/lstm_ctc_hip.h)."""
    _fields_ = [("keep", ctypes.c_float), ("seed", ctypes.c_uint32), ("stream0", ctypes.c_uint32),
                ("drop_width", ctypes.c_int), ("c_bf16", ctypes.c_void_p), ("ldc_bf16", ctypes.c_int),
                ("shadow_only", ctypes.c_int)]


class SeqExInfo(ctypes.Structure):
    """lc_seqex_info_t"""
    _fields_ = [("num_frames", ctypes.c_int64), ("dim", ctypes.c_int32), ("has_input", ctypes.c_int32),
                ("has_target", ctypes.c_int32), ("num_labels", ctypes.c_int64)]


class LstmFwdDir(ctypes.Structure):
    """lc_lstm_fwd_dir_t"""
    _fields_ = [("zx", c_void_p), ("R", c_void_p), ("w_f", c_void_p), ("w_i", c_void_p), ("w_o", c_void_p),
                ("cs", c_void_p), ("hs", c_void_p), ("reverse", c_int), ("hs_bf16", c_void_p), ("shadow_only", c_int)]


class LstmBwdDir(ctypes.Structure):
    """lc_lstm_bwd_dir_t"""
    _fields_ = [("gates", c_void_p), ("RT", c_void_p), ("w_f", c_void_p), ("w_i", c_void_p), ("w_o", c_void_p),
                ("cs", c_void_p), ("dh", c_void_p), ("dpeep", c_void_p), ("dbias", c_void_p), ("reverse", c_int),
                ("dz_bf16", c_void_p), ("shadow_only", c_int)]


OPTION_UNSET = -0x7fffffff - 1  # LC_OPTION_UNSET
LSTM_STATUS_OFFSET = 64        # LC_LSTM_STATUS_OFFSET: sticky status word inside the LSTM workspace

_lib = None


class LibraryError(RuntimeError):
    pass


def load():
    """Load the HIP library, declaring every prototype.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryError(
            "liblstm_ctc_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C lstm_ctc_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    try:
        # PyTorch-ROCm ships its own HIP runtime; whichever copy of libamdhip64 is loaded first serves the process.  Load
        # torch's BEFORE this library pulls in the system one, or tensors and kernels end up on two runtimes ("no
        # ROCm-capable device is detected" on the first launch - seen when build() and smoke() ran in one process).
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().lc_last_error().decode("utf-8", "replace")
        raise LibraryError("%s failed (rc=%d): %s" % (what or "liblstm_ctc_hip call", rc, msg))

"""ctypes binding of liblocator_hip.so (C ABI: include/locator_hip.h).

PyTorch is used only as plumbing (device memory, streams, graphs); every kernel on the hot
path comes from this library.  A missing library is a hard error — there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import os

# torch ships its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It must be the
# one already resident when liblocator_hip.so is dlopen'ed, otherwise the library binds /opt/rocm's
# copy and the process ends up with two HIP runtimes that do not share devices, streams or memory.
import torch  # noqa: F401  (side effect: loads torch's libamdhip64 first)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblocator_hip.so")


def use_library(path):
    """Measurement tools only (tools/rows_gemm_bench.py --lib): bind another build of the library, e.g. a timing
    ablation from `make -C locator_amd/csrc ablate A=1`.  Must be called before load()."""
    global LIB_PATH
    assert _lib is None, "library already loaded"
    LIB_PATH = path

c_i32p = C.POINTER(C.c_int32)
vp = C.c_void_p


class Dims(C.Structure):
    _fields_ = [("K", C.c_int), ("Kp", C.c_int), ("H", C.c_int), ("Hp", C.c_int), ("L", C.c_int),
                ("n_pre", C.c_int)]


class Layout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("w1", "gamma", "beta", "b1", "wh", "bh", "wa", "ba", "wb", "bb",
                                         "n_trainable", "mov_mean", "mov_var", "n_total")]


class Tuning(C.Structure):
    _fields_ = [("stack_helpers", C.c_int), ("stack_xcd_stride", C.c_int), ("l1b_nt_mask", C.c_int),
                ("l1b_rows", C.c_int), ("rows_rt", C.c_int), ("gemm_i8_unit_tiles", C.c_int), ("stack_rows", C.c_int), ("gemm_reduce", C.c_int), ("chain_tail", C.c_int),
                ("stack_train_rows", C.c_int)]


class Net(C.Structure):
    _fields_ = [("d", Dims), ("params", vp), ("adam_m", vp), ("adam_v", vp), ("alpha_tab", vp),
                ("alpha_tab_len", C.c_int), ("lr", vp), ("t_base", vp), ("X", vp), ("x_pitch", C.c_int64),
                ("Y", vp), ("drop_p", C.c_float), ("wht", vp), ("ws", vp), ("ws_predict", vp), ("l1_fwd_grid", C.c_int), ("l1_bwd_grid", C.c_int),
                ("slot_rows", C.c_int), ("predict_pieces", C.c_int), ("l1_image", vp), ("l1_image_bytes", C.c_int64), ("x_max", C.c_int), ("predict_digits", C.c_int),
                ("l1_image_ready", C.c_int), ("X2", vp), ("x2_pitch", C.c_int64), ("l1_scan_ready", C.c_int), ("tune", Tuning)]


class CbState(C.Structure):
    _fields_ = [("ck_best", C.c_double), ("es_best", C.c_double), ("rl_best", C.c_double), ("lr", C.c_float),
                ("lr_factor", C.c_float), ("es_wait", C.c_int), ("rl_wait", C.c_int), ("patience", C.c_int),
                ("lr_patience", C.c_int), ("epoch", C.c_int), ("stopped", C.c_int), ("stop_epoch", C.c_int),
                ("best_epoch", C.c_int), ("save_now", C.c_int), ("reserved", C.c_int)]


# name -> (restype, argtypes); mirrors include/locator_hip.h one to one
SIGNATURES = {
    "loc_last_error": (C.c_char_p, []),
    "loc_version": (C.c_int, []),
    "loc_make_dims": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(Dims)]),
    "loc_param_layout": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout)]),
    "loc_w1s_index": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "loc_workspace_floats": (C.c_int64, [C.POINTER(Dims)]),
    "loc_workspace_floats_batch": (C.c_int64, [C.POINTER(Dims), C.c_int]),
    "loc_init_glorot": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, vp]),
    "loc_init_uniform": (C.c_int, [vp, C.c_int64, C.c_float, C.c_uint64, C.c_uint64, vp]),
    "loc_dropout_mask_fill": (C.c_int, [vp, C.c_int64, C.c_float, C.c_uint64, C.c_uint64, vp]),
    "loc_gather_columns": (C.c_int, [vp, C.c_int64, vp, C.c_int, vp, C.c_int64, C.c_int, vp]),
    "loc_kde_peak_batch": (C.c_int, [vp, vp, C.c_int, C.c_double, vp, vp, vp]),
    "loc_w1_swizzle": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]),
    "loc_w1_unswizzle": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]),
    "loc_bn_batch_stats": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "loc_bn_infer_scale_shift": (C.c_int, [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "loc_l1_forward": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.POINTER(Dims), vp, vp, vp, vp, C.c_int, vp, vp, vp,
                                 C.c_float, vp]),
    "loc_l1_forward_in_dropout": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.POINTER(Dims), vp, vp, vp, vp, C.c_int, vp, vp,
                                            C.c_float, vp]),
    "loc_l1_backward_adam_in_dropout": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.POINTER(Dims), vp, vp, vp, vp, vp, vp, vp,
                                                  vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int,
                                                  vp, vp, C.POINTER(Tuning), vp, C.c_float, vp]),
    "loc_l1_rows_supported": (C.c_int, [C.c_int, C.c_int]),
    "loc_l1_forward_rows": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.POINTER(Dims), vp, vp, vp, vp, C.c_int64, vp,
                                      C.c_int, C.c_int, C.POINTER(Tuning), vp]),
    "loc_l1_gemm_supported": (C.c_int, [C.c_int, C.c_int]),
    "loc_l1_image_bytes": (C.c_int64, [C.POINTER(Dims), C.c_int]),
    "loc_l1_image_build": (C.c_int, [C.POINTER(Dims), vp, vp, C.c_int, vp, vp]),
    "loc_l1_forward_gemm": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.POINTER(Dims), vp, C.c_int, vp, vp, C.c_int64, vp,
                                      C.c_int, vp]),
    "loc_l1_gemm_i8_supported": (C.c_int, [C.c_int, C.c_int]),
    "loc_l1_image_i8_bytes": (C.c_int64, [C.POINTER(Dims), C.c_int]),
    "loc_l1_image_i8_build": (C.c_int, [C.POINTER(Dims), vp, vp, C.c_int, vp, vp]),
    "loc_l1_quant_scan": (C.c_int, [C.POINTER(Dims), vp, vp, vp, vp]),
    "loc_l1_image_i8_guard_offset": (C.c_int64, []),
    "loc_l1_image_i8_tiles_offset": (C.c_int64, [C.POINTER(Dims)]),
    "loc_l1_image_i8_build_scanned": (C.c_int, [C.POINTER(Dims), vp, vp, C.c_int, vp, vp]),
    "loc_predict_scan": (C.c_int, [C.POINTER(Net), vp]),
    "loc_l1_forward_gemm_i8_partial": (C.c_int, [vp, C.c_int64, C.c_int, vp, C.c_int, C.POINTER(Dims), vp, C.c_int, C.c_int, vp,
                                                 C.c_int64, C.c_int, C.POINTER(Tuning), C.POINTER(C.c_int), C.POINTER(vp), vp]),
    "loc_stack_forward_eval_partial": (C.c_int, [vp, C.c_int, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int,
                                                 C.c_int, vp, vp, vp, vp, C.c_int, vp]),
    "loc_stack_rows_supported": (C.c_int, [C.c_int, C.c_int]),
    "loc_stack_rows_min_rows": (C.c_int, []),
    "loc_stack_forward_eval_form": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp]),
    "loc_l1_forward_gemm_i8": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.POINTER(Dims), vp, C.c_int, C.c_int, vp, vp,
                                         C.c_int64, vp, C.c_int, C.POINTER(Tuning), vp]),
    "loc_genotype_max": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, vp, vp]),
    "loc_l1_backward_adam_main": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.POINTER(Dims), vp, vp, vp, vp, vp, vp, vp,
                                            vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.POINTER(Tuning), vp]),
    "loc_l1_backward_adam": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.POINTER(Dims), vp, vp, vp, vp, vp, vp, vp, vp,
                                       vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp,
                                       vp, C.POINTER(Tuning), vp]),
    "loc_bn_epoch_stats": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp,
                                     vp, vp, vp]),
    "loc_bn_epoch_stats_only": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "loc_bn_epoch_finish": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "loc_workspace_bn4": (vp, [C.POINTER(Net)]),
    "loc_event_create_notiming": (C.c_int, [C.POINTER(vp)]),
    "loc_dense_forward": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp, C.c_float, vp]),
    "loc_dense_backward": (C.c_int, [vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp,
                                     C.c_int, vp, vp, C.c_int, vp]),
    "loc_head_train": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int64, C.c_int64,
                                 C.c_int64, C.c_int64, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp]),
    "loc_head_eval": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "loc_stack_fused_supported": (C.c_int, [C.c_int]),
    "loc_transpose_hidden": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "loc_stack_forward_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(Tuning), vp]),
    "loc_stack_forward_eval": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "loc_stack_dw_adam": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                    C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, vp, vp,
                                    C.c_int, vp, vp, C.c_int, vp]),
    "loc_stack_dw_adam_tail": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                         C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, vp, vp,
                                         C.c_int, vp, vp, C.c_int, vp, vp]),
    "loc_train_step": (C.c_int, [C.POINTER(Net), vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp]),
    "loc_train_chain_supported": (C.c_int, [C.POINTER(Net)]),
    "loc_train_step_chain": (C.c_int, [C.POINTER(Net), vp, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "loc_l1_chain_supported": (C.c_int, [C.c_int]),
    "loc_l1_chain_groups_per_workgroup": (C.c_int, [C.c_int]),
    "loc_l1_backward_adam_chain": (C.c_int, [vp, C.c_int64, vp, C.c_int, vp, C.c_int, C.POINTER(Dims), vp, vp, vp] + [vp] * 12
                                   + [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_int64, C.POINTER(Tuning), vp]),
    "loc_pack_genotypes_2bit": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, vp, C.c_int64, vp]),
    "loc_l1_forward_gemm_i8_packed": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.POINTER(Dims), vp, C.c_int, vp, vp, C.c_int64,
                                                vp, C.c_int, C.POINTER(Tuning), vp]),
    "loc_predict": (C.c_int, [C.POINTER(Net), vp, C.c_int, vp, C.c_int, vp, vp]),
    "loc_predict_image_mode": (C.c_int, [C.POINTER(Net), C.c_int]),
    "loc_filter_snps_flags": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "loc_filter_snps_rows": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int64, vp]),
    "loc_epoch_callbacks": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]),
    "loc_snapshot_if": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
    "loc_event_create": (C.c_int, [C.POINTER(vp)]),
    "loc_event_destroy": (C.c_int, [vp]),
    "loc_event_record": (C.c_int, [vp, vp]),
    "loc_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
}


class LocatorHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load liblocator_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LocatorHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C locator_amd/csrc`.  locator_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().loc_last_error().decode(errors="replace")
        raise LocatorHipError(f"{what} failed (code {rc}): {msg}")


def make_dims(K, H, L):
    d = Dims()
    check(load().loc_make_dims(int(K), int(H), int(L), C.byref(d)), "loc_make_dims")
    return d


def param_layout(d):
    lay = Layout()
    check(load().loc_param_layout(C.byref(d), C.byref(lay)), "loc_param_layout")
    return lay

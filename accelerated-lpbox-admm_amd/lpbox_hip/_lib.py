"""ctypes loader for liblpbox_hip.so (the C-ABI declared in include/lpbox_hip.h).

There is no CPU fallback: if the library is missing this module raises, and every compute call fails
with LPBOX_E_NODEVICE when no HIP device is present.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# LPBOX_LIB_VARIANT=stamps selects the diagnostic build with in-kernel phase stamps (make -C csrc stamps)
_VARIANT = os.environ.get("LPBOX_LIB_VARIANT", "")
LIB_PATH = os.path.join(HERE, "liblpbox_hip%s.so" % (("_" + _VARIANT) if _VARIANT else ""))

FLAVOUR_LP = 0
FLAVOUR_SEG = 1

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")

# every symbol include/lpbox_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "lpbox_version": (C.c_char_p, []),
    "lpbox_last_error": (C.c_char_p, []),
    "lpbox_device_count": (C.c_int, []),
    "lpbox_set_device": (C.c_int, [C.c_int]),
    "lpbox_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int]),
    "lpbox_destroy": (None, [C.c_void_p]),
    "lpbox_set_problem_lp": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, C.c_void_p, _dp, C.c_void_p]),
    "lpbox_read_files_lp": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int]),
    "lpbox_read_file": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "lpbox_get_problem_lp": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "lpbox_init": (C.c_int, [C.c_void_p]),
    "lpbox_iterate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "lpbox_iterate_l2f": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]),
    "lpbox_get_n": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_get_org_n": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_get_l": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_get_iter": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_get_x_iters": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "lpbox_set_record": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_set_x_update": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_get_direct_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "lpbox_seg_get_x_history": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "lpbox_policy_layout": (C.c_int, [C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "lpbox_policy_encode_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lpbox_policy_f32frag_layout": (C.c_int, [C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "lpbox_policy_encode_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lpbox_policy_f32_layout": (C.c_int, [C.c_int, C.POINTER(C.c_long)]),
    "lpbox_policy_score_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lpbox_policy_rescore_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "lpbox_set_active": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lpbox_get_x_iters_device": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_long)]),
    "lpbox_get_x_sol": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "lpbox_get_final_x_sol": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "lpbox_cal_obj": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "lpbox_cur_bin_obj": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "lpbox_check_infeasible_lpbox": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_check_infeasible_l2f": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_set_problem_bqp": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _ip, _ip, _dp, _dp, C.c_double, C.c_int, C.c_int]),
    "lpbox_seg_set_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "lpbox_read_jpeg_gray": (C.c_int, [C.c_char_p, C.c_void_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lpbox_seg_legacy": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "lpbox_seg_legacy_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "lpbox_seg_get_obj": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "lpbox_seg_get_shape": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lpbox_seg_get_problem": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.POINTER(C.c_double)]),
    "lpbox_get_config": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lpbox_get_layout": (C.c_int, [C.c_void_p, C.c_int, _ip]),
    "lpbox_get_row_split": (C.c_int, [C.c_void_p, C.c_int, _ip]),
    "lpbox_get_col_split": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip]),
    "lpbox_get_counters": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "lpbox_get_stop": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lpbox_kernel_time": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]),
    "lpbox_debug_get_vec": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, _dp, C.c_int]),
    "lpbox_debug_get_scalar": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_double)]),
    "lpbox_set_log": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_get_log": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int]),
    "lpbox_big_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int]),
    "lpbox_big_destroy": (None, [C.c_void_p]),
    "lpbox_big_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lpbox_big_set_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "lpbox_big_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "lpbox_big_rccl_init": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lpbox_big_set_problem": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, _ip, _ip, _dp, C.c_void_p]),
    "lpbox_big_set_pcg_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_big_init": (C.c_int, [C.c_void_p]),
    "lpbox_big_iterate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "lpbox_big_set_record": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_big_iterate_l2f": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.POINTER(C.c_int)]),
    "lpbox_big_get_n": (C.c_int, [C.c_void_p]),
    "lpbox_big_get_x_iters": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "lpbox_big_get_x_iters_device": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "lpbox_big_get_x_sol": (C.c_int, [C.c_void_p, _dp]),
    "lpbox_big_cal_obj": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "lpbox_big_get_x": (C.c_int, [C.c_void_p, _dp]),
    "lpbox_big_get_vec": (C.c_int, [C.c_void_p, C.c_char_p, _dp, C.c_long]),
    "lpbox_big_get_scalar": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]),
    "lpbox_big_check_infeasible": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_bqp_create": (C.c_void_p, [C.c_int]),
    "lpbox_bqp_destroy": (None, [C.c_void_p]),
    "lpbox_bqp_preset": (C.c_int, [C.c_void_p, C.c_int]),
    "lpbox_bqp_set_params": (C.c_int, [C.c_void_p, _dp]),
    "lpbox_bqp_set_problem": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4),
    "lpbox_bqp_solve": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "lpbox_bqp_get_vec": (C.c_int, [C.c_void_p, C.c_char_p, _dp, C.c_long]),
    "lpbox_bqp_get_scalar": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]),
}

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p)

_lib = None


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64 (SONAME libamdhip64.so.7);
    if ours (from /opt/rocm) were mapped first, a later `import torch` would map a second copy and fail to see the GPU.
    Mapping torch's copy first (without importing torch) makes both resolve to the same object, in either import order."""
    if os.environ.get("LPBOX_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:       # no torch / unusual layout: fall back to the system runtime named in the library's RUNPATH
        pass


def load():
    """Load the shared library once; raise ImportError (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C accelerated-lpbox-admm_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class LpboxError(RuntimeError):
    code = None          # the LPBOX_E_* status that was returned


E_TOOLARGE = -9          # include/lpbox_hip.h: LPBOX_E_TOOLARGE


def check(rc, what="lpbox call"):
    """Turn a negative status into a Python exception carrying lpbox_last_error()."""
    if rc is not None and rc < 0:
        msg = load().lpbox_last_error()
        err = LpboxError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
        err.code = rc
        raise err
    return rc

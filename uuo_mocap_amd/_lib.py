"""ctypes binding of libuuo_hip.so (include/uuo_hip.h).  There is no CPU fallback: if the library is
missing the import of any operator fails loudly."""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuuo_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "uuo_hip.h")

UUO_STAGE_CHAMFER, UUO_STAGE_MARKER, UUO_STAGE_PART = 0, 1, 2


#: uuo_abi_version() of the library these bindings are written for
ABI_VERSION = 3


class UuoProblem(ctypes.Structure):
    _fields_ = [
        ("stage", c_int32), ("F", c_int32), ("M", c_int32),
        ("d_markers", c_void_p), ("d_o_pose", c_void_p), ("d_o_betas", c_void_p), ("d_root", c_void_p),
        ("d_assign", c_void_p), ("d_subset", c_void_p), ("n_subset", c_int32),
        ("w_data", c_float), ("w_pose", c_float), ("w_betas", c_float), ("marker_distance", c_float),
        ("pose_cache_id", ctypes.c_uint64),
        ("w_soft", c_float), ("soft_tau", c_float),   # EXTENSION: soft-assignment data term (part / chamfer stage)
        ("n_corners", c_int32), ("d_bary", c_void_p),  # marker stage on a three-corner (barycentric) placement
    ]


class UuoLbfgsOptions(ctypes.Structure):
    _fields_ = [
        ("max_iter", c_int32), ("history_size", c_int32), ("lr", c_float), ("tolerance_grad", c_float),
        ("tolerance_change", c_float), ("max_eval", c_int32), ("verbose", c_int32),
    ]


class UuoLbfgsStats(ctypes.Structure):
    _fields_ = [
        ("n_iter", c_int32), ("n_eval", c_int32), ("first_loss", c_float), ("final_loss", c_float),
        ("stop_reason", c_int32), ("device_ms", c_float),
    ]


class UuoReprojectionProblem(ctypes.Structure):
    _fields_ = [
        ("F", c_int32), ("M", c_int32), ("V", c_int32), ("J", c_int32),
        ("d_markers", c_void_p), ("d_joints0", c_void_p), ("d_verts0", c_void_p), ("d_kp_target", c_void_p),
        ("d_mask", c_void_p), ("focal", c_float * 2), ("center", c_float * 2),
        ("w_reprojection", c_float), ("w_chamfer", c_float),
    ]


EVAL_CALLBACK = ctypes.CFUNCTYPE(None, c_void_p, c_int, c_float, c_void_p)
# uuo_gather_fn (include/uuo_hip.h): int gather(user, const double* mine, int n, double* all /* [world][n] */)
GATHER_FN = ctypes.CFUNCTYPE(c_int, c_void_p, POINTER(ctypes.c_double), c_int, POINTER(ctypes.c_double))


class UuoShared(ctypes.Structure):
    _fields_ = [("gather", GATHER_FN), ("user", c_void_p), ("rank", c_int32), ("world", c_int32)]


_SIGNATURES = {
    "uuo_last_error": (c_char_p, []),
    "uuo_abi_version": (c_int, []),
    "uuo_model_create": (c_int, [c_void_p] * 7 + [c_int, POINTER(c_void_p)]),
    "uuo_model_destroy": (c_int, [c_void_p]),
    "uuo_model_num_verts": (c_int, [c_void_p]),
    "uuo_smpl_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "uuo_smpl_backward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "uuo_nn_argmin": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                              c_void_p, c_void_p]),
    "uuo_rigid_distance_std": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "uuo_assign_mean_argmin": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "uuo_soft_nn_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "uuo_soft_nn_backward": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "uuo_copy_to_host": (c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    "uuo_mesh_closest_points": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "uuo_fit_create": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "uuo_fit_destroy": (c_int, [c_void_p]),
    "uuo_problem_num_params": (c_int, [POINTER(UuoProblem)]),
    "uuo_closure_eval": (c_int, [c_void_p, c_void_p, POINTER(UuoProblem), c_void_p, c_void_p, c_void_p, c_void_p]),
    "uuo_lbfgs_solve": (c_int, [c_void_p, c_void_p, POINTER(UuoProblem), c_void_p, POINTER(UuoLbfgsOptions),
                                POINTER(UuoLbfgsStats), c_void_p, c_void_p]),
    "uuo_lbfgs_solve_shared": (c_int, [c_void_p, c_void_p, POINTER(UuoProblem), c_void_p, POINTER(UuoLbfgsOptions),
                                       POINTER(UuoLbfgsStats), POINTER(UuoShared), c_void_p, c_void_p]),
    "uuo_time_closure": (c_int, [c_void_p, c_void_p, POINTER(UuoProblem), c_void_p, c_int, c_int,
                                 POINTER(c_float)]),
    "uuo_lbfgs_minimize": (c_int, [c_void_p, c_int, c_void_p, POINTER(UuoLbfgsOptions), POINTER(UuoLbfgsStats), c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "uuo_reprojection_create": (c_int, [POINTER(UuoReprojectionProblem), POINTER(c_void_p)]),
    "uuo_reprojection_destroy": (c_int, [c_void_p]),
    "uuo_reprojection_num_params": (c_int, [POINTER(UuoReprojectionProblem)]),
    "uuo_reprojection_eval": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "uuo_reprojection_solve": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(UuoLbfgsOptions), POINTER(UuoLbfgsStats),
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "uuo_set_wait_policy": (c_int, [c_int, c_int]),
    "uuo_mailbox_open": (c_int, [c_char_p, c_int32, c_int32, ctypes.c_double, POINTER(c_void_p)]),
    "uuo_mailbox_close": (c_int, [c_void_p]),
    "uuo_mailbox_gather": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "uuo_mailbox_stats": (c_int, [c_void_p, POINTER(ctypes.c_ulonglong), POINTER(ctypes.c_ulonglong)]),
    "uuo_copy_device": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_size_t]),
    "uuo_batch_create": (c_int, [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    "uuo_batch_destroy": (c_int, [c_void_p]),
    "uuo_batch_solve": (c_int, [c_void_p, c_void_p, POINTER(UuoProblem), POINTER(c_void_p), c_int,
                                POINTER(UuoLbfgsOptions), POINTER(UuoLbfgsStats)]),
    "uuo_batch_part_scores": (c_int, [c_void_p, c_void_p, POINTER(UuoProblem), POINTER(c_void_p), c_int,
                                      POINTER(c_float)]),
}
# libuuo_hip_debug.so only (same sources built with -DUUO_DEBUG_HOOKS; loaded by tests/ and tools/, never by the
# package): the optimiser self-test on analytic objectives (tests/test_gpu_parity.py::test_lbfgs_*), buffer read-backs,
# kernel-variant knobs.  The shipped library exports include/uuo_hip.h and nothing else and reads no environment variable.
LIB_DEBUG_PATH = os.path.join(_HERE, "libuuo_hip_debug.so")
_DEBUG_SIGNATURES = {
    "uuo_lbfgs_selftest": (c_int, [c_void_p, c_int, c_int, c_void_p, POINTER(UuoLbfgsOptions),
                                   POINTER(UuoLbfgsStats), c_void_p, c_void_p]),
    "uuo_debug_fit_buffers": (c_int, [c_void_p, c_void_p, c_void_p]),
    "uuo_debug_nn_flags": (c_int, [c_void_p, c_void_p]),
    "uuo_debug_skin16_check": (c_int, [c_void_p, c_int]),
    "uuo_debug_small_coeffs": (c_int, [c_int, c_int, c_int, c_void_p]),
    "uuo_debug_time_small": (c_int, [c_int, c_int, c_int, POINTER(c_float)]),
    "uuo_debug_index_map": (c_int, [c_int, c_int, c_int, c_void_p]),
    "uuo_debug_staging_script": (c_int, [c_void_p, c_int, ctypes.c_longlong, c_void_p]),
}

# uuo_closure_fn (include/uuo_hip.h): int closure(user, stream, d_x_eval, d_loss, d_grad)
CLOSURE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)

_lib = None
_lib_debug = None


def header_symbols():
    """Function names declared in include/uuo_hip.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(uuo_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if n != "uuo_eval_callback_t"))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            "libuuo_hip.so is missing (%s): run `python __graft_entry__.py` to build the HIP extension. "
            "There is no CPU fallback for the fitted path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name in header_symbols():
        if not hasattr(lib, name):
            raise RuntimeError("libuuo_hip.so does not export %s declared in include/uuo_hip.h" % name)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.uuo_abi_version() != ABI_VERSION:  # the ctypes structures below mirror include/uuo_hip.h of THIS version
        raise RuntimeError("libuuo_hip.so has ABI version %d, this package binds version %d: rebuild it "
                           "(python __graft_entry__.py)" % (lib.uuo_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def load_debug():
    """The debug flavour of the library (tests/ and tools/ only): every public symbol plus the hooks above."""
    global _lib_debug
    if _lib_debug is not None:
        return _lib_debug
    if not os.path.isfile(LIB_DEBUG_PATH):
        raise RuntimeError("libuuo_hip_debug.so is missing (%s): run `python __graft_entry__.py`" % LIB_DEBUG_PATH)
    lib = ctypes.CDLL(LIB_DEBUG_PATH)
    for name, (res, args) in list(_SIGNATURES.items()) + list(_DEBUG_SIGNATURES.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib_debug = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().uuo_last_error()
        raise RuntimeError("%s failed (%d): %s" % (what or "libuuo_hip call", rc, msg.decode() if msg else ""))

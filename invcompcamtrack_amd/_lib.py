"""ctypes binding of libictr_hip.so (the C-ABI declared in include/ictr.h).

Follows the reference's only FFI idiom (misc_src/func_util_geom.py:582-606 loading libtriang.so): load the
shared object by absolute path, pass C-contiguous numpy buffers as typed pointers. Unlike the reference's
callers every argtype is declared, counts are int64 and every status is checked.

There is no fallback: if the library is missing, fails to load, or no HIP device is usable, the error is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libictr_hip.so")

FP = C.POINTER(C.c_float)
DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int)
FPP = C.POINTER(FP)
VP = C.c_void_p
I64 = C.c_int64


class IctrError(RuntimeError):
    pass


class OptParam(C.Structure):
    """optparam, utilities.h:46-61 (field order preserved)."""
    _fields_ = [("maxpttrack", C.c_int), ("psz", C.c_int), ("pszd2", C.c_int), ("pszd2m3", C.c_int),
                ("novals", C.c_int), ("lv_f", C.c_int), ("lv_l", C.c_int), ("donorm", C.c_bool),
                ("dopatchnorm", C.c_bool), ("maxiter", C.c_int), ("normdp_ratio", C.c_float),
                ("verbosity", C.c_int)]


class TraceRec(C.Structure):
    _fields_ = [("level", C.c_int), ("iter", C.c_int), ("H", C.c_float * 36), ("b", C.c_float * 6),
                ("dp", C.c_float * 6), ("p", C.c_float * 6)]


# name -> (restype, argtypes); every symbol declared in include/ictr.h
SIGNATURES = {
    "ictr_optparam_init": (C.c_int, [C.POINTER(OptParam), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                     C.c_int, C.c_int, C.c_int]),
    "ictr_last_error": (C.c_char_p, []),
    "ictr_version": (C.c_int, []),
    "ictr_device_count": (C.c_int, []),
    "ictr_set_device": (C.c_int, [C.c_int]),
    "ictr_stream_read_bandwidth": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "ictr_debug_transpose_reduce": (C.c_int, [FP, FP, IP, IP, C.c_int]),
    "ictr_cam_create": (C.c_int, [C.POINTER(VP), C.c_int, FP, FP, IP, C.c_int]),
    "ictr_cam_destroy": (None, [VP]),
    "ictr_cam_getfx": (C.c_float, [VP, C.c_int]),
    "ictr_cam_getfy": (C.c_float, [VP, C.c_int]),
    "ictr_cam_getcx": (C.c_float, [VP, C.c_int]),
    "ictr_cam_getcy": (C.c_float, [VP, C.c_int]),
    "ictr_cam_getswo": (C.c_float, [VP, C.c_int]),
    "ictr_cam_getsho": (C.c_float, [VP, C.c_int]),
    "ictr_cam_getsw": (C.c_float, [VP, C.c_int]),
    "ictr_cam_getsh": (C.c_float, [VP, C.c_int]),
    "ictr_se3_coeff_to_group_f": (None, [FP, FP]),
    "ictr_se3_coeff_to_group_d": (None, [DP, DP]),
    "ictr_se3_group_to_coeff_f": (None, [FP, FP]),
    "ictr_se3_group_to_coeff_d": (None, [DP, DP]),
    "ictr_solve6": (None, [FP, FP, FP]),
    "ictr_pyramid_create": (C.c_int, [C.POINTER(VP), FP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ictr_pyramid_create_device": (C.c_int, [C.POINTER(VP), VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "ictr_pyramid_rebuild_device": (C.c_int, [VP, VP, VP]),
    "ictr_pyramid_rebuild": (C.c_int, [VP, FP, VP]),
    "ictr_pyramid_create_from_host_planes": (C.c_int, [C.POINTER(VP), FPP, FPP, FPP, C.c_int, C.c_int, C.c_int,
                                                       C.c_int]),
    "ictr_pyramid_destroy": (None, [VP]),
    "ictr_pyramid_levels": (C.c_int, [VP]),
    "ictr_pyramid_level_dims": (C.c_int, [VP, C.c_int, IP, IP]),
    "ictr_pyramid_download": (C.c_int, [VP, C.c_int, C.c_int, FP]),
    "ictr_pyramid_device_plane": (VP, [VP, C.c_int, C.c_int]),
    "ictr_get_patch": (C.c_int, [VP, C.c_int, FP, I64, C.c_int, C.c_int, FP]),
    "ictr_get_patch_grad": (C.c_int, [VP, C.c_int, FP, I64, C.c_int, C.c_int, FP, FP, FP]),
    "ictr_ncc_score": (C.c_int, [VP, VP, VP, C.c_int, FP, I64, C.c_int, C.c_float, C.c_float, FP]),
    "ictr_pose_create": (C.c_int, [C.POINTER(VP), VP, C.POINTER(OptParam)]),
    "ictr_pose_destroy": (None, [VP]),
    "ictr_pose_setpose_se3": (C.c_int, [VP, DP, DP, C.c_double]),
    "ictr_pose_addpose_se3": (C.c_int, [VP, FP]),
    "ictr_pose_subpose_se3": (C.c_int, [VP, FP]),
    "ictr_pose_getpose_se3": (C.c_int, [VP, DP]),
    "ictr_pose_project_pt": (C.c_int, [VP, FP, FP, I64, C.c_int]),
    "ictr_pose_project_pt_save_rotated": (C.c_int, [VP, FP, FP, FP, I64, C.c_int]),
    "ictr_pose_get_state": (C.c_int, [VP, FP, FP]),
    "ictr_odometer_create": (C.c_int, [C.POINTER(VP), VP, C.POINTER(OptParam)]),
    "ictr_odometer_destroy": (None, [VP]),
    "ictr_odometer_set3dpoints": (C.c_int, [VP, DP, I64]),
    "ictr_odometer_setpose": (C.c_int, [VP, DP, VP, VP]),
    "ictr_odometer_setpose_host": (C.c_int, [VP, DP, FPP, FPP, FPP, FPP]),
    "ictr_odometer_trackpose": (C.c_int, [VP, DP]),
    "ictr_odometer_get2dpoints": (FP, [VP]),
    "ictr_odometer_set_stream": (C.c_int, [VP, VP]),
    "ictr_odometer_enable_trace": (C.c_int, [VP, C.c_int]),
    "ictr_odometer_trace": (C.c_int, [VP, C.POINTER(TraceRec), I64, C.POINTER(I64)]),
    "ictr_odometer_read_buffer": (C.c_int, [VP, C.c_int, FP, I64]),
    "ictr_odometer_get_norm": (C.c_int, [VP, DP, DP]),
    "ictr_odometer_set_variant": (C.c_int, [VP, C.c_int]),
    "ictr_odometer_set_team": (C.c_int, [VP, C.c_int, C.c_int, C.c_int]),
    "ictr_odometer_set_robust": (C.c_int, [VP, C.c_int, C.c_float]),
    "ictr_batch_create": (C.c_int, [C.POINTER(VP), VP, C.POINTER(OptParam), I64]),
    "ictr_batch_destroy": (None, [VP]),
    "ictr_batch_set_stream": (C.c_int, [VP, VP]),
    "ictr_batch_set3dpoints": (C.c_int, [VP, I64, DP, I64]),
    "ictr_batch_set3dpoints_norm": (C.c_int, [VP, I64, DP, I64, DP, C.c_double]),
    "ictr_batch_get_norm": (C.c_int, [VP, I64, DP, DP]),
    "ictr_batch_setpose": (C.c_int, [VP, I64, DP, VP, VP]),
    "ictr_batch_setpose_all": (C.c_int, [VP, DP, VP, VP]),
    "ictr_batch_track_async": (C.c_int, [VP]),
    "ictr_batch_get_poses": (C.c_int, [VP, DP]),
    "ictr_batch_get_iterations": (C.c_int, [VP, IP]),
    "ictr_batch_get2dpoints": (C.c_int, [VP, I64, FP]),
    "ictr_batch_set_variant": (C.c_int, [VP, C.c_int]),
    "ictr_batch_set_team": (C.c_int, [VP, C.c_int, C.c_int, C.c_int]),
    "ictr_batch_set_robust": (C.c_int, [VP, C.c_int, C.c_float]),
    "ictr_batch_read_buffer": (C.c_int, [VP, I64, C.c_int, FP, I64]),
    "ictr_batch_set_timing": (C.c_int, [VP, C.c_int]),
    "ictr_batch_get_level_times": (C.c_int, [VP, FP, FP]),
    "ictr_batch_get_kernel_times": (C.c_int, [VP, FP]),
    "ictr_batch_get_first_iter_times": (C.c_int, [VP, FP]),
    "ictr_batch_last_path": (C.c_int, [VP]),
    "ictr_batch_last_team": (C.c_int, [VP]),
    "ictr_timebase_mark": (C.c_int, []),
    "ictr_batch_get_kernel_intervals": (C.c_int, [VP, FP, FP]),
    "ictr_batch_get_setup_intervals": (C.c_int, [VP, FP, FP]),
    "ictr_batch_set_reduction_buffer": (C.c_int, [VP, VP]),
    "ictr_batch_enable_sharding": (C.c_int, [VP, C.c_int]),
    "ictr_batch_reduction_buffer": (VP, [VP]),
    "ictr_batch_begin": (C.c_int, [VP]),
    "ictr_batch_level_allreduce_needed": (C.c_int, [VP]),
    "ictr_batch_level_accumulate": (C.c_int, [VP, C.c_int]),
    "ictr_batch_level_finish": (C.c_int, [VP, C.c_int]),
    "ictr_batch_iter_accumulate": (C.c_int, [VP, C.c_int]),
    "ictr_batch_iter_finish": (C.c_int, [VP, C.c_int]),
    "ictr_flow_gather": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, DP, I64, DP]),
    "ictr_extract_bil_patches": (C.c_int, [DP, C.c_int, C.c_int, C.c_int, DP, I64, C.c_int, DP]),
    "ictr_p2p_create": (C.c_int, [C.POINTER(VP), C.c_int, C.c_int, I64]),
    "ictr_p2p_handle_bytes": (C.c_int, []),
    "ictr_p2p_local_handle": (C.c_int, [VP, VP]),
    "ictr_p2p_connect": (C.c_int, [VP, VP]),
    "ictr_p2p_allreduce": (C.c_int, [VP, VP, I64, VP]),
    "ictr_p2p_error": (C.c_int, [VP]),
    "ictr_p2p_destroy": (None, [VP]),
    "ictr_batch_set_peer_exchange": (C.c_int, [VP, VP]),
    "ictr_patchflow": (C.c_int, [VP, VP, FP, I64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, FP, IP, IP]),
    "ictr_patchflow_last_kernel_ms": (C.c_float, []),
    "ictr_icgn_create": (C.c_int, [C.POINTER(VP), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, IP,
                                   I64]),
    "ictr_icgn_destroy": (None, [VP]),
    "ictr_icgn_set_stream": (C.c_int, [VP, VP]),
    "ictr_icgn_set_frames": (C.c_int, [VP, I64, VP, VP]),
    "ictr_icgn_set_warp": (C.c_int, [VP, I64, DP]),
    "ictr_icgn_set_timing": (C.c_int, [VP, C.c_int]),
    "ictr_icgn_run_async": (C.c_int, [VP]),
    "ictr_icgn_get_results": (C.c_int, [VP, DP, IP, FP]),
    "ictr_icgn_get_kernel_times": (C.c_int, [VP, FP]),
    "ictr_icgn_set_rows": (C.c_int, [VP, C.c_int, C.c_int]),
    "ictr_icgn_enable_sharding": (C.c_int, [VP, C.c_int, VP]),
    "ictr_icgn_begin": (C.c_int, [VP]),
    "ictr_icgn_hess_accumulate": (C.c_int, [VP, C.c_int]),
    "ictr_icgn_hess_finish": (C.c_int, [VP, C.c_int]),
    "ictr_icgn_iter_accumulate": (C.c_int, [VP, C.c_int]),
    "ictr_icgn_iter_finish": (C.c_int, [VP, C.c_int]),
}

_lib = None


def _preload_torch_hip_runtime():
    """One process must hold ONE HIP/HSA runtime. PyTorch-ROCm wheels bundle their own libamdhip64.so (same
    SONAME as /opt/rocm's) and load it by file name, so if our library pulled in the system runtime first, a later
    `import torch` would bring up a second one and find no GPU. When torch is installed we therefore map its
    bundled runtime first (without importing torch); libictr_hip.so's NEEDED libamdhip64.so.7 then resolves to it
    by SONAME. ICTR_SYSTEM_HIP=1 keeps the system runtime (torch must then not be used in the process)."""
    if os.environ.get("ICTR_SYSTEM_HIP") == "1":
        return None
    import sys
    tlib = None
    if "torch" in sys.modules:
        tlib = os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib", "libamdhip64.so")
    else:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.origin:
            tlib = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if tlib and os.path.exists(tlib):
        try:
            return C.CDLL(tlib, mode=C.RTLD_GLOBAL)
        except OSError:
            return None
    return None


def load():
    """Load libictr_hip.so and declare every signature. Raises IctrError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IctrError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _preload_torch_hip_runtime()
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as exc:  # e.g. libamdhip64 missing
        raise IctrError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here means the .so is stale w.r.t. include/ictr.h
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = load().ictr_last_error()
        raise IctrError(f"ictr error {rc}: {msg.decode() if msg else '?'}")


def fp(a):
    return a.ctypes.data_as(FP)


def dp(a):
    return a.ctypes.data_as(DP)


def f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def f64c(a):
    return np.ascontiguousarray(a, dtype=np.float64)

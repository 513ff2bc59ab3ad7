"""ctypes binding of the C ABI in include/rtdm.h (librtdm_hip.so).

This is plumbing for tests and bench.py: it passes raw pointers (numpy buffers or
torch ``data_ptr()``) straight to the C entry points.  There is no Python or CPU fallback: if the
library is missing, or no HIP device is usable, loading / creating fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librtdm_hip.so")
# RTDM_LIB_VARIANT=NAME: a diagnostic / experimental build made by `make variant NAME=...` (lib/variants/); the shipped
# library is never rebuilt in place for an experiment
if os.environ.get("RTDM_LIB_VARIANT"):
    LIB_PATH = os.path.join(_HERE, "lib", "variants", "librtdm_hip_%s.so" % os.environ["RTDM_LIB_VARIANT"])

RTDM_OK = 0
STATUS = {0: "RTDM_OK", -1: "RTDM_ERR_BAD_PARAM", -2: "RTDM_ERR_BAD_SIZE", -3: "RTDM_ERR_NO_DEVICE",
          -4: "RTDM_ERR_HIP", -5: "RTDM_ERR_NOMEM", -6: "RTDM_ERR_UNSUPPORTED", -7: "RTDM_ERR_NULL"}
STAGES = ("prefilter", "search", "lrcheck", "speckle")


class RtdmError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__("%s failed: %s (%d) %s" % (where, STATUS.get(status, "?"), status, detail))


class Region(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("x", "y", "width", "height")]


class HsvRange(C.Structure):
    _fields_ = [("low", C.c_int * 3), ("high", C.c_int * 3)]


class SGMParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("blockSize", "minDisparity", "numDisparities", "P1", "P2", "uniquenessRatio",
                                       "speckleWindowSize", "speckleRange", "disp12MaxDiff", "paths")]


class BMParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "preFilterCap", "blockSize", "minDisparity", "numDisparities", "textureThreshold",
        "uniquenessRatio", "speckleWindowSize", "speckleRange", "disp12MaxDiff", "legacy_right_clamp")]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u8p, i16p, u16p, sz = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t
    sig = {
        "rtdm_strerror": (C.c_char_p, [C.c_int]),
        "rtdm_last_hip_error": (C.c_char_p, []),
        "rtdm_abi_version": (C.c_int, []),
        "rtdm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "rtdm_bm_default_params": (None, [C.POINTER(BMParams), C.c_int]),
        "rtdm_bm_create": (C.c_int, [C.POINTER(BMParams), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
        "rtdm_bm_destroy": (None, [vp]),
        "rtdm_bm_set_roi": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
        "rtdm_bm_get_params": (C.c_int, [vp, C.POINTER(BMParams)]),
        "rtdm_bm_compute": (C.c_int, [vp, u8p, sz, u8p, sz, C.c_int, C.c_int, i16p, sz]),
        "rtdm_bm_compute_device": (C.c_int, [vp, C.c_int, u8p, u8p, sz, sz, C.c_int, C.c_int, i16p, sz, sz, vp]),
        "rtdm_bm_compute_batch": (C.c_int, [vp, C.c_int, u8p, u8p, sz, sz, C.c_int, C.c_int, i16p, sz, sz]),
        "rtdm_bm_synchronize": (C.c_int, [vp]),
        "rtdm_bm_set_profiling": (C.c_int, [vp, C.c_int]),
        "rtdm_bm_get_stage_time": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_long)]),
        "rtdm_bm_reset_stage_times": (C.c_int, [vp]),
        "rtdm_bm_search_variant": (C.c_char_p, [vp]),
        "rtdm_bm_get_tuner_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
        "rtdm_debug_search_kernel": (None, [C.c_int]),
        "rtdm_morph_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
        "rtdm_morph_destroy": (None, [vp]),
        "rtdm_morph_in_buffer": (vp, [vp]),
        "rtdm_morph_out_buffer": (vp, [vp]),
        "rtdm_morph_run": (C.c_int, [vp, u8p, sz, u8p, sz, C.c_int, C.c_int]),
        "rtdm_morph_run_device": (C.c_int, [vp, C.c_int, u8p, sz, sz, u8p, sz, sz, C.c_int, C.c_int, vp]),
        "rtdm_sgm_default_params": (None, [C.POINTER(SGMParams), C.c_int, C.c_int]),
        "rtdm_sgm_create": (C.c_int, [C.POINTER(SGMParams), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
        "rtdm_sgm_destroy": (None, [vp]),
        "rtdm_sgm_compute": (C.c_int, [vp, u8p, sz, u8p, sz, C.c_int, C.c_int, i16p, sz]),
        "rtdm_sgm_compute_device": (C.c_int, [vp, C.c_int, u8p, u8p, sz, sz, C.c_int, C.c_int, i16p, sz, sz, vp]),
        "rtdm_sgm_get_pass_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.POINTER(C.c_int)]),
        "rtdm_bm_compute_depth": (C.c_int, [vp, u8p, sz, u8p, sz, C.c_int, C.c_int, C.POINTER(C.c_double), u8p, sz,
                                            C.POINTER(Region), C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int), i16p, sz]),
        "rtdm_depth_stats_device": (C.c_int, [C.c_int, i16p, sz, C.c_int, C.c_int, C.POINTER(C.c_double), u8p, sz,
                                              C.POINTER(Region), C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int), vp]),
        "rtdm_rectify_create": (C.c_int, [i16p, u16p, i16p, u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.POINTER(vp)]),
        "rtdm_rectify_destroy": (None, [vp]),
        "rtdm_rectify_gray": (C.c_int, [vp, u8p, sz, u8p, sz, u8p, sz, u8p, sz]),
        "rtdm_rectify_rgb": (C.c_int, [vp, C.c_int, u8p, sz, u8p, sz]),
        "rtdm_rectify_gray_device": (C.c_int, [vp, C.c_int, u8p, u8p, u8p, u8p, vp]),
        "rtdm_bm_compute_rgb": (C.c_int, [vp, vp, u8p, sz, u8p, sz, i16p, sz]),
        "rtdm_bm_compute_rgb_device": (C.c_int, [vp, vp, C.c_int, u8p, u8p, i16p, vp]),
        "rtdm_objects_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
        "rtdm_objects_destroy": (None, [vp]),
        "rtdm_objects_detect": (C.c_int, [vp, u8p, sz, C.POINTER(HsvRange), C.c_int, C.c_int, u8p, sz, C.POINTER(Region), C.c_int,
                                          C.POINTER(C.c_int), C.POINTER(Region)]),
        "rtdm_estimate_frame": (C.c_int, [vp, vp, vp, u8p, sz, u8p, sz, C.POINTER(C.c_double), C.POINTER(HsvRange), C.c_int, C.c_int,
                                          C.c_double, C.POINTER(Region), C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int,
                                          C.POINTER(C.c_int), i16p, sz]),
        "rtdm_synth_pairs_device": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p,
                                              sz, sz, C.c_int, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


EXPORTS = ("rtdm_strerror rtdm_last_hip_error rtdm_abi_version rtdm_device_count rtdm_bm_default_params "
           "rtdm_bm_create rtdm_bm_destroy rtdm_bm_set_roi rtdm_bm_get_params rtdm_bm_compute "
           "rtdm_bm_compute_device rtdm_bm_compute_batch rtdm_bm_synchronize rtdm_bm_set_profiling "
           "rtdm_bm_get_stage_time rtdm_bm_reset_stage_times rtdm_bm_search_variant rtdm_bm_get_tuner_stats rtdm_debug_search_kernel rtdm_morph_create "
           "rtdm_morph_destroy rtdm_morph_in_buffer rtdm_morph_out_buffer rtdm_morph_run "
           "rtdm_morph_run_device rtdm_synth_pairs_device rtdm_sgm_default_params rtdm_sgm_create rtdm_sgm_destroy "
           "rtdm_sgm_compute rtdm_sgm_compute_device rtdm_sgm_get_pass_stats rtdm_bm_compute_depth rtdm_depth_stats_device "
           "rtdm_rectify_create rtdm_rectify_destroy rtdm_rectify_gray rtdm_rectify_rgb rtdm_rectify_gray_device "
           "rtdm_bm_compute_rgb rtdm_bm_compute_rgb_device rtdm_objects_create rtdm_objects_destroy rtdm_objects_detect "
           "rtdm_estimate_frame").split()


def check(status, where):
    if status != RTDM_OK:
        detail = lib().rtdm_last_hip_error().decode() if status == -4 else ""
        raise RtdmError(status, where, detail)


def make_params(preFilterCap=31, blockSize=13, minDisparity=0, numDisparities=64, textureThreshold=10,
                uniquenessRatio=10, speckleWindowSize=100, speckleRange=32, disp12MaxDiff=1, legacy_right_clamp=0):
    """Defaults are the reference's literals (main.cpp:134-135)."""
    return BMParams(preFilterCap, blockSize, minDisparity, numDisparities, textureThreshold,
                    uniquenessRatio, speckleWindowSize, speckleRange, disp12MaxDiff, legacy_right_clamp)

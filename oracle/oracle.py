"""ctypes loader for the CPU oracle (librtdm_oracle.so).

TEST INFRASTRUCTURE ONLY -- parity unpinned, see oracle/rtdm_oracle.h.  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the product
package (rt-depth-map_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librtdm_oracle.so")


class SGMParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("blockSize", "minDisparity", "numDisparities", "P1", "P2", "uniquenessRatio",
                                       "speckleWindowSize", "speckleRange", "disp12MaxDiff", "paths")]


class BMParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "preFilterCap", "blockSize", "minDisparity", "numDisparities", "textureThreshold",
        "uniquenessRatio", "speckleWindowSize", "speckleRange", "disp12MaxDiff")] + [
        ("roi1", C.c_int * 4), ("roi2", C.c_int * 4)]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


def build_native():
    """bench.py's cpu_baseline leg: the same sources compiled -march=native ON THE BOX that runs them (the portable
    build travels with the repository and must run anywhere).  Falls back to the portable library."""
    global _SO, _lib
    nat = os.path.join(_HERE, "_build", "librtdm_oracle_native.so")
    try:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "OUT=_build/librtdm_oracle_native.so",
                               "CFLAGS=-O3 -march=native -fPIC -Wall -Wextra -std=c11"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        C.CDLL(nat)
    except (subprocess.CalledProcessError, OSError):
        return build()
    if _SO != nat:
        _SO, _lib = nat, None
    return nat


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p, i16p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.POINTER(C.c_int32)
        L.orc_prefilter_xsobel.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t, C.c_int]
        L.orc_prefilter_xsobel.restype = None
        L.orc_bm_compute.argtypes = [C.POINTER(BMParams), u8p, C.c_size_t, u8p, C.c_size_t, C.c_int,
                                     C.c_int, i16p, C.c_size_t, C.c_int]
        L.orc_bm_compute.restype = C.c_int
        L.orc_bm_valid_rect.argtypes = [C.POINTER(BMParams), C.c_int, C.c_int, C.POINTER(C.c_int * 4)]
        L.orc_bm_valid_rect.restype = C.c_int
        L.orc_bm_search.argtypes = [C.POINTER(BMParams), u8p, C.c_size_t, u8p, C.c_size_t, C.c_int, C.c_int,
                                    C.c_int, C.c_int, i16p, C.c_size_t, i32p, C.c_size_t]
        L.orc_bm_search.restype = None
        L.orc_bm_set_legacy_right_clamp.argtypes = [C.c_int]
        L.orc_bm_set_legacy_right_clamp.restype = None
        L.orc_validate_disparity.argtypes = [i16p, C.c_size_t, i32p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int]
        L.orc_validate_disparity.restype = None
        L.orc_filter_speckles.argtypes = [i16p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_filter_speckles.restype = None
        L.orc_ellipse_element.argtypes = [C.c_int, C.c_int, u8p]
        L.orc_ellipse_element.restype = None
        for n in ("orc_erode", "orc_dilate"):
            getattr(L, n).argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]
            getattr(L, n).restype = None
        L.orc_morph_open_close.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_int, C.c_int]
        L.orc_morph_open_close.restype = None
        u16p = C.POINTER(C.c_uint16)
        L.orc_sgm_compute.argtypes = [C.POINTER(SGMParams), u8p, C.c_size_t, u8p, C.c_size_t, C.c_int, C.c_int, i16p, C.c_size_t]
        L.orc_sgm_compute.restype = C.c_int
        L.orc_sgm_pixel_cost.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, u16p]
        L.orc_sgm_pixel_cost.restype = None
        L.orc_sgm_block_cost.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, u16p]
        L.orc_sgm_block_cost.restype = C.c_uint32
        L.orc_sgm_aggregate.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u16p]
        L.orc_sgm_aggregate.restype = None
        L.orc_sgm_aggregate_paths.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u16p]
        L.orc_sgm_aggregate_paths.restype = None
        L.orc_sgm_select.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i16p, C.c_size_t]
        L.orc_sgm_select.restype = None
        L.orc_median3x3_s16.argtypes = [i16p, C.c_size_t, i16p, C.c_size_t, C.c_int, C.c_int]
        L.orc_median3x3_s16.restype = None
        L.orc_depth_stats.argtypes = [i16p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double), u8p, C.c_size_t,
                                      C.POINTER(C.c_int), C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_depth_stats.restype = C.c_int
        dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int)
        L.orc_rgb2gray.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t]
        L.orc_rgb2gray.restype = None
        L.orc_remap_bilinear.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, i16p, u16p, C.c_int, C.c_int, u8p, C.c_size_t]
        L.orc_remap_bilinear.restype = None
        for n in ("orc_rectify_gray", "orc_rectify_rgb"):
            getattr(L, n).argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, i16p, u16p, ip, u8p, C.c_size_t]
            getattr(L, n).restype = C.c_int
        L.orc_init_undistort_rectify_map.argtypes = [dp, dp, dp, dp, C.c_int, C.c_int, i16p, u16p]
        L.orc_init_undistort_rectify_map.restype = C.c_int
        L.orc_rgb2hsv.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t]
        L.orc_rgb2hsv.restype = None
        L.orc_hsv_inrange.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, ip, ip, u8p, C.c_size_t]
        L.orc_hsv_inrange.restype = None
        L.orc_external_boxes.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, ip, C.c_int]
        L.orc_external_boxes.restype = C.c_int
        L.orc_union_box.argtypes = [ip, C.c_int, ip]
        L.orc_union_box.restype = None
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def make_params(preFilterCap=31, blockSize=13, minDisparity=0, numDisparities=64, textureThreshold=10,
                uniquenessRatio=10, speckleWindowSize=100, speckleRange=32, disp12MaxDiff=1,
                roi1=None, roi2=None):
    """Defaults = the reference's literals at main.cpp:134-135 (numDisparities from BASELINE)."""
    p = BMParams(preFilterCap, blockSize, minDisparity, numDisparities, textureThreshold,
                 uniquenessRatio, speckleWindowSize, speckleRange, disp12MaxDiff)
    p.roi1 = (C.c_int * 4)(*(roi1 or (0, 0, 0, 0)))
    p.roi2 = (C.c_int * 4)(*(roi2 or (0, 0, 0, 0)))
    return p


def prefilter_xsobel(img, cap):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    out = np.empty_like(img)
    lib().orc_prefilter_xsobel(_p(img, C.c_uint8), W, W, H, _p(out, C.c_uint8), W, cap)
    return out


def bm_compute(left, right, nthreads=1, **kw):
    """left/right: 2-D uint8 arrays (any row stride).  Returns int16 HxW disparity (x16)."""
    assert left.dtype == np.uint8 and right.dtype == np.uint8 and left.shape == right.shape
    assert left.strides[1] == 1 and right.strides[1] == 1
    H, W = left.shape
    p = kw.pop("params", None) or make_params(**kw)
    disp = np.empty((H, W), np.int16)
    rc = lib().orc_bm_compute(C.byref(p), _p(left, C.c_uint8), left.strides[0], _p(right, C.c_uint8),
                              right.strides[0], W, H, _p(disp, C.c_int16), W * 2, nthreads)
    if rc != 0:
        raise ValueError("orc_bm_compute failed: %d" % rc)
    return disp


def bm_search(lp, rp, row0, row1, **kw):
    lp = np.ascontiguousarray(lp, np.uint8); rp = np.ascontiguousarray(rp, np.uint8)
    H, W = lp.shape
    p = kw.pop("params", None) or make_params(**kw)
    fil = (p.minDisparity - 1) * 16
    disp = np.full((H, W), fil, np.int16)
    cost = np.zeros((H, W), np.int32)
    lib().orc_bm_search(C.byref(p), _p(lp, C.c_uint8), W, _p(rp, C.c_uint8), W, W, H, row0, row1,
                        _p(disp, C.c_int16), W, _p(cost, C.c_int32), W)
    return disp, cost


def set_legacy_right_clamp(on):
    """Hazard H1 of rtdm_oracle.h: 1 = the OpenCV 3.x right-border clamp (over-reads the row)."""
    lib().orc_bm_set_legacy_right_clamp(int(bool(on)))


def valid_rect(W, H, **kw):
    p = kw.pop("params", None) or make_params(**kw)
    r = (C.c_int * 4)()
    ok = lib().orc_bm_valid_rect(C.byref(p), W, H, C.byref(r))
    return tuple(r) if ok else None


def validate_disparity(disp, cost, minD, numD, disp12MaxDiff):
    disp = np.ascontiguousarray(disp, np.int16).copy(); cost = np.ascontiguousarray(cost, np.int32)
    H, W = disp.shape
    lib().orc_validate_disparity(_p(disp, C.c_int16), W, _p(cost, C.c_int32), W, W, H, minD, numD, disp12MaxDiff)
    return disp


def filter_speckles(disp, newVal, maxSpeckleSize, maxDiff):
    disp = np.ascontiguousarray(disp, np.int16).copy()
    H, W = disp.shape
    lib().orc_filter_speckles(_p(disp, C.c_int16), W, W, H, newVal, maxSpeckleSize, maxDiff)
    return disp


def ellipse_element(kw=10, kh=10):
    e = np.zeros((kh, kw), np.uint8)
    lib().orc_ellipse_element(kw, kh, _p(e, C.c_uint8))
    return e


def erode(img, kw=10, kh=10):
    img = np.ascontiguousarray(img, np.uint8); out = np.empty_like(img); H, W = img.shape
    lib().orc_erode(_p(img, C.c_uint8), W, _p(out, C.c_uint8), W, W, H, kw, kh)
    return out


def dilate(img, kw=10, kh=10):
    img = np.ascontiguousarray(img, np.uint8); out = np.empty_like(img); H, W = img.shape
    lib().orc_dilate(_p(img, C.c_uint8), W, _p(out, C.c_uint8), W, W, H, kw, kh)
    return out


def morph_open_close(img):
    img = np.ascontiguousarray(img, np.uint8); out = np.empty_like(img); H, W = img.shape
    lib().orc_morph_open_close(_p(img, C.c_uint8), W, _p(out, C.c_uint8), W, W, H)
    return out


def make_sgm_params(blockSize=5, minDisparity=0, numDisparities=128, P1=600, P2=2400, uniquenessRatio=10,
                    speckleWindowSize=100, speckleRange=32, disp12MaxDiff=1, paths=8):
    """Defaults: P1/P2 from sgbm-sw.cpp:17-18, the rest the literals main.cpp:134-135 uses for the BM matcher."""
    return SGMParams(blockSize, minDisparity, numDisparities, P1, P2, uniquenessRatio, speckleWindowSize,
                     speckleRange, disp12MaxDiff, paths)


def sgm_compute(left, right, **kw):
    left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
    H, W = left.shape
    p = kw.pop("params", None) or make_sgm_params(**kw)
    disp = np.empty((H, W), np.int16)
    rc = lib().orc_sgm_compute(C.byref(p), _p(left, C.c_uint8), W, _p(right, C.c_uint8), W, W, H, _p(disp, C.c_int16), W * 2)
    if rc != 0:
        raise ValueError("orc_sgm_compute failed: %d" % rc)
    return disp


def median3x3(img):
    img = np.ascontiguousarray(img, np.int16); H, W = img.shape
    out = np.empty_like(img)
    lib().orc_median3x3_s16(_p(img, C.c_int16), W, _p(out, C.c_int16), W, W, H)
    return out


def sgm_stages(left, right, **kw):
    """-> (pixel cost, block cost, aggregated S) as uint16 [H, W1, D] arrays, W1 = W - (minD + D) for minD >= 0."""
    left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
    H, W = left.shape
    p = kw.pop("params", None) or make_sgm_params(**kw)
    D, minD = p.numDisparities, p.minDisparity
    W1 = (W + min(minD, 0)) - max(minD + D, 0)
    pix = np.zeros((H, W1, D), np.uint16); Cc = np.zeros_like(pix); S = np.zeros_like(pix)
    L = lib()
    L.orc_sgm_pixel_cost(_p(left, C.c_uint8), W, _p(right, C.c_uint8), W, W, H, minD, D, _p(pix, C.c_uint16))
    L.orc_sgm_block_cost(_p(pix, C.c_uint16), W1, H, D, p.blockSize, _p(Cc, C.c_uint16))
    P1 = p.P1 if p.P1 > 0 else 2
    P2 = max(p.P2 if p.P2 > 0 else 5, P1 + 1)
    L.orc_sgm_aggregate_paths(_p(Cc, C.c_uint16), W1, H, D, P1, P2, 5 if p.paths == 5 else 8, _p(S, C.c_uint16))
    return pix, Cc, S


def depth_stats(disp16, Q, mask, regions, calibration_unit=25.0):
    """disp16: int16 HxW (x16); Q: 4x4; mask: uint8 HxW; regions: list of (x,y,w,h) -> (mean_cm[n], counts[n])."""
    disp16 = np.ascontiguousarray(disp16, np.int16); mask = np.ascontiguousarray(mask, np.uint8)
    H, W = disp16.shape
    q = np.ascontiguousarray(Q, np.float64).reshape(16)
    reg = np.ascontiguousarray(regions, np.int32).reshape(-1, 4)
    n = len(reg)
    mean = np.zeros(n, np.float64); cnt = np.zeros(n, np.int32)
    rc = lib().orc_depth_stats(_p(disp16, C.c_int16), W, W, H, _p(q, C.c_double), _p(mask, C.c_uint8), W,
                               _p(reg, C.c_int), n, calibration_unit, _p(mean, C.c_double), _p(cnt, C.c_int))
    if rc != 0:
        raise ValueError("orc_depth_stats failed: %d" % rc)
    return mean, cnt


# ---- rectification in front of the matcher (rectify_oracle.c) -------------------------------------------------
def rgb2gray(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8); H, W, _ = rgb.shape
    out = np.empty((H, W), np.uint8)
    lib().orc_rgb2gray(_p(rgb, C.c_uint8), W * 3, W, H, _p(out, C.c_uint8), W)
    return out


def remap_bilinear(src, map1, map2):
    """src: HxW or HxWx3 uint8; map1: dHxdWx2 int16; map2: dHxdW uint16 -> dHxdW(x3) uint8."""
    src = np.ascontiguousarray(src, np.uint8)
    cn = 1 if src.ndim == 2 else src.shape[2]
    sH, sW = src.shape[:2]
    map1 = np.ascontiguousarray(map1, np.int16); map2 = np.ascontiguousarray(map2, np.uint16)
    dH, dW = map2.shape
    out = np.empty((dH, dW) if cn == 1 else (dH, dW, cn), np.uint8)
    lib().orc_remap_bilinear(_p(src, C.c_uint8), sW * cn, sW, sH, cn, _p(map1, C.c_int16), _p(map2, C.c_uint16), dW, dH,
                             _p(out, C.c_uint8), dW * cn)
    return out


def rectify_gray(rgb, map1, map2, roi):
    rgb = np.ascontiguousarray(rgb, np.uint8); H, W, _ = rgb.shape
    map1 = np.ascontiguousarray(map1, np.int16); map2 = np.ascontiguousarray(map2, np.uint16)
    r = (C.c_int * 4)(*roi)
    out = np.empty((roi[3], roi[2]), np.uint8)
    rc = lib().orc_rectify_gray(_p(rgb, C.c_uint8), W * 3, W, H, _p(map1, C.c_int16), _p(map2, C.c_uint16), r,
                                _p(out, C.c_uint8), roi[2])
    if rc != 0:
        raise ValueError("orc_rectify_gray failed: %d" % rc)
    return out


def rectify_rgb(rgb, map1, map2, roi):
    rgb = np.ascontiguousarray(rgb, np.uint8); H, W, _ = rgb.shape
    map1 = np.ascontiguousarray(map1, np.int16); map2 = np.ascontiguousarray(map2, np.uint16)
    r = (C.c_int * 4)(*roi)
    out = np.empty((roi[3], roi[2], 3), np.uint8)
    rc = lib().orc_rectify_rgb(_p(rgb, C.c_uint8), W * 3, W, H, _p(map1, C.c_int16), _p(map2, C.c_uint16), r,
                               _p(out, C.c_uint8), roi[2] * 3)
    if rc != 0:
        raise ValueError("orc_rectify_rgb failed: %d" % rc)
    return out


def init_undistort_rectify_map(M, D, R, P, W, H):
    """M 3x3, D up to 14 coefficients, R 3x3, P 3x4 -> (map1 HxWx2 int16, map2 HxW uint16)."""
    m = np.ascontiguousarray(M, np.float64).reshape(9); r = np.ascontiguousarray(R, np.float64).reshape(9)
    d = np.zeros(14, np.float64); dd = np.asarray(D, np.float64).reshape(-1); d[:len(dd)] = dd
    p = np.ascontiguousarray(P, np.float64).reshape(12)
    map1 = np.empty((H, W, 2), np.int16); map2 = np.empty((H, W), np.uint16)
    dp = C.c_double
    rc = lib().orc_init_undistort_rectify_map(_p(m, dp), _p(d, dp), _p(r, dp), _p(p, dp), W, H, _p(map1, C.c_int16), _p(map2, C.c_uint16))
    if rc != 0:
        raise ValueError("orc_init_undistort_rectify_map failed: %d" % rc)
    return map1, map2


# ---- object detection -> ROI (objects_oracle.c) ---------------------------------------------------------------
HSV_LOW, HSV_HIGH = (0, 150, 0), (9, 255, 255)          # estimator.cpp:110-115


def rgb2hsv(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8); H, W, _ = rgb.shape
    out = np.empty_like(rgb)
    lib().orc_rgb2hsv(_p(rgb, C.c_uint8), W * 3, W, H, _p(out, C.c_uint8), W * 3)
    return out


def hsv_inrange(rgb, lo=HSV_LOW, hi=HSV_HIGH):
    rgb = np.ascontiguousarray(rgb, np.uint8); H, W, _ = rgb.shape
    out = np.empty((H, W), np.uint8)
    lib().orc_hsv_inrange(_p(rgb, C.c_uint8), W * 3, W, H, (C.c_int * 3)(*lo), (C.c_int * 3)(*hi), _p(out, C.c_uint8), W)
    return out


def external_boxes(mask, min_area=100, zero_border=True, max_boxes=4096):
    """-> list of (x, y, w, h) in the order of the reference's obj_boundings."""
    mask = np.ascontiguousarray(mask, np.uint8); H, W = mask.shape
    boxes = np.zeros((max_boxes, 4), np.int32)
    n = lib().orc_external_boxes(_p(mask, C.c_uint8), W, W, H, int(zero_border), min_area, _p(boxes, C.c_int), max_boxes)
    if n < 0:
        raise ValueError("orc_external_boxes failed: %d" % n)
    return [tuple(int(v) for v in b) for b in boxes[:min(n, max_boxes)]]


def union_box(boxes):
    b = np.ascontiguousarray(boxes, np.int32).reshape(-1, 4)
    roi = (C.c_int * 4)()
    lib().orc_union_box(_p(b, C.c_int), len(b), roi)
    return tuple(roi)

"""Rectification in front of the matcher: oracle/rectify_oracle.c against an
independent numpy implementation and known answers.  The reference only calls cvtColor / remap /
initUndistortRectifyMap here (estimator.cpp:29-39, main.cpp:95-96); their published 8-bit fixed-point behaviour
is restated -- PARITY UNPINNED (no OpenCV in the image, no reference tests)."""
import numpy as np
import pytest

import bruteforce as bf
import rectify_util as ru


def test_bilinear_table_is_the_closed_form_and_needs_no_fixup():
    tab = bf.bilinear_table()
    fy, fx = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
    closed = np.stack([(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32], -1)
    assert np.array_equal(tab, closed) and (tab.sum(-1) == 32768).all() and tab.max() <= 32767 + 1


def test_gray_weights(oracle):
    rgb = np.zeros((1, 4, 3), np.uint8)
    rgb[0, 0] = (255, 0, 0); rgb[0, 1] = (0, 255, 0); rgb[0, 2] = (0, 0, 255); rgb[0, 3] = (255, 255, 255)
    assert oracle.rgb2gray(rgb).tolist() == [[76, 150, 29, 255]]      # 0.299 / 0.587 / 0.114 of 255, rounded
    r = np.random.default_rng(1).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(oracle.rgb2gray(r), bf.rgb2gray(r))


def test_remap_known_answers(oracle):
    src = np.arange(6 * 8, dtype=np.uint8).reshape(6, 8) * 3
    ys, xs = np.meshgrid(np.arange(6), np.arange(8), indexing="ij")
    ident = np.stack([xs, ys], -1).astype(np.int16)
    zero = np.zeros((6, 8), np.uint16)
    assert np.array_equal(oracle.remap_bilinear(src, ident, zero), src)                       # identity
    sh = ident.copy(); sh[..., 0] += 2                                                        # integer shift, zero border
    want = np.zeros_like(src); want[:, :6] = src[:, 2:]
    assert np.array_equal(oracle.remap_bilinear(src, sh, zero), want)
    half = np.full((6, 8), 16, np.uint16)                                                     # fx = 1/2, fy = 0
    got = oracle.remap_bilinear(src, ident, half)
    nxt = np.concatenate([src[:, 1:], np.zeros((6, 1), np.uint8)], 1).astype(np.int32)
    assert np.array_equal(got, ((src.astype(np.int32) + nxt + 1) >> 1).astype(np.uint8))      # (a+b)/2, half rounds up
    out = ident.copy(); out[..., 0] = -2
    assert not oracle.remap_bilinear(src, out, half).any()                                    # fully outside -> 0
    edge = ident.copy(); edge[..., 0] = -1                                                    # left sample outside
    assert np.array_equal(oracle.remap_bilinear(src, edge, half)[:, 0], ((src[:, 0].astype(np.int32) + 1) >> 1).astype(np.uint8))


@pytest.mark.parametrize("cn", [1, 3])
def test_remap_matches_numpy_on_random_maps(oracle, cn):
    rng = np.random.default_rng(7 + cn)
    src = rng.integers(0, 256, (41, 67) if cn == 1 else (41, 67, 3), dtype=np.uint8)
    map1 = np.stack([rng.integers(-3, 70, (50, 60)), rng.integers(-3, 44, (50, 60))], -1).astype(np.int16)
    map2 = rng.integers(0, 1024, (50, 60)).astype(np.uint16)
    assert np.array_equal(oracle.remap_bilinear(src, map1, map2), bf.remap_bilinear(src, map1, map2))


@pytest.mark.parametrize("res", ["320x240", "640x480", "1280x720"])
def test_maps_from_the_reference_calibration(oracle, res):
    c, (l1, l2, r1, r2) = ru.maps(oracle, res)
    for (m1, m2), (M, D, R, P) in (((l1, l2), ("M1", "D1", "R1", "P1")), ((r1, r2), ("M2", "D2", "R2", "P2"))):
        b1, b2 = bf.init_undistort_rectify_map(c[M], c[D], c[R], c[P], c["W"], c["H"])
        fo = m1.astype(np.int64) * 32 + np.stack([m2 & 31, m2 >> 5], -1)       # back to 1/32-pixel fixed point
        fb = b1.astype(np.int64) * 32 + np.stack([b2 & 31, b2 >> 5], -1)
        diff = np.abs(fo - fb)
        assert diff.max() <= 1 and (diff != 0).mean() < 1e-3, (res, diff.max(), (diff != 0).mean())
    # inside the crop the reference uses, (nearly) every sample comes from inside the sensor frame
    x, y, w, h = c["roi"]
    for m1 in (l1, r1):
        roi = m1[y:y + h, x:x + w].astype(np.int32)
        inside = (roi[..., 0] >= 0) & (roi[..., 0] + 1 < c["W"]) & (roi[..., 1] >= 0) & (roi[..., 1] + 1 < c["H"])
        assert inside.mean() > 0.95, (res, inside.mean())
    # the maps are smooth inside the crop: neighbouring pixels sample neighbouring source pixels
    assert np.abs(np.diff(l1[y:y + h, x:x + w, 0].astype(np.int32), axis=1)).max() <= 3


def test_rectify_is_gray_then_remap_then_crop(oracle, synth):
    c, (l1, l2, r1, r2) = ru.maps(oracle, "320x240")
    left, _ = ru.rgb_pair(synth, 0, c["W"], c["H"])
    x, y, w, h = c["roi"]
    full = oracle.remap_bilinear(oracle.rgb2gray(left), l1, l2)
    assert np.array_equal(oracle.rectify_gray(left, l1, l2, c["roi"]), full[y:y + h, x:x + w])
    # remapping the colour frame and converting afterwards is NOT the same thing (rounding), but close
    col = oracle.rectify_rgb(left, l1, l2, c["roi"])
    assert np.array_equal(col, bf.remap_bilinear(left, l1, l2)[y:y + h, x:x + w])
    assert np.abs(bf.rgb2gray(col).astype(int) - full[y:y + h, x:x + w]).max() <= 1

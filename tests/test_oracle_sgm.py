"""SGM-8 oracle (oracle/sgm_oracle.c, BASELINE config 5) against the independent numpy implementation,
stage by stage.  PARITY UNPINNED against cv::StereoSGBM (the reference's SWSemiGlobalMatcher is a
wrapper that main.cpp never instantiates); the algorithm checked here is the one sgm_oracle.c defines."""
import numpy as np
import pytest

import bruteforce as bf


@pytest.mark.parametrize("W,H,D,minD,bs", [(56, 24, 16, 0, 5), (61, 19, 16, 0, 3), (70, 20, 32, 0, 5), (60, 16, 16, 3, 5),
                                           (60, 16, 16, -4, 5), (48, 18, 16, 0, 7)])
def test_volumes_match_bruteforce(oracle, synth, W, H, D, minD, bs):
    L, R = synth.make_pair(synth.STREAM_SEED + 500 + W, W, H, D)
    pix, C, S = oracle.sgm_stages(L, R, numDisparities=D, minDisparity=minD, blockSize=bs)
    bp, bC, bS = bf.sgm_volumes(L, R, D, minD, bs)
    assert np.array_equal(pix, bp)
    assert np.array_equal(C, bC)
    assert np.array_equal(S, bS)


@pytest.mark.parametrize("kw", [dict(), dict(disp12MaxDiff=-1, speckleWindowSize=0), dict(uniquenessRatio=0),
                                dict(minDisparity=2), dict(speckleWindowSize=20, speckleRange=2), dict(P1=8, P2=32)])
def test_full_sgm_matches_bruteforce(oracle, synth, kw):
    L, R = synth.make_pair(synth.STREAM_SEED + 600, 72, 28, 16)
    a = oracle.sgm_compute(L, R, numDisparities=16, **kw)
    b = bf.sgm(L, R, numDisparities=16, **kw)
    assert np.array_equal(a, b)
    assert (a != ((kw.get("minDisparity", 0) - 1) * 16)).any()


def test_five_path_mode_matches_bruteforce(oracle, synth):
    # paths=5: the directions of cv::StereoSGBM's default MODE_SGBM (what sgbm-sw.cpp:15 creates): no upward path
    L, R = synth.make_pair(synth.STREAM_SEED + 610, 64, 22, 16)
    _, _, S5 = oracle.sgm_stages(L, R, numDisparities=16, paths=5)
    _, _, S8 = oracle.sgm_stages(L, R, numDisparities=16)
    _, _, b5 = bf.sgm_volumes(L, R, 16, paths=5)
    assert np.array_equal(S5, b5) and not np.array_equal(S5, S8) and (S5 <= S8).all()
    assert np.array_equal(oracle.sgm_compute(L, R, numDisparities=16, paths=5), bf.sgm(L, R, numDisparities=16, paths=5))


def test_kat_shifted_pair_recovers_shift(oracle, synth):
    W, H, D, k = 160, 60, 32, 9
    y, x = np.mgrid[0:H, 0:W]
    L = synth.left_value(4321, x, y).astype(np.uint8)
    R = synth.left_value(4321, x + k, y).astype(np.uint8)
    d = oracle.sgm_compute(L, R, numDisparities=D, speckleWindowSize=0, disp12MaxDiff=-1)
    v = d[:, D:][d[:, D:] != -16]
    assert v.size > 0.9 * H * (W - D)
    assert (v == k * 16).mean() > 0.98          # exact match: the parabola is symmetric, sub-pixel term 0
    assert (d[:, :D] == -16).all()


def test_kat_constant_image_picks_disparity_zero(oracle):
    img = np.full((20, 64), 90, np.uint8)
    d = oracle.sgm_compute(img, img, numDisparities=16, uniquenessRatio=0, speckleWindowSize=0, disp12MaxDiff=-1)
    assert (d[:, 16:] == 0).all()               # all costs tie: first minimum = d 0


def test_parameter_validation(oracle, synth):
    L, R = synth.make_pair(synth.STREAM_SEED, 64, 20, 16)
    for bad in (dict(numDisparities=20), dict(blockSize=4), dict(P1=0), dict(P1=10, P2=10), dict(uniquenessRatio=101)):
        kw = dict(numDisparities=16); kw.update(bad)
        with pytest.raises(ValueError):
            oracle.sgm_compute(L, R, **kw)

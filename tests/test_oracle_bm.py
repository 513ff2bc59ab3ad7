"""CPU suite: the C oracle against the independent brute-force implementation and against the
known-answer properties of SURVEY.md section 8c.  The reference holds no tests or golden vectors for
this path (its arithmetic lives in OpenCV), so these are the pins the oracle has: PARITY UNPINNED
against the real SWMatcherKonolige."""
import numpy as np
import pytest

import bruteforce as bf

FIL = -16


def pair(synth, W, H, D, seed=0):
    return synth.make_pair(synth.STREAM_SEED + seed, W, H, D)


# ---- stage gate 1: prefilter --------------------------------------------------------------------
@pytest.mark.parametrize("W,H", [(64, 48), (97, 65), (33, 21), (16, 2), (8, 1)])
@pytest.mark.parametrize("cap", [1, 31, 63])
def test_prefilter_matches_bruteforce(oracle, W, H, cap):
    rng = np.random.default_rng(W * 1000 + H + cap)
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    assert np.array_equal(oracle.prefilter_xsobel(img, cap), bf.prefilter_xsobel(img, cap))


def test_prefilter_odd_height_last_row_is_cap(oracle):
    # KAT (7): rows are produced in pairs, a trailing odd row is all ftzero
    img = np.random.default_rng(1).integers(0, 256, (41, 50), dtype=np.uint8)
    out = oracle.prefilter_xsobel(img, 31)
    assert (out[-1] == 31).all() and (out[:, 0] == 31).all() and (out[:, -1] == 31).all()
    assert out.max() <= 62


# ---- stage gate 2: SAD search alone (A.3), then +LR (A.4), then +speckle (A.5) ------------------
CASES = [
    # W, H, D, w, minD
    (64, 48, 16, 5, 0), (96, 64, 32, 7, 0), (97, 65, 16, 9, 0), (80, 50, 16, 13, 0),
    (96, 64, 16, 7, 4), (96, 64, 16, 7, -5), (90, 40, 32, 5, -40), (72, 56, 48, 9, 0),
]


@pytest.mark.parametrize("W,H,D,w,minD", CASES)
def test_search_only_matches_bruteforce(oracle, synth, W, H, D, w, minD):
    L, R = pair(synth, W, H, D, seed=W + H)
    kw = dict(numDisparities=D, blockSize=w, minDisparity=minD, speckleWindowSize=0, disp12MaxDiff=-1)
    assert np.array_equal(oracle.bm_compute(L, R, **kw), bf.stereo_bm(L, R, **kw))


@pytest.mark.parametrize("W,H,D,w,minD", CASES)
@pytest.mark.parametrize("maxdiff", [0, 1])
def test_search_plus_lr_matches_bruteforce(oracle, synth, W, H, D, w, minD, maxdiff):
    L, R = pair(synth, W, H, D, seed=W + H + 7)
    kw = dict(numDisparities=D, blockSize=w, minDisparity=minD, speckleWindowSize=0, disp12MaxDiff=maxdiff)
    assert np.array_equal(oracle.bm_compute(L, R, **kw), bf.stereo_bm(L, R, **kw))


@pytest.mark.parametrize("W,H,D,w,minD", CASES[:5])
def test_full_pipeline_matches_bruteforce(oracle, synth, W, H, D, w, minD):
    L, R = pair(synth, W, H, D, seed=W + H + 13)
    kw = dict(numDisparities=D, blockSize=w, minDisparity=minD, speckleWindowSize=20, speckleRange=32)
    a, b = oracle.bm_compute(L, R, **kw), bf.stereo_bm(L, R, **kw)
    assert np.array_equal(a, b)
    assert (a != (minD - 1) * 16).any()


@pytest.mark.parametrize("uniq,tex", [(0, 0), (0, 50), (15, 10), (40, 0), (300, 10)])
def test_rejection_thresholds_match_bruteforce(oracle, synth, uniq, tex):
    L, R = pair(synth, 96, 64, 16, seed=99)
    kw = dict(numDisparities=16, blockSize=7, uniquenessRatio=uniq, textureThreshold=tex,
              speckleWindowSize=0, disp12MaxDiff=-1)
    assert np.array_equal(oracle.bm_compute(L, R, **kw), bf.stereo_bm(L, R, **kw))


def test_roi_matches_bruteforce_and_masks(oracle, synth):
    # estimator.cpp:54 sets ROI1 per frame; ROI2 stays unset (estimator.cpp:55)
    L, R = pair(synth, 120, 80, 16, seed=5)
    roi = (30, 20, 70, 40)
    kw = dict(numDisparities=16, blockSize=7, roi1=roi, speckleWindowSize=0)
    a = oracle.bm_compute(L, R, **kw)
    assert np.array_equal(a, bf.stereo_bm(L, R, **kw))
    rect = oracle.valid_rect(120, 80, numDisparities=16, blockSize=7, roi1=roi)
    assert rect == (30 + 3, 20 + 3, 70 - 6, 40 - 6)
    m = np.ones_like(a, bool); m[rect[1]:rect[1] + rect[3], rect[0]:rect[0] + rect[2]] = False
    assert (a[m] == FIL).all() and (a[~m] != FIL).any()


def test_pitched_views_equal_contiguous(oracle, synth):
    # estimator.cpp:33,36 pass non-contiguous ROI views (cols = roif.width, step = full width);
    # backup/320x240/extrinsics.yml:56-57 + main.cpp:80-85 give the 233x156 crop at (49,46).
    L, R = pair(synth, 320, 240, 32)
    lv, rv = L[46:46 + 156, 49:49 + 233], R[46:46 + 156, 49:49 + 233]
    kw = dict(numDisparities=32, blockSize=7)
    a = oracle.bm_compute(lv, rv, **kw)
    b = oracle.bm_compute(np.ascontiguousarray(lv), np.ascontiguousarray(rv), **kw)
    assert np.array_equal(a, b) and a.shape == (156, 233)


# ---- known-answer tests ---------------------------------------------------
def test_kat_constant_image_is_all_filtered(oracle):
    img = np.full((48, 64), 77, np.uint8)
    assert (oracle.bm_compute(img, img, numDisparities=16, blockSize=5) == FIL).all()


def test_kat_borders_are_filtered(oracle, synth):
    W, H, D, w = 96, 64, 16, 7
    L, R = pair(synth, W, H, D)
    d = oracle.bm_compute(L, R, numDisparities=D, blockSize=w, speckleWindowSize=0, disp12MaxDiff=-1)
    r = w // 2
    assert (d[:, :D - 1 + r] == FIL).all() and (d[:, W - r:] == FIL).all()
    assert (d[:r] == FIL).all() and (d[H - r:] == FIL).all()
    assert (d[r:H - r, D - 1 + r:W - r] != FIL).mean() > 0.5


def test_kat_independent_of_stripe_count(oracle, synth):
    L, R = pair(synth, 160, 120, 32, seed=3)
    ref = oracle.bm_compute(L, R, numDisparities=32, blockSize=9, nthreads=1)
    for nt in (2, 3, 8):
        assert np.array_equal(ref, oracle.bm_compute(L, R, numDisparities=32, blockSize=9, nthreads=nt))


def test_kat_ties_resolve_to_largest_disparity(oracle):
    # a pattern with period 4 along x makes disparities k and k+4 exact ties; the first minimum in
    # the reversed index is the LARGEST disparity.  Uniqueness is off so ties are not rejected.
    H, W, D = 40, 96, 16
    x = np.arange(W)
    row = (np.array([10, 200, 90, 30])[x % 4]).astype(np.uint8)
    img = np.tile(row, (H, 1))
    img = (img + (np.arange(H)[:, None] % 3) * 7).astype(np.uint8)
    d = oracle.bm_compute(img, img, numDisparities=D, blockSize=5, uniquenessRatio=0, textureThreshold=0,
                          speckleWindowSize=0, disp12MaxDiff=-1)
    # (the last valid column sees the prefilter's constant border column, which only matches at
    # disparity 0, so it is left out)
    inner = d[2:-2, D - 1 + 2:W - 3]
    assert (((inner + 8) >> 4) == 12).all()   # candidates 0,4,8,12 tie; 12 wins


def test_kat_shifted_pair_recovers_shift(oracle, synth):
    W, H, D, k = 200, 100, 32, 11
    y, x = np.mgrid[0:H, 0:W]
    L = synth.left_value(1234, x, y).astype(np.uint8)
    R = synth.left_value(1234, x + k, y).astype(np.uint8)
    d = oracle.bm_compute(L, R, numDisparities=D, blockSize=9, speckleWindowSize=0, disp12MaxDiff=-1)
    v = d[d != FIL]
    assert v.size > 0.8 * (H - 8) * (W - D - 8)
    assert (((v + 8) >> 4) == k).mean() >= 0.99


def test_kat_speckle_window_boundary(oracle):
    # isolated 10x10 patch: removed with window 100, kept with window 99
    d = np.full((60, 80), 320, np.int16)
    d[20:30, 30:40] = 800
    out100 = oracle.filter_speckles(d, FIL, 100, 32)
    out99 = oracle.filter_speckles(d, FIL, 99, 32)
    assert (out100[20:30, 30:40] == FIL).all() and (out100 == 320).sum() == 60 * 80 - 100
    assert np.array_equal(out99, d)


@pytest.mark.parametrize("seed", range(4))
def test_speckle_matches_label_propagation(oracle, seed):
    rng = np.random.default_rng(seed)
    d = (rng.integers(0, 6, (40, 56)) * 40).astype(np.int16)
    d[rng.random((40, 56)) < 0.2] = FIL
    assert np.array_equal(oracle.filter_speckles(d, FIL, 12, 32), bf.speckle(d, FIL, 12, 32))


def test_validate_matches_bruteforce_on_random_fields(oracle):
    rng = np.random.default_rng(7)
    D = 16
    d = rng.integers(0, D * 16, (12, 90)).astype(np.int16)
    d[rng.random(d.shape) < 0.3] = FIL
    c = rng.integers(0, 500, d.shape).astype(np.int32)
    for md in (0, 1, 2):
        assert np.array_equal(oracle.validate_disparity(d, c, 0, D, md), bf.validate(d, c, 0, D, md))


def test_parameter_validation(oracle, synth):
    L, R = pair(synth, 64, 48, 16)
    for bad in (dict(numDisparities=20), dict(numDisparities=0), dict(blockSize=8), dict(blockSize=3),
                dict(blockSize=49), dict(preFilterCap=0), dict(preFilterCap=64),
                dict(textureThreshold=-1), dict(uniquenessRatio=-1)):
        kw = dict(numDisparities=16, blockSize=5); kw.update(bad)
        with pytest.raises(ValueError):
            oracle.bm_compute(L, R, **kw)


def test_too_narrow_image_is_all_filtered(oracle):
    img = np.random.default_rng(0).integers(0, 255, (40, 30), dtype=np.uint8)
    assert (oracle.bm_compute(img, img, numDisparities=32, blockSize=5) == FIL).all()


# ---- version hazard H1 (oracle/rtdm_oracle.h): right-image border clamp ---------------------------
def test_right_clamp_hazard_is_confined(oracle, synth):
    """OpenCV 3.x clamps the right sample base to W-rofs-1 and over-reads the row, 4.x (and the oracle) to W-D.
    The choice can only change the search output in the last w/2 columns, which lie outside the valid rectangle;
    in the final map it can only reach pixels through their left-right-check votes, i.e. columns >= W - w/2 - D - 1."""
    W, H, D, w = 160, 120, 32, 9
    r = w // 2
    L, R = pair(synth, W, H, D, seed=5)
    lp, rp = oracle.prefilter_xsobel(L, 31), oracle.prefilter_xsobel(R, 31)
    kw = dict(numDisparities=D, blockSize=w)
    d0, _ = oracle.bm_search(lp, rp, r, H - r - 1, **kw)
    f0 = oracle.bm_compute(L, R, **kw)
    oracle.set_legacy_right_clamp(True)
    try:
        d1, _ = oracle.bm_search(lp, rp, r, H - r - 1, **kw)
        f1 = oracle.bm_compute(L, R, **kw)
    finally:
        oracle.set_legacy_right_clamp(False)
    diff = d0 != d1
    assert not diff[:, :W - r].any()
    assert diff[:, W - r:].any()                      # the switch is live
    fd = f0 != f1
    assert not fd[:, :W - r - D - 1].any()
    assert fd.mean() < 0.01                           # a handful of pixels near the right edge at most

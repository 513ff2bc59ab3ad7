"""Depth statistics after the matcher: oracle/depth_oracle.c against numpy.
calc_depth is the reference's own code (estimator.cpp:206-263); `/= 16` and reprojectImageTo3D are OpenCV
calls whose published behaviour is restated -- PARITY UNPINNED.  Tolerance: the two CPU implementations sum
in the same order and agree to 1e-12 relative; counts are exact."""
import numpy as np
import pytest

import bruteforce as bf

# Q as stereoRectify builds it: [[1,0,0,-cx],[0,1,0,-cy],[0,0,0,f],[0,0,-1/Tx,(cx-cx')/Tx]]
Q = np.array([[1, 0, 0, -160.5], [0, 1, 0, -120.25], [0, 0, 0, 310.7], [0, 0, 1 / 2.4, 0.0]])


def scene(oracle, synth, seed=0):
    L, R = synth.make_pair(synth.STREAM_SEED + 900 + seed, 320, 240, 32)
    d = oracle.bm_compute(L, R, numDisparities=32, blockSize=9)
    mask = ((L > 110) * 255).astype(np.uint8)
    return d, mask


@pytest.mark.parametrize("seed", range(3))
def test_depth_stats_match_numpy(oracle, synth, seed):
    d, mask = scene(oracle, synth, seed)
    regions = [(40, 30, 100, 80), (0, 0, 320, 240), (200, 100, 60, 90), (5, 5, 1, 1), (100, 200, 50, 0)]
    m, c = oracle.depth_stats(d, Q, mask, regions)
    bm, bc = bf.depth_stats(d, Q, mask, regions)
    assert np.array_equal(c, bc) and c[1] > 1000 and c[4] == 0
    assert np.allclose(m, bm, rtol=1e-12, atol=0)


def test_missing_values_and_mask_are_excluded(oracle):
    d = np.full((20, 30), 160, np.int16)       # disparity 10 everywhere ...
    d[:, :10] = -16                            # ... except FILTERED columns (the image minimum -> Z = 10000)
    mask = np.full((20, 30), 255, np.uint8); mask[:5] = 0
    m, c = oracle.depth_stats(d, Q, mask, [(0, 0, 30, 20)], calibration_unit=25.0)
    assert c[0] == 15 * 20
    z = np.float32(310.7 / (10 / 2.4))
    assert abs(m[0] - float(z) * 2.5) < 1e-9


def test_rounding_of_the_x16_disparity_is_ties_to_even(oracle):
    # 8/16 -> 0, 24/16 -> 2, 40/16 -> 2, -8/16 -> -0, -24/16 -> -2   (convertTo rounds half to even)
    vals = np.array([[8, 24, 40, -8, -24, 23, 25, -40]], np.int16)
    want = np.array([0, 2, 2, 0, -2, 1, 2, -2])
    Qi = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])     # Z = d, W = 1
    mask = np.full(vals.shape, 255, np.uint8)
    for i, w in enumerate(want):
        if w == want.min():
            continue                               # the minimum is the "missing" marker
        m, c = oracle.depth_stats(vals, Qi, mask, [(i, 0, 1, 1)], calibration_unit=10.0)
        assert c[0] == 1 and m[0] == float(w), (i, m[0], w)


def test_region_outside_image_is_an_error(oracle, synth):
    d, mask = scene(oracle, synth)
    with pytest.raises(ValueError):
        oracle.depth_stats(d, Q, mask, [(300, 200, 40, 60)])

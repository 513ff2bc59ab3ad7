"""Object detection that produces the matcher's ROI: oracle/objects_oracle.c against
independent implementations and known answers.  cvtColor(BGR2HSV) / inRange / findContours(RETR_EXTERNAL) /
boundingRect are OpenCV calls whose published behaviour is restated -- PARITY UNPINNED; fill_bounding_rects_of_contours
and find_relevant_matching_region are the reference's own code (estimator.cpp:167-204)."""
import numpy as np
import pytest

import bruteforce as bf


def test_hsv_known_answers(oracle):
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [128, 64, 64], [10, 200, 30], [200, 10, 201]]], np.uint8)
    hsv = oracle.rgb2hsv(px)[0]
    assert hsv[0].tolist() == [0, 255, 255] and hsv[1].tolist() == [60, 255, 255] and hsv[2].tolist() == [120, 255, 255]
    assert hsv[3].tolist() == [0, 0, 255] and hsv[4].tolist() == [0, 0, 0]
    assert hsv[5].tolist() == [0, 128, 128]                        # s = 64*255/128 = 127.5 -> 128
    assert np.array_equal(oracle.rgb2hsv(px), bf.rgb2hsv(px))


def test_hsv_matches_python_ints_on_random_pixels(oracle):
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    rgb[:5] = rng.integers(0, 4, (5, 50, 3))                        # dark / nearly gray pixels
    rgb[5:10, :, 1] = rgb[5:10, :, 0]                               # ties between channels
    assert np.array_equal(oracle.rgb2hsv(rgb), bf.rgb2hsv(rgb))


def test_inrange_is_the_reference_red_filter(oracle):
    # estimator.cpp:110-115: H 0..9, S 150..255, V 0..255
    px = np.array([[[200, 20, 20], [200, 60, 20], [200, 120, 120], [20, 200, 20], [90, 5, 10], [0, 0, 0]]], np.uint8)
    # red; orange-red (hue 7); washed-out red (S too low); green; red with a blue tint (hue wraps to 178); black
    assert oracle.hsv_inrange(px)[0].tolist() == [255, 255, 0, 0, 0, 0]
    hsv = oracle.rgb2hsv(px)[0]
    assert hsv[1].tolist() == [7, 229, 200] and hsv[4][0] == 178 and hsv[2][1] == 102
    full = oracle.hsv_inrange(px, (0, 0, 0), (255, 255, 255))
    assert (full == 255).all()


def blobs(H=60, W=80):
    m = np.zeros((H, W), np.uint8)
    m[5:25, 5:30] = 255; m[10:20, 10:25] = 0          # A: a frame with a hole ...
    m[13:17, 14:20] = 255                             # ... and B nested in the hole
    m[30:50, 40:70] = 255                             # C: solid
    m[52:55, 2:6] = 255                               # D: small (area 12)
    m[0:4, 60:80] = 255                               # E: touches the top and right frame
    m[26:29, 31:34] = 255; m[25, 30] = 255            # F: joined to A only diagonally?  (24,29) is A's corner -> 8-connected
    return m


def test_external_boxes_known_answers(oracle):
    m = blobs()
    got = oracle.external_boxes(m, min_area=1, zero_border=False)
    # discovery order (first pixel, raster): E (0,60), A+F (5,5), C (30,40), D (52,2); B is nested -> not external; reversed
    assert got == [(2, 52, 4, 3), (40, 30, 30, 20), (5, 5, 29, 24), (60, 0, 20, 4)]
    assert oracle.external_boxes(m, min_area=100, zero_border=False) == [(40, 30, 30, 20), (5, 5, 29, 24)]      # 12 and 80 drop out
    # OpenCV <= 3.1 clears the outermost rows/columns first: E loses row 0 and column 79
    assert oracle.external_boxes(m, min_area=1, zero_border=True)[-1] == (60, 1, 19, 3)
    assert oracle.union_box(got) == (2, 0, 78, 55)
    assert oracle.external_boxes(np.zeros((10, 10), np.uint8), 1) == []


@pytest.mark.parametrize("seed", range(12))
def test_external_boxes_match_scipy(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    H, W = int(rng.integers(20, 70)), int(rng.integers(20, 90))
    density = float(rng.choice([0.35, 0.5, 0.62]))
    m = (rng.random((H, W)) < density).astype(np.uint8) * 255
    if seed % 3 == 0:                                   # blobby instead of salt and pepper: rings inside rings
        from scipy import ndimage as ndi
        m = (ndi.uniform_filter(rng.random((H, W)), 5) > 0.5).astype(np.uint8) * 255
        m[H // 4:3 * H // 4, W // 4] = 255; m[H // 4:3 * H // 4, 3 * W // 4] = 255; m[H // 4, W // 4:3 * W // 4 + 1] = 255; m[3 * H // 4, W // 4:3 * W // 4 + 1] = 255
    for zb in (True, False):
        for area in (1, 6):
            assert oracle.external_boxes(m, area, zb) == bf.external_boxes(m, area, zb), (seed, zb, area)

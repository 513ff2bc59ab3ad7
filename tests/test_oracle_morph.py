"""Morphology oracle (mf-sw.cpp:19-28) against the hand-listed element rows of SURVEY.md Appendix B
and the brute-force numpy implementation.  PARITY UNPINNED vs cv::erode/cv::dilate."""
import numpy as np
import pytest

import bruteforce as bf


def test_ellipse_rows_match_appendix_b(oracle):
    e = oracle.ellipse_element(10, 10)
    for i, (j1, j2) in enumerate(bf.ELLIPSE_10):
        want = np.zeros(10, np.uint8); want[j1:j2 + 1] = 1
        assert np.array_equal(e[i], want), i


@pytest.mark.parametrize("W,H", [(64, 48), (37, 29), (12, 9), (233, 156)])
def test_erode_dilate_match_bruteforce(oracle, W, H):
    rng = np.random.default_rng(W + H)
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    assert np.array_equal(oracle.erode(img), bf.morph(img, False))
    assert np.array_equal(oracle.dilate(img), bf.morph(img, True))
    mask = ((rng.random((H, W)) < 0.5) * 255).astype(np.uint8)
    assert np.array_equal(oracle.morph_open_close(mask), bf.morph_open_close(mask))


def test_kat_single_pixel_is_erased_by_opening(oracle):
    img = np.zeros((40, 40), np.uint8); img[20, 20] = 255
    assert (oracle.morph_open_close(img) == 0).all()


def test_kat_element_shaped_blob(oracle):
    # a blob that is exactly the element placed with its anchor on (30,30): erosion leaves the
    # anchor pixel only; because dilation uses the SAME offsets (no reflection, Appendix B), the
    # opening returns the point-reflected element, which is what dilating the seed alone gives.
    e = oracle.ellipse_element()
    blob = np.zeros((60, 60), np.uint8)
    blob[25:35, 25:35] = e * 255
    er = oracle.erode(blob)
    assert er[30, 30] == 255 and er.sum() == 255
    seed = np.zeros((60, 60), np.uint8); seed[30, 30] = 255
    opened = oracle.dilate(er)
    assert np.array_equal(opened, oracle.dilate(seed))
    assert opened.sum() // 255 == e.sum()
    assert np.array_equal(opened[26:36, 26:36], e[::-1, ::-1] * 255)


def test_border_samples_never_win(oracle):
    img = np.full((20, 20), 200, np.uint8)
    assert (oracle.erode(img) == 200).all() and (oracle.dilate(img) == 200).all()

"""GPU parity of the object detection and of the whole per-frame chain
(estimator.cpp:29-77 without capture, decode and drawing) against the oracle, through the C ABI
(rtdm_objects_detect, rtdm_estimate_frame).  Masks, boxes, ROI, disparity and pixel counts are bit-exact; the mean
depth is a floating-point mean summed in a different order (tolerance 1e-9 relative, as for rtdm_bm_compute_depth)."""
import numpy as np
import pytest

import rectify_util as ru

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch                         # torch first: it brings its own HIP runtime and must initialise before ours
    assert torch.cuda.is_available()
    from conftest import load
    return load()


def paint(mask, rng):
    """RGB crop whose red filter response is exactly `mask`."""
    H, W = mask.shape
    g = rng.integers(0, 256, (H, W), dtype=np.uint8)
    rgb = np.stack([g, g, g], -1)
    rgb[mask] = np.stack([150 + g.astype(np.int32) // 3, g // 6, g // 6], -1).astype(np.uint8)[mask]
    return np.ascontiguousarray(rgb)


def oracle_detect(oracle, rgb, min_area, zero_border, low=None, high=None):
    m = oracle.hsv_inrange(rgb, low or oracle.HSV_LOW, high or oracle.HSV_HIGH)
    out = oracle.morph_open_close(m)
    boxes = oracle.external_boxes(out, min_area, zero_border)
    return m, out, boxes


@pytest.mark.parametrize("seed", range(8))
def test_detection_matches_the_oracle(pkg, oracle, seed):
    from scipy import ndimage as ndi
    rng = np.random.default_rng(40 + seed)
    H, W = int(rng.integers(60, 260)), int(rng.integers(80, 400))
    field = ndi.uniform_filter(rng.random((H, W)), int(rng.choice([9, 15, 25])))
    mask = field > np.quantile(field, float(rng.choice([0.5, 0.7, 0.85])))
    if seed % 2:
        mask[H // 5:4 * H // 5, W // 5:4 * W // 5] |= True                       # a big block ...
        mask[H // 4:3 * H // 4, W // 4:3 * W // 4] &= field[H // 4:3 * H // 4, W // 4:3 * W // 4] > np.quantile(field, 0.8)   # ... with islands inside
    rgb = paint(mask, rng)
    det = pkg.HIPObjectDetector(W, H)
    for zb in (True, False):
        for area in (100, 1):
            boxes, roi, out = det.detect(rgb, min_area=area, zero_border=zb, max_boxes=1024)
            m, want_out, want = oracle_detect(oracle, rgb, area, zb)
            assert np.array_equal(m != 0, mask)
            assert np.array_equal(out, want_out), (seed, zb, area)
            assert boxes == want, (seed, zb, area)
            if want:
                assert roi == oracle.union_box(want)
    det.close()


@pytest.mark.parametrize("W,H", [(1, 1), (2, 2), (5, 3), (11, 40), (300, 12), (23, 23)])
def test_tiny_and_degenerate_crops(pkg, oracle, W, H):
    # all foreground, all background, a checkerboard (8-connected into one component), random
    rng = np.random.default_rng(W * 100 + H)
    yy, xx = np.mgrid[0:H, 0:W]
    det = pkg.HIPObjectDetector(W, H)
    for mask in (np.ones((H, W), bool), np.zeros((H, W), bool), (yy + xx) % 2 == 0, rng.random((H, W)) < 0.5):
        rgb = paint(mask, rng)
        for zb in (True, False):
            boxes, roi, out = det.detect(rgb, min_area=1, zero_border=zb, max_boxes=512)
            _, want_out, want = oracle_detect(oracle, rgb, 1, zb)
            assert np.array_equal(out, want_out) and boxes == want, (W, H, zb, boxes, want)
    det.close()


def test_hsv_thresholds_other_than_the_default(pkg, oracle):
    rng = np.random.default_rng(9)
    rgb = rng.integers(0, 256, (120, 200, 3), dtype=np.uint8)
    rgb[30:90, 50:150] = (20, 180, 40)                                            # a green card
    det = pkg.HIPObjectDetector(200, 120)
    lo, hi = (50, 100, 50), (70, 255, 255)
    boxes, roi, out = det.detect(rgb, low=lo, high=hi, min_area=50, zero_border=True)
    _, want_out, want = oracle_detect(oracle, rgb, 50, True, lo, hi)
    assert np.array_equal(out, want_out) and boxes == want and len(boxes) >= 1
    det.close()


@pytest.mark.parametrize("res,D,w", [("320x240", 32, 7), ("1280x720", 64, 9)])
def test_whole_frame_chain(pkg, oracle, synth, res, D, w):
    c, maps = ru.maps(oracle, res)
    left, right = ru.red_scene(synth, 1, c["W"], c["H"], D)
    x, y, rw, rh = c["roi"]
    rect = pkg.HIPRectifier(*maps, roi=c["roi"])
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=rw, height=rh)
    det = pkg.HIPObjectDetector(rw, rh)
    min_area = 100 if res != "320x240" else 40
    boxes, mean, cnt, disp = pkg.estimate_frame(m, rect, det, left, right, c["Q"], min_area=min_area, want_disp=True)
    # the same chain on the CPU oracle
    gl = oracle.rectify_gray(left, maps[0], maps[1], c["roi"]); gr = oracle.rectify_gray(right, maps[2], maps[3], c["roi"])
    col = oracle.rectify_rgb(left, maps[0], maps[1], c["roi"])
    _, fout, want_boxes = oracle_detect(oracle, col, min_area, True)
    assert len(want_boxes) >= 2 and boxes == want_boxes[:64]
    roi = oracle.union_box(want_boxes)
    want_disp = oracle.bm_compute(gl, gr, numDisparities=D, blockSize=w, roi1=roi, nthreads=8)
    assert np.array_equal(disp, want_disp)
    wm, wc = oracle.depth_stats(want_disp, c["Q"], fout, want_boxes[:64])
    assert np.array_equal(cnt, wc) and wc.sum() > 0
    assert np.allclose(mean, wm, rtol=1e-9, atol=0)
    # a frame without red objects: the matcher is skipped
    gray = np.ascontiguousarray(np.stack([left[..., 1]] * 3, -1))
    boxes, mean, cnt = pkg.estimate_frame(m, rect, det, gray, gray, c["Q"])
    assert boxes == [] and len(mean) == 0
    det.close(); m.close(); rect.close()

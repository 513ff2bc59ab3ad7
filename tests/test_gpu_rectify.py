"""GPU parity of the rectification step in front of the matcher against
oracle/rectify_oracle.c, through the C ABI (rtdm_rectify_*, rtdm_bm_compute_rgb*).  Bit-exact (8-bit fixed point)."""
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


@pytest.mark.parametrize("res", ["320x240", "640x480", "1280x720"])
def test_gray_and_colour_rectification_match_the_oracle(pkg, oracle, synth, res):
    c, maps = ru.maps(oracle, res)
    left, right = ru.rgb_pair(synth, 1, c["W"], c["H"])
    r = pkg.HIPRectifier(*maps, roi=c["roi"])
    gl, gr = r.gray(left, right)
    assert np.array_equal(gl, oracle.rectify_gray(left, maps[0], maps[1], c["roi"]))
    assert np.array_equal(gr, oracle.rectify_gray(right, maps[2], maps[3], c["roi"]))
    assert np.array_equal(r.rgb(left, 0), oracle.rectify_rgb(left, maps[0], maps[1], c["roi"]))
    assert np.array_equal(r.rgb(right, 1), oracle.rectify_rgb(right, maps[2], maps[3], c["roi"]))
    # a row-padded caller frame (pitch > 3*W) gives the same bytes
    padded = np.zeros((c["H"], c["W"] + 5, 3), np.uint8); padded[:, :c["W"]] = left
    gl2, _ = r.gray(padded[:, :c["W"]], right)
    assert np.array_equal(gl2, gl)
    r.close()


def test_wild_maps_and_odd_sizes(pkg, oracle):
    # maps that point far outside the frame, at its edges and at every fractional offset; odd frame size (so
    # frames are not 4-byte aligned in a batch) and a crop at the very corner
    rng = np.random.default_rng(3)
    W, H = 211, 97
    maps = []
    for k in range(2):
        maps.append(np.stack([rng.integers(-4, W + 3, (H, W)), rng.integers(-4, H + 3, (H, W))], -1).astype(np.int16))
        maps.append(rng.integers(0, 1024, (H, W)).astype(np.uint16))
    import torch
    n = 3
    L = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8); R = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    for roi in ((0, 0, W, H), (W - 33, H - 21, 33, 21), (5, 7, 100, 50)):
        r = pkg.HIPRectifier(*maps, roi=roi, max_batch=n)
        dl = torch.empty((n, roi[3], roi[2]), dtype=torch.uint8, device="cuda"); dr = torch.empty_like(dl)
        r.gray_device(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), dl, dr)
        torch.cuda.synchronize()
        for i in range(n):
            assert np.array_equal(dl[i].cpu().numpy(), oracle.rectify_gray(L[i], maps[0], maps[1], roi)), (roi, i)
            assert np.array_equal(dr[i].cpu().numpy(), oracle.rectify_gray(R[i], maps[2], maps[3], roi)), (roi, i)
        assert np.array_equal(r.rgb(L[1], 1), oracle.rectify_rgb(L[1], maps[2], maps[3], roi))
        r.close()


@pytest.mark.parametrize("res,D,w", [("320x240", 32, 7), ("1280x720", 64, 9)])
def test_raw_frames_to_disparity_in_one_call(pkg, oracle, synth, res, D, w):
    # estimator.cpp:29-36 + 56: the rectified gray pair never leaves HBM; result = oracle rectify -> oracle matcher
    c, maps = ru.maps(oracle, res)
    left, right = ru.rgb_pair(synth, 2, c["W"], c["H"])
    x, y, rw, rh = c["roi"]
    r = pkg.HIPRectifier(*maps, roi=c["roi"], max_batch=2)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=rw, height=rh, max_batch=2)
    want = oracle.bm_compute(oracle.rectify_gray(left, maps[0], maps[1], c["roi"]),
                             oracle.rectify_gray(right, maps[2], maps[3], c["roi"]), numDisparities=D, blockSize=w, nthreads=8)
    assert np.array_equal(r.compute(m, left, right), want)
    # device batch of 3 (two chunks) on torch's stream
    import torch
    dL = torch.from_numpy(np.stack([left, right, left])).cuda(); dR = torch.from_numpy(np.stack([right, left, right])).cuda()
    dD = torch.empty((3, rh, rw), dtype=torch.int16, device="cuda")
    r.compute_device(m, dL, dR, dD)
    torch.cuda.synchronize()
    assert np.array_equal(dD[0].cpu().numpy(), want) and np.array_equal(dD[2].cpu().numpy(), want)
    want1 = oracle.bm_compute(oracle.rectify_gray(right, maps[0], maps[1], c["roi"]),
                              oracle.rectify_gray(left, maps[2], maps[3], c["roi"]), numDisparities=D, blockSize=w, nthreads=8)
    assert np.array_equal(dD[1].cpu().numpy(), want1)
    m.close(); r.close()


def test_one_pixel_crop_and_tiny_frame(pkg, oracle):
    rng = np.random.default_rng(11)
    W, H = 7, 5
    maps = []
    for k in range(2):
        maps.append(np.stack([rng.integers(-2, W + 1, (H, W)), rng.integers(-2, H + 1, (H, W))], -1).astype(np.int16))
        maps.append(rng.integers(0, 1024, (H, W)).astype(np.uint16))
    L = rng.integers(0, 256, (H, W, 3), dtype=np.uint8); R = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    for roi in ((3, 2, 1, 1), (0, 0, W, H), (6, 4, 1, 1)):
        r = pkg.HIPRectifier(*maps, roi=roi)
        gl, gr = r.gray(L, R)
        assert np.array_equal(gl, oracle.rectify_gray(L, maps[0], maps[1], roi)) and np.array_equal(gr, oracle.rectify_gray(R, maps[2], maps[3], roi))
        assert np.array_equal(r.rgb(R, 0), oracle.rectify_rgb(R, maps[0], maps[1], roi))
        r.close()


def test_argument_errors(pkg, oracle):
    c, maps = ru.maps(oracle, "320x240")
    with pytest.raises(pkg.binding.RtdmError):
        pkg.HIPRectifier(*maps, roi=(300, 0, 40, 40))           # crop sticks out of the frame
    r = pkg.HIPRectifier(*maps, roi=c["roi"])
    small = pkg.HIPMatcher(numOfDisparities=32, blockSize=7, width=100, height=100)
    frame = np.zeros((c["H"], c["W"], 3), np.uint8)
    with pytest.raises(pkg.binding.RtdmError):
        r.compute(small, frame, frame)                          # matcher created for a smaller frame than the crop
    small.close(); r.close()

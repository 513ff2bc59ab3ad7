"""Randomised parity sweep on the GPU: every kernel variant (fast quad-SAD instantiations, border columns,
generic LDS kernel in both tile widths and both accumulator types, left-right check with 32/64-bit keys,
speckle filter, SGM-8) is driven through the C ABI with random sizes / parameters / ROIs and compared
bit-for-bit with the oracle.  Seeds are fixed, so a failure is reproducible from its parameter string."""
import numpy as np
import pytest

from conftest import load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available()
    return load()


def _case(rng):
    D = int(rng.choice([16, 32, 48, 64, 96, 128, 192, 256], p=[.15, .2, .05, .25, .05, .15, .1, .05]))
    w = int(rng.choice([5, 7, 9, 11, 13, 15, 21, 25]))
    minD = int(rng.choice([0, 0, 0, 0, 3, -7, -20]))
    W = int(rng.integers(D + abs(minD) + w + 20, D + abs(minD) + w + 260))
    H = int(rng.integers(w + 3, w + 70))
    kw = dict(numDisparities=D, blockSize=w, minDisparity=minD,
              preFilterCap=int(rng.choice([31, 31, 31, 15, 63, 5])),
              textureThreshold=int(rng.choice([10, 0, 40, 200])),
              uniquenessRatio=int(rng.choice([10, 0, 15, 50])),
              speckleWindowSize=int(rng.choice([100, 0, 20, 400])),
              speckleRange=int(rng.choice([32, 4, 0, 64])),
              disp12MaxDiff=int(rng.choice([1, -1, 0, 3])))
    roi1 = roi2 = None
    if rng.random() < 0.35:
        x0, y0 = int(rng.integers(0, W // 2)), int(rng.integers(0, H // 2))
        roi1 = (x0, y0, int(rng.integers(1, W - x0 + 1)), int(rng.integers(1, H - y0 + 1)))
    if rng.random() < 0.15:
        x0, y0 = int(rng.integers(0, W // 3)), int(rng.integers(0, H // 3))
        roi2 = (x0, y0, int(rng.integers(W // 2, W - x0 + 1)), int(rng.integers(H // 2, H - y0 + 1)))
    return W, H, kw, roi1, roi2


@pytest.mark.parametrize("seed", range(48))
def test_random_block_matching_configuration(pkg, oracle, synth, seed):
    rng = np.random.default_rng(1000 + seed)
    W, H, kw, roi1, roi2 = _case(rng)
    L, R = synth.make_pair(synth.STREAM_SEED + 5000 + seed, W, H, kw["numDisparities"])
    if seed % 5 == 0:                       # pitched, odd-aligned views like estimator.cpp:33,36
        pad = np.zeros((H + 3, W + 37), np.uint8)
        pl, pr = pad.copy(), pad.copy()
        pl[2:2 + H, 5:5 + W] = L; pr[2:2 + H, 5:5 + W] = R
        L, R = pl[2:2 + H, 5:5 + W], pr[2:2 + H, 5:5 + W]
    want = oracle.bm_compute(L, R, roi1=roi1, roi2=roi2, **kw)
    m = pkg.HIPMatcher(numOfDisparities=kw["numDisparities"], blockSize=kw["blockSize"], minDisparity=kw["minDisparity"],
                       preFilterCap=kw["preFilterCap"], textureThreshold=kw["textureThreshold"],
                       uniquenessRatio=kw["uniquenessRatio"], speckleWindowSize=kw["speckleWindowSize"],
                       speckleRange=kw["speckleRange"], disp12MaxDiff=kw["disp12MaxDiff"], width=W, height=H)
    if roi1: m.setROI1(roi1)
    if roi2: m.setROI2(roi2)
    got = m.compute(L, R)
    variant = m.search_variant
    m.close()
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError("seed %d %dx%d %s roi1=%s roi2=%s variant=%s: %d pixels differ, first (y,x)=%s got %d want %d" % (
            seed, W, H, kw, roi1, roi2, variant, len(bad), tuple(bad[0]), got[tuple(bad[0])], want[tuple(bad[0])]))


@pytest.mark.parametrize("seed", range(16))
def test_random_sgm_configuration(pkg, oracle, synth, seed):
    # cv::StereoSGBM as restated in oracle/sgm_oracle.c: both modes, any blockSize the 16-bit costs allow (window <= 17 at the
    # reference's P2; an even size runs as the next odd one), the library's coercion of out-of-range P1 / P2 / uniquenessRatio / disp12MaxDiff
    rng = np.random.default_rng(2000 + seed)
    D = int(rng.choice([16, 32, 64, 128, 192]))
    bs = int(rng.choice([1, 3, 5, 7, 9, 11, 13, 17, 4, 6, 12, 16]))
    minD = int(rng.choice([0, 0, 2, -3]))
    W, H = int(rng.integers(D + 30, D + 150)), int(rng.integers(12, 60))
    kw = dict(blockSize=bs, minDisparity=minD, uniquenessRatio=int(rng.choice([10, 0, 30, -1])),
              speckleWindowSize=int(rng.choice([100, 0, 15])), speckleRange=int(rng.choice([32, 1, 2])),
              disp12MaxDiff=int(rng.choice([1, -1, 0, 2])), P1=int(rng.choice([600, 8, 100, 0])),
              P2=int(rng.choice([2400, 700, 3000, 0])), paths=int(rng.choice([8, 5])))
    L, R = synth.make_pair(synth.STREAM_SEED + 7000 + seed, W, H, D)
    want = oracle.sgm_compute(L, R, numDisparities=D, **kw)
    m = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, **kw)
    got = m.compute(L, R)
    m.close()
    assert np.array_equal(got, want), (seed, W, H, D, kw, int((got != want).sum()))


FAST_TABLE = [(D, np_) for D in range(16, 257, 16) for np_ in (2, 3, 4)] + \
             [(D, np_) for D in range(16, 193, 16) for np_ in (5, 6)]


@pytest.mark.parametrize("D,pieces", FAST_TABLE)
def test_every_fast_instantiation(pkg, oracle, synth, D, pieces):
    # one window size per instantiation (pieces = ceil(w / 4)), alternating the 1- and 3-byte tails
    w = 4 * (pieces - 1) + (1 if (D // 16 + pieces) % 2 else 3)
    W, H = D + 4 * w + 150, w + 37
    L, R = synth.make_pair(synth.STREAM_SEED + 9000 + D + pieces, W, H, D)
    kw = dict(numDisparities=D, blockSize=w, preFilterCap=31 if 62 * w * w <= 32766 else 15)
    pkg.binding.lib().rtdm_debug_search_kernel(0)          # this test is about k_search_fast
    try:
        m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, preFilterCap=kw["preFilterCap"], width=W, height=H)
        got = m.compute(L, R)
        assert m.search_variant == "fast_qsad", (D, w, m.search_variant)
        m.close()
    finally:
        pkg.binding.lib().rtdm_debug_search_kernel(-1)
    assert np.array_equal(got, oracle.bm_compute(L, R, **kw)), (D, w)


RING_TABLE = [(64, 9, 2), (64, 7, 2), (64, 5, 2), (32, 7, 2), (32, 9, 2), (32, 11, 2), (32, 13, 2), (48, 7, 2), (48, 9, 2),
              (16, 5, 2), (16, 7, 2), (16, 9, 2), (64, 9, 4), (64, 7, 4), (64, 5, 4), (64, 11, 4), (64, 13, 4),
              (128, 7, 8), (128, 9, 8), (128, 11, 8), (128, 13, 8), (96, 7, 4), (96, 9, 4), (96, 11, 4), (96, 13, 4),
              (48, 11, 2), (48, 13, 2), (16, 11, 2), (16, 13, 2), (32, 5, 2), (32, 15, 2), (48, 5, 2), (64, 15, 4), (128, 15, 8),
              # round 3: rows per group < lanes per pixel (D = 192: eight lanes, four rows; D = 256: sixteen lanes, four rows)
              (192, 9, 8), (192, 11, 8), (192, 13, 8), (192, 15, 8), (256, 9, 16), (256, 11, 16), (256, 13, 16), (256, 15, 16)]


@pytest.mark.parametrize("D,w,lpp", RING_TABLE)
def test_every_ring_instantiation(pkg, oracle, synth, D, w, lpp):
    # k_search_ring (prefix sums in a register ring; two or four lanes per pixel): odd and even row counts and every
    # remainder of the row groups, a strip boundary inside the frame (rows > the 16-bit cap of a strip at cap 63), ROI,
    # negative and positive minDisparity, thresholds off
    pkg.binding.lib().rtdm_debug_search_kernel(lpp)
    try:
        for k, (W, H, kw) in enumerate([
                (D + 4 * w + 150, w + 37, {}),
                (D + 300, 2 * w + 120, dict(preFilterCap=63 if 126 * w * w <= 32766 else 31)),
                (D + 260, w + 46, dict(minDisparity=-3, uniquenessRatio=0, textureThreshold=0)),
                (D + 333, w + 51, dict(minDisparity=5, disp12MaxDiff=-1, speckleWindowSize=0)),
                (D + 400, 131, dict(roi1=(D + 20, 9, 250, 90))),
                (D + 70, w + 40, dict(uniquenessRatio=25)), (D + 197, w + 41, {}), (D + 64 + 2 * (w // 2), w + 42, {})]):
            L, R = synth.make_pair(synth.STREAM_SEED + 9500 + D + w + k, W, H, D)
            roi1 = kw.pop("roi1", None)
            m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, **kw)
            if roi1: m.setROI1(roi1)
            got = m.compute(L, R)
            assert m.search_variant == {2: "fast_ring_qsad", 4: "fast_ring4_qsad", 8: "fast_ring8_qsad", 16: "fast_ring16_qsad"}[lpp], (D, w, m.search_variant)
            m.close()
            okw = dict(kw); okw.update(numDisparities=D, blockSize=w)
            if roi1: okw["roi1"] = roi1
            want = oracle.bm_compute(L, R, **okw)
            assert np.array_equal(got, want), (D, w, k, int((got != want).sum()))
    finally:
        pkg.binding.lib().rtdm_debug_search_kernel(-1)


@pytest.mark.parametrize("W,H,D,w", [(4096, 48, 64, 9), (4095, 31, 128, 11), (90, 700, 64, 9), (70, 10, 64, 9), (25, 25, 16, 5),
                                     (300, 6, 16, 5), (2050, 70, 256, 15), (1920, 1080, 64, 9)])
def test_extreme_shapes(pkg, oracle, synth, W, H, D, w):
    # widest supported row (4096: whole rows live in LDS), tall-and-narrow, barely larger than the window,
    # fewer columns than disparities (all FILTERED), 1080p
    L, R = synth.make_pair(synth.STREAM_SEED + 12000 + W + H, W, H, D)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H)
    got = m.compute(L, R)
    m.close()
    assert np.array_equal(got, oracle.bm_compute(L, R, numDisparities=D, blockSize=w, nthreads=8)), (W, H, D, w)


@pytest.mark.parametrize("W,H", [(328, 200), (333, 201), (336, 57), (1288, 64)])
def test_device_tensors_of_every_alignment_class(pkg, oracle, synth, W, H):
    # contiguous device tensors: W % 16 == 0 takes the 128-bit row kernels, W % 8 == 0 the 64-bit prefilter,
    # anything else the byte-wise prefilter / scalar left-right check / unvectorised speckle merge
    import torch
    n, D, w = 3, 32, 7
    L, R = synth.make_stream(77 + W, n, W, H, D)
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=n)
    m.compute_device(dL, dR, dD, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = dD.cpu().numpy()
    m.close()
    for i in range(n):
        assert np.array_equal(got[i], oracle.bm_compute(L[i], R[i], numDisparities=D, blockSize=w, nthreads=8)), (W, H, i)


def test_batches_large_enough_to_be_autotuned(pkg, oracle, synth):
    # batches of >= 16 frames time several row-strip counts on the first call (rtdm_api.hip, tune_strips); whatever is
    # chosen, every frame must still be the oracle's, on the tuning call and on the calls after it
    import torch
    n, W, H, D, w = 24, 400, 150, 64, 9
    L, R = synth.make_stream(4242, n, W, H, D)
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=n)
    outs = []
    for _ in range(3):
        dD = torch.full((n, H, W), 12345, dtype=torch.int16, device="cuda")
        m.compute_device(dL, dR, dD, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(dD.cpu().numpy())
    m.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])
    for i in (0, 7, 23):
        assert np.array_equal(outs[0][i], oracle.bm_compute(L[i], R[i], numDisparities=D, blockSize=w, nthreads=8)), i


@pytest.mark.parametrize("seed", range(16))
def test_random_morphology_shapes(pkg, oracle, seed):
    # the filter works on 64x32 tiles with a 20/16 halo and four-pixel groups: sizes below, at and across those
    # granularities, blobby masks (what the HSV threshold delivers) and arbitrary gray images, aligned and odd pitches
    import torch
    from scipy import ndimage as ndi
    rng = np.random.default_rng(500 + seed)
    W = int(rng.choice([1, 3, 4, 17, 63, 64, 65, 100, 128, 131, 200, 256, 257, 300]))
    H = int(rng.choice([1, 2, 9, 31, 32, 33, 50, 64, 70, 99]))
    field = ndi.uniform_filter(rng.random((3, H, W)), (0, min(9, H), min(9, W)))
    imgs = np.stack([(field[0] > np.median(field[0])).astype(np.uint8) * 255,
                     rng.integers(0, 256, (H, W), dtype=np.uint8),
                     (field[2] * 255).astype(np.uint8)])
    mf = pkg.HIPMorphologicalFilter(W, H, 8, max_batch=2)
    d_in = torch.from_numpy(imgs).cuda(); d_out = torch.empty_like(d_in)
    mf.run_device(d_in, d_out, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], oracle.morph_open_close(imgs[i])), (seed, W, H, i)
    assert np.array_equal(mf.run(imgs[1]), got[1])              # host path = device path
    mf.close()


def test_large_batch_every_position_is_served(pkg, oracle, synth):
    # the search kernel decodes (tile, strip, frame) from a 1-D workgroup id: frames at the start, in the middle and at
    # the end of a batch of 96, and the frames of a second call with a different batch size on the same handle
    import torch
    n, W, H, D, w = 96, 640, 360, 64, 9
    L, R = synth.make_stream(999, n, W, H, D)
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=n)
    st = torch.cuda.current_stream().cuda_stream
    m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize()
    got = dD.cpu().numpy()
    want = {i: oracle.bm_compute(L[i], R[i], numDisparities=D, blockSize=w, nthreads=8) for i in (0, 1, 37, 94, 95)}
    for i, wnt in want.items():
        assert np.array_equal(got[i], wnt), i
    dD2 = torch.empty((40, H, W), dtype=torch.int16, device="cuda")
    m.compute_device(dL[50:90].contiguous(), dR[50:90].contiguous(), dD2, st)
    torch.cuda.synchronize()
    assert torch.equal(dD2, dD[50:90])
    m.close()


def test_too_wide_is_refused_not_crashed(pkg):
    with pytest.raises(pkg.binding.RtdmError) as e:
        pkg.HIPMatcher(numOfDisparities=64, blockSize=9, width=4097, height=32)
    assert e.value.status == -6

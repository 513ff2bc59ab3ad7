"""GPU parity suite (-m gpu): every case goes through the C ABI (rtdm_bm_* / rtdm_morph_*) and is
compared BIT-FOR-BIT with the CPU oracle on the same seeded inputs, with the committed golden
vectors, and -- at BASELINE sizes -- through size-independent properties.  Integer path: the bar is
exact equality, no tolerance anywhere.  (The oracle itself is parity-unpinned against the real
OpenCV-backed SWMatcherKonolige, see oracle/rtdm_oracle.h.)"""
import numpy as np
import pytest

import golden_util as gu
from conftest import load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "the -m gpu suite needs an MI355X"
    return load()


def run_hip(pkg, L, R, roi1=None, roi2=None, **kw):
    H, W = L.shape
    m = pkg.HIPMatcher(numOfDisparities=kw.pop("numDisparities"), width=W, height=H, **kw)
    if roi1: m.setROI1(roi1)
    if roi2: m.setROI2(roi2)
    out = m.compute(L, R)
    m.close()
    return out


def both(pkg, oracle, L, R, **kw):
    want = oracle.bm_compute(L, R, **kw)
    got = run_hip(pkg, L, R, **kw)
    return got, want


def assert_same(got, want):
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError("%d / %d pixels differ; first at (y,x)=%s got %d want %d" % (
            len(bad), got.size, tuple(bad[0]), got[tuple(bad[0])], want[tuple(bad[0])]))


# ---- golden vectors --------------------------------------------------------------------------
@pytest.mark.parametrize("name", gu.bm_cases())
def test_hip_reproduces_bm_golden(pkg, name):
    L, R, disp, kw = gu.load_bm(name)
    assert_same(run_hip(pkg, L, R, **kw), disp)


@pytest.mark.parametrize("name", gu.morph_cases())
def test_hip_reproduces_morph_golden(pkg, name):
    z = gu.load_morph(name)
    H, W = z["mask"].shape
    mf = pkg.HIPMorphologicalFilter(W, H, 8)
    assert np.array_equal(mf.run(z["mask"]), z["mask_out"])
    assert np.array_equal(mf.run(z["gray"]), z["gray_out"])


# ---- staged parity: A.3 alone, + A.4, + A.5 --------------------------------------------------
CASES = [
    (64, 48, 16, 5, 0), (96, 64, 32, 7, 0), (97, 65, 16, 9, 0), (80, 50, 16, 13, 0), (131, 77, 48, 11, 0),
    (96, 64, 16, 7, 4), (96, 64, 16, 7, -5), (90, 40, 32, 5, -40), (200, 90, 96, 9, 0), (72, 60, 16, 21, 0),
    (300, 70, 128, 7, 0), (90, 80, 16, 33, 0),
]


@pytest.mark.parametrize("W,H,D,w,minD", CASES)
def test_search_only(pkg, oracle, synth, W, H, D, w, minD):
    L, R = synth.make_pair(synth.STREAM_SEED + W + H, W, H, D)
    assert_same(*both(pkg, oracle, L, R, numDisparities=D, blockSize=w, minDisparity=minD,
                      disp12MaxDiff=-1, speckleWindowSize=0))


@pytest.mark.parametrize("W,H,D,w,minD", CASES)
def test_search_plus_lrcheck(pkg, oracle, synth, W, H, D, w, minD):
    L, R = synth.make_pair(synth.STREAM_SEED + W + H + 7, W, H, D)
    for md in (0, 1):
        assert_same(*both(pkg, oracle, L, R, numDisparities=D, blockSize=w, minDisparity=minD,
                          disp12MaxDiff=md, speckleWindowSize=0))


@pytest.mark.parametrize("W,H,D,w,minD", CASES)
def test_full_pipeline(pkg, oracle, synth, W, H, D, w, minD):
    L, R = synth.make_pair(synth.STREAM_SEED + W + H + 13, W, H, D)
    assert_same(*both(pkg, oracle, L, R, numDisparities=D, blockSize=w, minDisparity=minD,
                      speckleWindowSize=25, speckleRange=32))


@pytest.mark.parametrize("uniq,tex,cap", [(0, 0, 31), (0, 50, 31), (15, 10, 63), (40, 0, 1), (300, 10, 15)])
def test_rejection_thresholds(pkg, oracle, synth, uniq, tex, cap):
    L, R = synth.make_pair(synth.STREAM_SEED + 99, 96, 64, 16)
    assert_same(*both(pkg, oracle, L, R, numDisparities=16, blockSize=7, uniquenessRatio=uniq,
                      textureThreshold=tex, preFilterCap=cap, speckleWindowSize=0, disp12MaxDiff=-1))


# ---- BASELINE configs ------------------------------------------------------------------------
def test_config1_320x240_d32_w7_and_roi_crop(pkg, oracle, synth):
    # backup/320x240/extrinsics.yml:56-57 + main.cpp:80-85 -> the 233x156 crop at (49,46), pitch 320
    L, R = synth.make_pair(synth.STREAM_SEED, 320, 240, 32)
    assert_same(*both(pkg, oracle, L, R, numDisparities=32, blockSize=7))
    lv, rv = L[46:46 + 156, 49:49 + 233], R[46:46 + 156, 49:49 + 233]
    assert not lv.flags["C_CONTIGUOUS"]
    assert_same(*both(pkg, oracle, lv, rv, numDisparities=32, blockSize=7))
    # estimator.cpp:54: setROI1(matching_roi) before every compute; ROI2 stays unset
    roi = (60, 40, 120, 80)
    assert_same(*both(pkg, oracle, lv, rv, numDisparities=32, blockSize=7, roi1=roi))


@pytest.mark.parametrize("roi", [(200, 100, 180, 150), (70, 10, 560, 460), (0, 0, 300, 200), (500, 300, 140, 180),
                                 (66, 4, 10, 12), (320, 200, 64, 64)])
def test_roi_column_skipping_is_exact_on_fast_path(pkg, oracle, synth, roi):
    # estimator.cpp:54: the caller shrinks ROI1 to the union of the object boxes before every compute;
    # the search then only covers the columns that can influence the valid rectangle
    L, R = synth.make_pair(synth.STREAM_SEED + 31, 640, 480, 64)
    for md in (1, -1):
        assert_same(*both(pkg, oracle, L, R, numDisparities=64, blockSize=9, roi1=roi, disp12MaxDiff=md))
    assert_same(*both(pkg, oracle, L, R, numDisparities=64, blockSize=9, roi1=roi, roi2=(40, 20, 560, 400)))


def test_config2_640x480_d64_w9(pkg, oracle, synth):
    L, R = synth.make_pair(synth.STREAM_SEED + 1, 640, 480, 64)
    got, want = both(pkg, oracle, L, R, numDisparities=64, blockSize=9)
    assert_same(got, want)
    assert (got != -16).mean() > 0.5


def test_config3_1280x720_d128_w11_plus_morph(pkg, oracle, synth):
    L, R = synth.make_pair(synth.STREAM_SEED + 2, 1280, 720, 128)
    assert_same(*both(pkg, oracle, L, R, numDisparities=128, blockSize=11))
    mask = ((L > 120) * 255).astype(np.uint8)
    mf = pkg.HIPMorphologicalFilter(1280, 720, 8)
    assert np.array_equal(mf.run(mask), oracle.morph_open_close(mask))


def test_reference_default_1280x720_d192_w13(pkg, oracle, synth):
    # the reference's own flags: nd=192 (cmdline-parser.cpp:22), blockSize 13 (main.cpp:134)
    L, R = synth.make_pair(synth.STREAM_SEED + 3, 1280, 720, 192)
    assert_same(*both(pkg, oracle, L, R, numDisparities=192, blockSize=13))


def test_headline_1280x720_d64_w9_batch_device(pkg, oracle, synth):
    import torch
    n, W, H, D = 6, 1280, 720, 64
    L, R = synth.make_stream(10, n, W, H, D)
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=9, width=W, height=H, max_batch=4)   # forces 2 chunks
    m.compute_device(dL, dR, dD, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = dD.cpu().numpy()
    for i in (0, 3, 5):
        assert_same(got[i], oracle.bm_compute(L[i], R[i], numDisparities=D, blockSize=9, nthreads=8))
    # batch-position independence: frame 5 alone and in a permuted batch
    perm = [5, 0, 1, 2, 3, 4]
    dD2 = torch.empty_like(dD)
    m.compute_device(dL[perm].contiguous(), dR[perm].contiguous(), dD2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(dD2[0], dD[5]) and torch.equal(dD2[1], dD[0])
    # host batch entry point gives the same bytes
    assert np.array_equal(m.compute_batch(L, R), got)


# ---- known-answer properties on the device ---------------------------------------------------
def test_kat_constant_image_is_all_filtered(pkg):
    img = np.full((48, 64), 77, np.uint8)
    assert (run_hip(pkg, img, img, numDisparities=16, blockSize=5) == -16).all()


def test_kat_ties_resolve_to_largest_disparity(pkg, oracle):
    H, W, D = 40, 96, 16
    row = (np.array([10, 200, 90, 30])[np.arange(W) % 4]).astype(np.uint8)
    img = (np.tile(row, (H, 1)) + (np.arange(H)[:, None] % 3) * 7).astype(np.uint8)
    kw = dict(numDisparities=D, blockSize=5, uniquenessRatio=0, textureThreshold=0, speckleWindowSize=0, disp12MaxDiff=-1)
    got, want = both(pkg, oracle, img, img, **kw)
    assert_same(got, want)
    assert (((got[2:-2, D + 1:W - 3] + 8) >> 4) == 12).all()


def test_kat_too_narrow_image_is_all_filtered(pkg):
    img = np.random.default_rng(0).integers(0, 255, (40, 30), dtype=np.uint8)
    assert (run_hip(pkg, img, img, numDisparities=32, blockSize=5) == -16).all()


def test_kat_speckle_boundary_through_pipeline(pkg, oracle, synth):
    L, R = synth.make_pair(synth.STREAM_SEED + 21, 200, 120, 32)
    for win in (99, 100, 400):
        assert_same(*both(pkg, oracle, L, R, numDisparities=32, blockSize=9, speckleWindowSize=win))


def test_error_behaviour_matches_header(pkg, synth):
    L, R = synth.make_pair(synth.STREAM_SEED, 64, 48, 16)
    m = pkg.HIPMatcher(numOfDisparities=16, blockSize=5, width=64, height=48)
    big = np.zeros((60, 80), np.uint8)
    with pytest.raises(pkg.binding.RtdmError) as e:
        m.compute(big, big)
    assert e.value.status == -2
    m2 = pkg.HIPMatcher(numOfDisparities=16, blockSize=49, width=64, height=48)   # >= min(W,H): cv::StereoBM rejects at compute
    with pytest.raises(pkg.binding.RtdmError) as e:
        m2.compute(L, R)
    assert e.value.status == -1


# ---- morphology ------------------------------------------------------------------------------
@pytest.mark.parametrize("W,H", [(64, 48), (37, 29), (12, 9), (233, 156), (640, 480)])
def test_morph_matches_oracle(pkg, oracle, W, H):
    rng = np.random.default_rng(W * H)
    mf = pkg.HIPMorphologicalFilter(W, H, 8)
    mask = ((rng.random((H, W)) < 0.5) * 255).astype(np.uint8)
    gray = rng.integers(0, 256, (H, W), dtype=np.uint8)
    assert np.array_equal(mf.run(mask), oracle.morph_open_close(mask))
    assert np.array_equal(mf.run(gray), oracle.morph_open_close(gray))


def test_morph_video_buffers_round_trip(pkg, oracle):
    # estimator.cpp:141-142 wraps getVideoInBuffer()/getVideoOutBuffer() as Mats once and reuses them
    W, H = 233, 156
    mf = pkg.HIPMorphologicalFilter(W, H, 8)
    vin, vout = mf.getVideoInBuffer(), mf.getVideoOutBuffer()
    assert vin.shape == (H, W) and mf.getFrameSize() == W * H
    vin[:] = ((np.random.default_rng(3).random((H, W)) < 0.6) * 255).astype(np.uint8)
    mf.run(vin, vout)
    assert np.array_equal(vout, oracle.morph_open_close(vin))


# ---- synthetic stream ------------------------------------------------------------------------
@pytest.mark.parametrize("W,H,D", [(96, 64, 16), (320, 240, 32), (1280, 720, 64)])
def test_device_synth_is_bit_identical_to_numpy(pkg, synth, W, H, D):
    import torch
    n = 3
    dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
    pkg.synth_pairs_device(dL, dR, first_frame=5, numDisparities=D)
    L, R = synth.make_stream(5, n, W, H, D)
    assert np.array_equal(dL.cpu().numpy(), L) and np.array_equal(dR.cpu().numpy(), R)


def test_morph_device_batch_and_timing_shape(pkg, oracle):
    # device-resident batch through rtdm_morph_run_device: every frame equals the oracle's result
    import torch
    n, W, H = 5, 1280, 720
    rng = np.random.default_rng(11)
    masks = ((rng.random((n, H, W)) < 0.5) * 255).astype(np.uint8)
    masks[:, 100:300, 200:700] = 255
    d_in = torch.from_numpy(masks).cuda(); d_out = torch.empty_like(d_in)
    mf = pkg.HIPMorphologicalFilter(W, H, 8, max_batch=2)      # forces chunking
    mf.run_device(d_in, d_out, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], oracle.morph_open_close(masks[i])), i


# ---- cv::StereoSGBM (rows S / f4; BASELINE config 5) ----------------------------------------------
# Integer algorithm restated in oracle/sgm_oracle.c (MODE_SGBM = paths 5, MODE_HH = paths 8): the stated tolerance
# against that oracle is 0 (exact equality); against a real cv::StereoSGBM the comparison is unpinned.
@pytest.mark.parametrize("W,H,D,minD,bs", [(72, 28, 16, 0, 5), (200, 120, 32, 0, 5), (161, 75, 48, 0, 3), (150, 60, 32, 3, 5),
                                           (150, 60, 32, -4, 5), (300, 100, 128, 0, 5), (96, 40, 16, 0, 7), (180, 70, 32, 0, 13),
                                           (140, 50, 16, 0, 17)])
def test_sgm_matches_oracle(pkg, oracle, synth, W, H, D, minD, bs):
    L, R = synth.make_pair(synth.STREAM_SEED + 700 + W, W, H, D)
    for kw in (dict(), dict(paths=5), dict(disp12MaxDiff=-1, speckleWindowSize=0, paths=5),
               dict(uniquenessRatio=0, speckleWindowSize=30, speckleRange=2)):
        m = pkg.HIPSemiGlobalMatcher(blockSize=bs, minDisparity=minD, numOfDisparities=D, width=W, height=H, **kw)
        got = m.compute(L, R)
        m.close()
        assert_same(got, oracle.sgm_compute(L, R, blockSize=bs, minDisparity=minD, numDisparities=D, **kw))


def test_sgm_device_batch(pkg, oracle, synth):
    import torch
    n, W, H, D = 3, 320, 240, 64
    L, R = synth.make_stream(40, n, W, H, D)
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    m = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, max_batch=2)
    m.compute_device(dL, dR, dD, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i in range(n):
        assert_same(dD[i].cpu().numpy(), oracle.sgm_compute(L[i], R[i], numDisparities=D))


def test_device_api_is_ordered_on_the_callers_stream(pkg, synth):
    # torch's current stream is the HIP null stream (handle 0): work enqueued by rtdm_bm_compute_device
    # must be ordered against torch ops issued before and after it WITHOUT an explicit synchronise.
    import torch
    sh = load("sharding")
    W, H, D, N = 640, 480, 64, 12
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    left = torch.empty((N, H, W), dtype=torch.uint8, device=dev); right = torch.empty_like(left)
    pkg.synth_pairs_device(left, right, first_frame=0, numDisparities=D)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=9, width=W, height=H, max_batch=8)
    ref = torch.empty((N, H, W), dtype=torch.int16, device=dev)
    m.compute_device(left, right, ref, st)
    torch.cuda.synchronize()

    def compute(L, R):
        out = torch.empty(L.shape, dtype=torch.int16, device=dev)
        m.compute_device(L.contiguous(), R.contiguous(), out, st)
        return out                                   # consumed by torch copies right away, no sync

    class Work:                                       # what an asynchronous collective hands back: wait() orders the
        def __init__(self):                           # CURRENT stream behind the stream the collective was issued from
            self.ev = torch.cuda.Event(); self.ev.record(torch.cuda.current_stream())
        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    class Solo:                                       # world of one without a process group (the real RCCL group of size
        get_world_size = staticmethod(lambda: 1)      # one is exercised by tests/test_gpu_round2.py through bench.py)
        get_rank = staticmethod(lambda: 0)
        scatter = staticmethod(lambda t, l, src=0, async_op=False: (t.copy_(l[0]), Work())[1])
        gather = staticmethod(lambda t, l, dst=0, async_op=False: (l[0].copy_(t), Work())[1])
    for chunk in (3, None):
        out = sh.scatter_compute_gather(Solo, left, right, N, (H, W), compute, dev, chunk=chunk)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


# ---- next row: /= 16 + reprojectImageTo3D + calc_depth on the device ---------
# Floating point: Z is a float from a double quotient, the mean a double sum.  The device sums in a
# different (fixed) order than the oracle, hence a tolerance: 1e-9 relative on the mean, counts exact.
Q_TEST = np.array([[1, 0, 0, -640.3], [0, 1, 0, -360.8], [0, 0, 0, 700.25], [0, 0, 1 / 12.0, 0.0]])


def test_compute_depth_matches_oracle(pkg, oracle, synth):
    W, H, D = 1280, 720, 64
    L, R = synth.make_pair(synth.STREAM_SEED + 77, W, H, D)
    mask = ((L > 100) * 255).astype(np.uint8)
    regions = [(200, 150, 300, 220), (0, 0, W, H), (900, 400, 120, 200), (10, 10, 1, 1), (600, 700, 80, 0)]
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=9, width=W, height=H)
    mean, cnt, disp = m.compute_depth(L, R, Q_TEST, mask, regions, calibration_unit=25.0, want_disp=True)
    want_disp = oracle.bm_compute(L, R, numDisparities=D, blockSize=9, nthreads=8)
    assert_same(disp, want_disp)
    wm, wc = oracle.depth_stats(want_disp, Q_TEST, mask, regions, 25.0)
    assert np.array_equal(cnt, wc) and wc[1] > 100000
    assert np.allclose(mean, wm, rtol=1e-9, atol=0)
    mean2, cnt2 = m.compute_depth(L, R, Q_TEST, mask, regions)            # reproducible run to run, bit for bit
    assert np.array_equal(mean, mean2) and np.array_equal(cnt, cnt2)
    with pytest.raises(pkg.binding.RtdmError):
        m.compute_depth(L, R, Q_TEST, mask, [(1200, 700, 100, 100)])


def test_depth_stats_device_matches_oracle(pkg, oracle, synth):
    import torch
    L, R = synth.make_pair(synth.STREAM_SEED + 78, 320, 240, 32)
    d = oracle.bm_compute(L, R, numDisparities=32, blockSize=9)
    mask = ((L > 110) * 255).astype(np.uint8)
    Qs = np.array([[1, 0, 0, -160.5], [0, 1, 0, -120.25], [0, 0, 0, 310.7], [0, 0, 1 / 2.4, 0.0]])
    regions = [(40, 30, 100, 80), (0, 0, 320, 240), (317, 200, 3, 40)]
    mean, cnt = pkg.depth_stats_device(torch.from_numpy(d).cuda(), Qs, torch.from_numpy(mask).cuda(), regions)
    wm, wc = oracle.depth_stats(d, Qs, mask, regions)
    assert np.array_equal(cnt, wc) and np.allclose(mean, wm, rtol=1e-9, atol=0)


# ---- the C++ host layer (HIPMatcherCore / HIPRectifierCore) on the device ---------------------
def test_cpp_host_layer_on_the_gpu(oracle, synth, tmp_path):
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "rt-depth-map_amd", "lib", "host_selftest")
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "ctor_status=0" in out.stdout and "rectifier_status=0 rectify_status=0" in out.stdout, out.stdout + out.stderr
    W, H, D, w = 400, 220, 64, 9
    L, R = synth.make_pair(synth.STREAM_SEED + 31, W, H, D)
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    src.write_bytes(L.tobytes() + R.tobytes())
    rc = subprocess.run([exe, str(src), str(dst), str(W), str(H), str(D), str(w)], capture_output=True, text=True)
    assert rc.returncode == 0, rc.stdout + rc.stderr
    got = np.frombuffer(dst.read_bytes(), np.int16).reshape(H, W)
    want = oracle.bm_compute(L, R, preFilterCap=31, blockSize=w, minDisparity=0, textureThreshold=10, numDisparities=D,
                             uniquenessRatio=10, speckleWindowSize=100, speckleRange=32, disp12MaxDiff=1, nthreads=8)
    assert np.array_equal(got, want)

"""The cv::StereoSGBM restatement (oracle/sgm_oracle.c: MODE_SGBM = paths 5, what sgbm-sw.cpp:15 creates; MODE_HH =
paths 8, BASELINE config 5) against the independent numpy implementation, stage by stage, plus known answers for the
rules that are easy to get wrong (R1, R5, R7, R9, R10 of that file's header).  PARITY UNPINNED against the library
itself: OpenCV is absent and the reference holds no fixtures."""
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
                                dict(minDisparity=2), dict(minDisparity=-3), dict(speckleWindowSize=20, speckleRange=2),
                                dict(P1=8, P2=32), dict(P1=0, P2=0), dict(uniquenessRatio=-1, disp12MaxDiff=3),
                                dict(paths=5), dict(paths=5, blockSize=9), dict(paths=5, blockSize=13, speckleWindowSize=0)])
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
    for bad in (dict(numDisparities=20), dict(blockSize=0), dict(uniquenessRatio=101), dict(blockSize=257)):
        kw = dict(numDisparities=16); kw.update(bad)
        with pytest.raises(ValueError):
            oracle.sgm_compute(L, R, **kw)
    # R12: P1 <= 0 -> 2, P2 <= 0 -> 5, P2 >= P1 + 1: coerced, not refused
    a = oracle.sgm_compute(L, R, numDisparities=16, P1=0, P2=0)
    assert np.array_equal(a, oracle.sgm_compute(L, R, numDisparities=16, P1=2, P2=5))
    assert np.array_equal(oracle.sgm_compute(L, R, numDisparities=16, P1=10, P2=10), oracle.sgm_compute(L, R, numDisparities=16, P1=10, P2=11))


def test_even_block_size_is_the_next_odd_one(oracle, synth):
    # cv::StereoSGBM never checks the parity of SADWindowSize: SW2 = SH2 = SADWindowSize / 2 (sgbm-sw.cpp:15 passes the
    # caller's blockSize through unchanged); the second implementation (tests/bruteforce.py: r = blockSize // 2) agrees
    L, R = synth.make_pair(synth.STREAM_SEED + 3, 72, 24, 16)
    for even in (4, 8):
        a = oracle.sgm_compute(L, R, numDisparities=16, blockSize=even)
        assert np.array_equal(a, oracle.sgm_compute(L, R, numDisparities=16, blockSize=even + 1))
        assert np.array_equal(a, bf.sgm(L, R, numDisparities=16, blockSize=even))


def overflow_pair(W=96, H=40, D=16):
    """Every pixel pair at the maximum intensity term of the pixel cost (255 >> 2 = 63; the sampling-insensitive measure
    needs flat images for that: any structure brings a half-way point near the other image's value)."""
    return np.full((H, W), 255, np.uint8), np.zeros((H, W), np.uint8)


def test_windows_above_17_run_until_a_cost_would_wrap(oracle, synth):
    # deviation (a) of sgm_oracle.c: path cost <= block cost + P2; the library's short arithmetic wraps above 32767 and that is
    # not restated.  Windows that can get there are no longer refused outright: the frame is, if one of its block costs does.
    L, R = synth.make_pair(synth.STREAM_SEED + 5, 90, 44, 16)
    for bs in (19, 21, 20):
        a = oracle.sgm_compute(L, R, numDisparities=16, blockSize=bs)
        assert np.array_equal(a, bf.sgm(L, R, numDisparities=16, blockSize=bs)), bs
    Lo, Ro = overflow_pair()
    pix, Cc, _ = oracle.sgm_stages(Lo, Ro, numDisparities=16, blockSize=25)
    assert int(pix.max()) * 25 * 25 + 2400 > 32767                      # (the uint16 block costs themselves may have wrapped)
    with pytest.raises(ValueError):
        oracle.sgm_compute(Lo, Ro, numDisparities=16, blockSize=25)
    oracle.sgm_compute(Lo, Ro, numDisparities=16, blockSize=17)        # 93 * 17^2 + 2400 < 32767: can never overflow


def test_kat_left_right_check_cannot_be_switched_off(oracle, synth):
    # R9: disp12MaxDiff <= 0 means 1
    L, R = synth.make_pair(synth.STREAM_SEED + 620, 96, 30, 16)
    a = oracle.sgm_compute(L, R, numDisparities=16, disp12MaxDiff=-1)
    assert np.array_equal(a, oracle.sgm_compute(L, R, numDisparities=16, disp12MaxDiff=0))
    assert np.array_equal(a, oracle.sgm_compute(L, R, numDisparities=16, disp12MaxDiff=1))


def test_kat_median_of_nine_with_replicated_border(oracle):
    # R10: an isolated spike disappears, a 2-pixel-wide vertical bar survives, the corner uses the clamped neighbourhood
    img = np.zeros((6, 8), np.int16)
    img[3, 3] = 500
    img[:, 6:8] = 77
    img[0, 0] = -16
    out = oracle.median3x3(img)
    assert out[3, 3] == 0 and (out[:, 6:8] == 77).all() and out[0, 0] == 0
    col = np.arange(12, dtype=np.int16).reshape(12, 1) * 16
    assert np.array_equal(oracle.median3x3(col), col)            # a single column: median of three replicated triples
    rng = np.random.default_rng(5)
    big = rng.integers(-16, 2000, (37, 53)).astype(np.int16)
    assert np.array_equal(oracle.median3x3(big), bf.median3x3(big).astype(np.int16))


def test_kat_raw_border_columns_are_ftzero(oracle):
    # R1: with flat gradients (constant rows) the only cost comes from the raw intensities; a left pixel in the LAST column
    # is compared as if its intensity were 15
    W, H, D = 40, 6, 16
    L = np.full((H, W), 200, np.uint8); R = np.full((H, W), 200, np.uint8)
    pix, _, _ = oracle.sgm_stages(L, R, numDisparities=D)
    assert (pix[:, :-1, :] == 0).all()                           # interior: identical intensities
    # x = W-1: u = 15 with half-way points (15+200)//2 = 107 towards the left neighbour; v = 200: cost min(200-107, ...) >> 2
    assert (pix[:, -1, 1:] == ((200 - 107) >> 2)).all()
    assert (pix[:, -1, 0] == 0).all()                            # d = 0: the right pixel is column W-1 as well, also 15


def test_kat_sums_saturate_at_short_max(oracle):
    # R5: S = min(sum of the path costs, 32767)
    C = np.full((4, 5, 16), 9000, np.uint16)
    import ctypes as Ct
    S = np.zeros_like(C)
    oracle.lib().orc_sgm_aggregate_paths(C.ctypes.data_as(Ct.POINTER(Ct.c_uint16)), 5, 4, 16, 600, 2400, 5,
                                          S.ctypes.data_as(Ct.POINTER(Ct.c_uint16)))
    assert (S == 32767).all()                                    # five paths of >= 9000 each

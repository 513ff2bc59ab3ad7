"""GPU suite, round-3 additions (-m gpu).  Everything goes through the C ABI and is compared bit for bit with the CPU
oracle (parity unpinned against OpenCV itself: oracle/rtdm_oracle.h)."""
import os
import threading

import numpy as np
import pytest

from conftest import load, ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "the -m gpu suite needs an MI355X"
    return load()


def assert_same(got, want):
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError("%d / %d pixels differ; first at (y,x)=%s got %d want %d" % (
            len(bad), got.size, tuple(bad[0]), got[tuple(bad[0])], want[tuple(bad[0])]))


# ---- guard bands: a view's pad columns belong to the caller (estimator.cpp:33,36 passes views) ------------------------
# The internal disparity plane has rows of Ws = width rounded up to 8; a caller whose pitch happens to equal Ws * 2 must
# still only see its `width` columns written.
def test_pageable_view_whose_pitch_equals_the_internal_pitch_keeps_its_pad_columns(pkg, oracle, synth):
    W, H, D, w = 233, 156, 32, 7                                # 233 -> Ws = 240
    Lf, Rf = synth.make_pair(synth.STREAM_SEED + 7, 240, H, D)
    L, R = Lf[:, :W], Rf[:, :W]                                 # views at x = 0 of 240-wide planes
    plane = np.full((H, 240), 777, np.int16)
    O = plane[:, :W]                                            # pitch 480 == Ws * 2
    assert O.strides[0] == 480
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H)
    m.compute(L, R, O)
    m.close()
    assert_same(O, oracle.bm_compute(np.ascontiguousarray(L), np.ascontiguousarray(R), numDisparities=D, blockSize=w))
    assert (plane[:, W:] == 777).all()                          # columns [233, 240) untouched


@pytest.mark.parametrize("pinned", [False, True])
def test_host_batch_with_240_wide_buffers_at_width_233_keeps_its_pad_columns(pkg, oracle, synth, pinned):
    import torch
    n, W, H, D, w = 5, 233, 120, 32, 7
    Lf, Rf = synth.make_stream(91, n, 240, H, D)
    planes = torch.full((n, H, 240), 777, dtype=torch.int16)
    if pinned:
        Lt, Rt, planes = torch.from_numpy(Lf).pin_memory(), torch.from_numpy(Rf).pin_memory(), planes.pin_memory()
        Lf, Rf = Lt.numpy(), Rt.numpy()
    out = planes.numpy()
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=4)
    m.compute_batch(Lf[:, :, :W], Rf[:, :, :W], out[:, :, :W])
    m.close()
    for i in range(n):
        assert_same(out[i, :, :W], oracle.bm_compute(np.ascontiguousarray(Lf[i, :, :W]), np.ascontiguousarray(Rf[i, :, :W]),
                                                     numDisparities=D, blockSize=w))
    assert (out[:, :, W:] == 777).all()


# ---- rtdm_bm_params.legacy_right_clamp: the right-border rule of the reference's OpenCV era (oracle hazard H1) ----------
# Compared with the oracle's own legacy switch (oracle/bm_oracle.c: rbase clamp W-rofs-1, a plane of step W, zeros after
# the last row).  Still parity-unpinned: the switch restates the 3.x source from memory, no OpenCV here to run.
def _legacy_case(rng):
    D = int(rng.choice([16, 32, 48, 64, 96, 128, 192, 256]))
    w = int(rng.choice([5, 7, 9, 11, 13, 15, 21, 25]))          # 25: the generic kernel; 21: the border kernel beside k_search_fast
    minD = int(rng.choice([0, 0, 0, 3, -7]))
    W = int(rng.integers(D + abs(minD) + w + 20, D + abs(minD) + w + 200))
    H = int(rng.integers(w + 3, w + 60))
    kw = dict(numDisparities=D, blockSize=w, minDisparity=minD, preFilterCap=int(rng.choice([31, 31, 15])),
              textureThreshold=int(rng.choice([10, 0])), uniquenessRatio=int(rng.choice([10, 0, 15])),
              speckleWindowSize=int(rng.choice([100, 0])), speckleRange=32, disp12MaxDiff=int(rng.choice([1, 1, -1, 0, 3])))
    return W, H, kw


def _hip_kw(kw):
    k = dict(kw)
    k["numOfDisparities"] = k.pop("numDisparities")
    return k


@pytest.mark.parametrize("seed", range(24))
def test_legacy_right_clamp_matches_the_oracles_legacy_switch(pkg, oracle, synth, seed):
    rng = np.random.default_rng(3000 + seed)
    W, H, kw = _legacy_case(rng)
    L, R = synth.make_pair(synth.STREAM_SEED + 11000 + seed, W, H, kw["numDisparities"])
    if seed % 4 == 0:                       # pitched views: the device must wrap at W, not at its own pitch
        pl, pr = np.zeros((H + 2, W + 29), np.uint8), np.zeros((H + 2, W + 29), np.uint8)
        pl[1:1 + H, 3:3 + W] = L; pr[1:1 + H, 3:3 + W] = R
        L, R = pl[1:1 + H, 3:3 + W], pr[1:1 + H, 3:3 + W]
    base = oracle.bm_compute(L, R, **kw)
    oracle.set_legacy_right_clamp(True)
    try:
        want = oracle.bm_compute(L, R, **kw)
    finally:
        oracle.set_legacy_right_clamp(False)
    m = pkg.HIPMatcher(width=W, height=H, legacy_right_clamp=1, **_hip_kw(kw))
    got = m.compute(L, R)
    variant = m.search_variant
    m.close()
    assert_same(got, want)
    m0 = pkg.HIPMatcher(width=W, height=H, **_hip_kw(kw))               # the default stays the 4.x rule
    assert_same(m0.compute(L, R), base)
    m0.close()
    assert variant


def test_legacy_right_clamp_is_live_and_confined_on_the_golden_shape(pkg, oracle, synth):
    # the shape of tests/test_oracle_bm.py::test_right_clamp_hazard_is_confined, batched through the device entry point
    import torch
    n, W, H, D, w = 20, 320, 240, 32, 9
    Ls, Rs = synth.make_stream(55, n, W, H, D)
    dL, dR = torch.from_numpy(Ls).cuda(), torch.from_numpy(Rs).cuda()
    outs = {}
    for flag in (0, 1):
        m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=n, legacy_right_clamp=flag)
        dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
        m.compute_device(dL, dR, dD, torch.cuda.current_stream().cuda_stream)      # n >= 16: border kernel on the side stream
        torch.cuda.synchronize()
        outs[flag] = dD.cpu().numpy()
        m.close()
    oracle.set_legacy_right_clamp(True)
    try:
        for i in (0, 7, n - 1):
            assert_same(outs[1][i], oracle.bm_compute(Ls[i], Rs[i], numDisparities=D, blockSize=w))
    finally:
        oracle.set_legacy_right_clamp(False)
    for i in (0, 7, n - 1):
        assert_same(outs[0][i], oracle.bm_compute(Ls[i], Rs[i], numDisparities=D, blockSize=w))
    diff = outs[0] != outs[1]
    assert not diff[:, :, :W - w // 2 - D - 1].any()                   # H1: only votes near the right edge can differ


def test_legacy_right_clamp_value_is_validated(pkg):
    from importlib import import_module
    B = import_module("rt-depth-map_amd.binding")
    with pytest.raises(B.RtdmError) as e:
        pkg.HIPMatcher(numOfDisparities=32, blockSize=9, width=128, height=64, legacy_right_clamp=2)
    assert e.value.status == -1


# ---- the strip-count tuner is keyed on the SHAPE of the work: counted, not timed -------------------------------------
def test_roi_moving_batched_caller_never_triggers_the_tuner(pkg, synth):
    import torch
    n, W, H, D = 16, 640, 480, 64
    dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
    dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    pkg.synth_pairs_device(dL, dR, first_frame=0, numDisparities=D, stream=st)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=9, width=W, height=H, max_batch=n)
    for k in range(50):                      # 50 distinct ROI sizes: every shape is seen once, none is measured
        m.setROI1((80 + k, 40 + (k % 7), 200 + 5 * k, 150 + 3 * k))
        m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize()
    assert m.tuner_stats() == (0, 0)
    for k in range(6):                       # one size at six positions: ONE shape, measured once (on its second sighting)
        m.setROI1((140 + 11 * k, 30 + 5 * k, 300, 200))         # (far enough from the borders that nothing is clipped)
        m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize()
    shapes, launches = m.tuner_stats()
    assert shapes == 1 and 0 < launches <= 30
    m.setROI1((0, 0, 0, 0))
    for _ in range(3):
        m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize()
    assert m.tuner_stats()[0] == 2           # the full frame is a second shape
    m.close()


# ---- two handles driven from two host threads (process-wide state is atomics / once-flags only) ------------------------
def test_two_handles_on_two_threads_give_the_single_thread_bytes(pkg, oracle, synth):
    import torch
    cfgs = [(400, 150, 64, 9, 22), (333, 120, 128, 11, 18)]      # both take k_search_ring forms that raise their LDS limit
    want, errs, got = {}, [], {}

    def run(k):
        try:
            W, H, D, w, n = cfgs[k]
            Ls, Rs = synth.make_stream(400 + k, n, W, H, D)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                dL, dR = torch.from_numpy(Ls).cuda(), torch.from_numpy(Rs).cuda()
                dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
                m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=n)
                for _ in range(4):
                    m.compute_device(dL, dR, dD, s.cuda_stream)
                s.synchronize()
                one = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H)
                h = one.compute(Ls[0], Rs[0])
                one.close()
                got[k] = (dD.cpu().numpy(), h, Ls, Rs)
                m.close()
        except Exception as e:              # noqa: BLE001 -- reported by the main thread
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errs, errs
    for k, (W, H, D, w, n) in enumerate(cfgs):
        out, h, Ls, Rs = got[k]
        for i in (0, n - 1):
            assert_same(out[i], oracle.bm_compute(Ls[i], Rs[i], numDisparities=D, blockSize=w))
        assert_same(h, out[0])


# ---- bench.py's own line: every north_star size with parity, and the evidence kept at N > 1 ---------------------------
def _bench(args, env=None, timeout=900):
    import json, subprocess, sys
    e = dict(os.environ)
    e.update(env or {})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, env=e, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_line_carries_every_north_star_size_with_parity():
    out = _bench(["--steps", "3", "--warmup", "2", "--batch", "64"])
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1 and out["steps"] == 3 and out["parity_ok"] is True
    keys = [c["key"] for c in out["configs"]]
    assert keys == ["config1", "config2", "config3", "reference_default", "config5"]
    for c in out["configs"][:4]:
        assert c["parity_ok"] is True and c["steps"] >= 20 and c["pairs_per_s"] > 0 and c["cpu_pairs_per_s"] > 0
        assert 0 < c["frac_of_hbm"] < 1 and c["search_kernel"].startswith("fast")
    sg = out["configs"][4]                                   # the SWSemiGlobalMatcher counterpart: both modes, tolerance 0
    assert sg["parity_ok"] is True and sg["tolerance"] == 0 and 0 < sg["frac_of_hbm"] < 1
    for mode, sweeps in (("mode_hh_8_paths", 2), ("mode_sgbm_5_paths", 1)):
        assert sg[mode]["parity_ok"] is True and sg[mode]["pairs_per_s"] > 0 and sg[mode]["cpu_pairs_per_s_1_thread"] > 0
        assert sg[mode]["row_synchronous_sweeps_per_call"] == sweeps and sg[mode]["sweep_gave_up"] is False
    assert out["configs"][2]["morph_parity_ok"] is True
    assert out["sustained"]["seconds"] >= 1.5
    assert out["roofline"]["qsad_issue_floor"]["frac_of_floor"] < 1.0
    cb = out["cpu_baseline"]
    assert cb["mode"] in ("frames", "rows") and cb["cores"] == cb["threads"] and cb["value"] > 0
    ref = out["single_frame"]["reference_call"]
    assert ref["parity_ok"] is True and out["single_frame"]["same_as_batched"] is True


def test_bench_two_ranks_keep_cpu_baseline_and_roofline():
    # gloo: both ranks share this GPU; the N > 1 line must be as complete as the N = 1 line
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "2", "--batch", "16", "--no-configs"], env={"RTDM_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["parity_ok"] is True and out["parity_checked_frames"] == 6
    assert out["cpu_baseline"] is not None and out["cpu_baseline"]["value"] > 0
    assert out["roofline"]["achieved"] > 0


def test_two_gpus_over_rccl_when_the_box_has_them():
    """Multi-rank RCCL execution: runs only where torch sees >= 2 GPUs (the driver's 8-GPU node); a one-GPU box skips.  No
    scaling curve is produced or simulated here."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs: multi-rank RCCL is unmeasured on a one-GPU box")
    env = {"RTDM_DIST_BACKEND": "nccl"}
    out = _bench(["--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "64", "--no-configs"], env=env)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["parity_ok"] is True
    assert out["cpu_baseline"] is not None
    out = _bench(["--gpus", "2", "--rccl-stream", "64", "--chunk", "8", "--steps", "2", "--warmup", "1"], env=env)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["backend"] == "nccl"
    assert out["parity_ok"] is True and out["equals_direct_call"] is True
    ph = out["phase_ms_per_pass"]
    assert ph["chunks"] == 4 and ph["scatter_ms"] > 0 and ph["compute_ms"] > 0 and ph["gather_ms"] > 0


# ---- the reference's own parameter set (main.cpp:134-135 + cmdline-parser.cpp:22: d = 192, 13 x 13 at EVERY resolution) ---
# on the ring kernel now (eight lanes per pixel, four rows per group), as a device batch: autotuned strips, side-stream border
def test_reference_default_1280x720_d192_w13_batch32(pkg, oracle, synth):
    import torch
    n, W, H, D, w = 32, 1280, 720, 192, 13
    dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
    dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    pkg.synth_pairs_device(dL, dR, first_frame=300, numDisparities=D, stream=st)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=n)
    outs = []
    for _ in range(3):                      # model, measured strip count, its use
        dD.zero_()
        m.compute_device(dL, dR, dD, st)
        torch.cuda.synchronize()
        outs.append(dD.clone())
    assert m.search_variant == "fast_ring8_qsad"
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    for i in (0, 17, 31):
        assert_same(outs[2][i].cpu().numpy(), oracle.bm_compute(dL[i].cpu().numpy(), dR[i].cpu().numpy(), numDisparities=D, blockSize=w, nthreads=16))
    m.close()


@pytest.mark.parametrize("W,H,crop", [(320, 240, (49, 46, 233, 156)), (640, 480, (63, 61, 534, 378)), (1280, 720, (192, 177, 934, 404))])
def test_reference_call_shape_d192_w13_views_with_roi1(pkg, oracle, synth, W, H, crop):
    # estimator.cpp:33-36,54-56: roif-sized views at the crop origin of full-pitch planes, setROI1 before compute;
    # d = 192 does not fit the 233-wide crop (everything FILTERED there, as in the reference)
    cx, cy, cw, ch = crop
    L, R = synth.make_pair(synth.STREAM_SEED + 21 + W, W, H, 64)
    Lv, Rv = L[cy:cy + ch, cx:cx + cw], R[cy:cy + ch, cx:cx + cw]
    plane = np.full((H, W), 777, np.int16)
    Ov = plane[cy:cy + ch, cx:cx + cw]
    roi = (cw // 3, ch // 5, cw // 3, ch // 2)
    m = pkg.HIPMatcher(numOfDisparities=192, blockSize=13, width=cw, height=ch)
    m.setROI1(roi)
    m.compute(Lv, Rv, Ov)
    m.close()
    assert_same(Ov, oracle.bm_compute(np.ascontiguousarray(Lv), np.ascontiguousarray(Rv), numDisparities=192, blockSize=13, roi1=roi))
    guard = plane.copy(); guard[cy:cy + ch, cx:cx + cw] = 777
    assert (guard == 777).all()


# ---- k_search_border2 (rows in the lanes) against k_search_border (disparities in the lanes) and the oracle ------------
# The switch is read once per process, so each setting runs in a process of its own.  Shapes: rows fewer / more than one
# wave's 64 - (w - 1), ROI1 (vy0 > r, searched columns cut), both minDisparity signs, every window class (NSP 2, 3, 4), a
# frame too narrow for the anchored span (falls back), cap 63 (falls back: packed prefix sums), batches on the side stream.
_BORDER_CASES = r'''
import importlib, sys, zlib, numpy as np, torch
sys.path.insert(0, %r)
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
orc.build()
cases = [(64, 9, 0, 300, 40, None, 31, 1), (64, 9, 0, 300, 131, None, 31, 1), (32, 7, 0, 233, 156, (49, 30, 150, 90), 31, 1),
         (16, 5, 0, 90, 70, None, 31, 1), (128, 11, 0, 420, 100, None, 31, 1), (64, 15, 0, 260, 90, None, 15, 1),
         (64, 13, -5, 300, 80, None, 31, 1), (32, 9, 4, 200, 75, None, 31, 1), (64, 9, 0, 80, 60, None, 31, 1),
         (32, 9, 0, 250, 66, None, 63, 1), (192, 13, 0, 500, 120, None, 31, 1), (64, 9, 0, 640, 200, None, 31, 20),
         (256, 15, 0, 560, 90, None, 15, 1), (48, 7, 0, 301, 57, (0, 0, 301, 40), 31, 1)]
for (D, w, minD, W, H, roi, cap, n) in cases:
    Ls, Rs = pkg.synth.make_stream(123 + D + w, n, W, H, D)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, minDisparity=minD, preFilterCap=cap, width=W, height=H, max_batch=n)
    if roi: m.setROI1(roi)
    if n == 1:
        got = m.compute(Ls[0], Rs[0])[None]
    else:
        dL, dR = torch.from_numpy(Ls).cuda(), torch.from_numpy(Rs).cuda()
        dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
        m.compute_device(dL, dR, dD, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
        got = dD.cpu().numpy()
    m.close()
    for i in sorted({0, n - 1}):
        want = orc.bm_compute(Ls[i], Rs[i], numDisparities=D, blockSize=w, minDisparity=minD, preFilterCap=cap, roi1=roi)
        assert np.array_equal(got[i], want), ("oracle", D, w, minD, W, H, roi, cap, i, int((got[i] != want).sum()))
    print("CRC", D, w, minD, W, H, zlib.crc32(got.tobytes()))
print("ok")
'''


def test_border2_equals_border_and_the_oracle():
    import subprocess, sys
    outs = {}
    for flag in ("1", "0"):
        env = dict(os.environ, RTDM_BORDER2=flag)
        p = subprocess.run([sys.executable, "-c", _BORDER_CASES % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           timeout=900, env=env)
        assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stderr[-3000:]
        outs[flag] = [ln for ln in p.stdout.splitlines() if ln.startswith("CRC")]
    assert outs["1"] == outs["0"] and len(outs["1"]) == 14


def test_left_right_check_forms_agree_and_match_the_oracle():
    # the left-right check + speckle init runs as k_lrcheck_vec<.., NIT = 2> (vertical contacts inside two-row-pair blocks found in
    # the kernel, k_spk_merge_rec) for small launches and narrow frames, as k_lrcheck_pk (packed arithmetic, one row pair) for
    # large 1280-wide batches; RTDM_LR_PAIRS fixes the form, RTDM_LR_PACKED=0 takes the unpacked one-pair form: same bytes
    import subprocess, sys
    outs = {}
    for name, extra in (("auto", {}), ("pairs1", {"RTDM_LR_PAIRS": "1"}), ("pairs2", {"RTDM_LR_PAIRS": "2"}), ("pairs4", {"RTDM_LR_PAIRS": "4"}),
                        ("vec1", {"RTDM_LR_PAIRS": "1", "RTDM_LR_PACKED": "0"}),
                        # the frame fill as a launch of its own instead of inside the prefilter's grid, k_spk_merge_rec likewise
                        ("unfused", {"RTDM_FILL_IN_PREFILTER": "0", "RTDM_MERGE_REC_FUSED": "0"})):
        p = subprocess.run([sys.executable, "-c", _BORDER_CASES % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           timeout=900, env=dict(os.environ, **extra))
        assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (name, p.stderr[-3000:])
        outs[name] = [ln for ln in p.stdout.splitlines() if ln.startswith("CRC")]
    assert len(outs["auto"]) == 14 and all(v == outs["auto"] for v in outs.values()), [k for k, v in outs.items() if v != outs["auto"]]


# ---- StereoSGBM: any block size (sgbm-sw.cpp:15 passes the caller's through) ---------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("bs,paths", [(19, 8), (21, 5), (4, 8), (18, 5), (25, 8)])
def test_sgm_windows_above_17_and_even_sizes(pkg, oracle, synth, bs, paths):
    # windows > 17 run the generic block-sum kernel with the cost check on; an even size is the next odd one (as in the library)
    D = 32
    L, R = synth.make_pair(synth.STREAM_SEED + 900 + bs, 150, 70, D)
    m = pkg.HIPSemiGlobalMatcher(blockSize=bs, numOfDisparities=D, width=150, height=70, paths=paths)
    got = m.compute(L, R)
    m.close()
    want = oracle.sgm_compute(L, R, numDisparities=D, blockSize=bs, paths=paths)
    assert np.array_equal(got, want), (bs, paths, int((got != want).sum()))


@pytest.mark.gpu
def test_sgm_refuses_the_frame_whose_costs_would_wrap(pkg, oracle, synth):
    # flat 255 against flat 0: every pixel cost is 63, a 25 x 25 block cost 39375 > 32767 - P2 -- the library's short
    # arithmetic would wrap there, which neither the oracle nor the device restates: both refuse THIS FRAME, and the
    # handle keeps working
    D, W, H = 16, 96, 40
    Lo, Ro = np.full((H, W), 255, np.uint8), np.zeros((H, W), np.uint8)
    m = pkg.HIPSemiGlobalMatcher(blockSize=25, numOfDisparities=D, width=W, height=H, paths=8)
    with pytest.raises(Exception):
        m.compute(Lo, Ro)
    with pytest.raises(ValueError):
        oracle.sgm_compute(Lo, Ro, numDisparities=D, blockSize=25)
    L, R = synth.make_pair(synth.STREAM_SEED + 77, W, H, D)
    assert np.array_equal(m.compute(L, R), oracle.sgm_compute(L, R, numDisparities=D, blockSize=25))
    m.close()


# ---- StereoSGBM path passes on half-waves (k_sgm_path_h): every numDisparities, both modes ----------------------------------------
_SGM_HALF_CASES = r'''
import sys, zlib
sys.path.insert(0, %r)
import importlib
import numpy as np
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
orc.build()
synth = pkg.synth
for D in range(16, 257, 16):
    for paths, W, H in ((8, D + 61, 23), (5, D + 44, 18)):
        L, R = synth.make_pair(synth.STREAM_SEED + 4000 + D + paths, W, H, D)
        m = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, paths=paths, P1=600 if D %% 32 else 37, P2=2400 if D %% 48 else 30000)
        got = m.compute(L, R)
        sw = m.pass_stats()[0]
        m.close()
        want = orc.sgm_compute(L, R, numDisparities=D, paths=paths, P1=600 if D %% 32 else 37, P2=2400 if D %% 48 else 30000)
        assert np.array_equal(got, want), (D, paths, int((got != want).sum()))
        print("CRC", D, paths, zlib.crc32(got.tobytes()))
print("SWEEPS", sw)
print("ok")
'''


@pytest.mark.gpu
def test_sgm_half_wave_paths_every_d_and_equal_to_the_wave_form():
    # a lane of k_sgm_path_h holds 2 / 4 / 8 disparities (D <= 64 / 128 / 256): every multiple of 16 leaves a different number of
    # dead lanes; W1 = 61 (odd: the last wave of a vertical pass carries one line) and 44; P2 = 30000 (packed u16 sums:
    # minimum + P2 stays below 65536).  RTDM_SGM_HALF=0 runs the round-2 form (one wave per line, 32-bit): same bytes.
    # RTDM_SGM_SWEEP=0: the six directions that advance a row per step as passes of their own instead of two row-synchronous sweeps.
    import subprocess, sys
    outs = {}
    # RTDM_SGM_SWEEP_COLS = 1 / 2 / 4: strips of 8 / 16 / 32 columns (one, two, four columns per half-wave).
    # The round-2 forms stay reachable too: RTDM_SGM_FUSE_SELECT=0 (k_sgm_select reads S back), RTDM_SGM_WAVE_PATHS=0 (a workgroup per line).
    for flag, sweep, cols in (("2", "1", "1"), ("2", "1", "2"), ("2", "1", "4"), ("2", "0", "0"), ("1", "0", "0"), ("0", "0", "0"),
                              ("0", "0", "nofuse"), ("0", "0", "nowave"), ("2", "1", "nodual"), ("2", "1", "nopixbox")):
        env = dict(os.environ, RTDM_SGM_HALF=flag, RTDM_SGM_SWEEP=sweep, RTDM_SGM_SWEEP_COLS=cols if cols.isdigit() else "0")
        if cols == "nofuse": env["RTDM_SGM_FUSE_SELECT"] = "0"
        if cols == "nowave": env["RTDM_SGM_WAVE_PATHS"] = "0"
        if cols == "nopixbox": env["RTDM_SGM_PIXBOX"] = "0"       # pixel cost and block sum as two kernels with the u8 volume between them
        if cols == "nodual": env["RTDM_SGM_DUAL"] = "0"           # the two horizontal directions one after the other instead of side by side
        p = subprocess.run([sys.executable, "-c", _SGM_HALF_CASES % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           timeout=900, env=env)
        assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (flag, sweep, cols, p.stdout[-500:], p.stderr[-3000:])
        outs[flag + sweep + cols] = [ln for ln in p.stdout.splitlines() if ln.startswith("CRC")]
        assert ("SWEEPS 0" in p.stdout) == (sweep == "0"), (flag, sweep, p.stdout[-300:])
    assert len(outs["211"]) == 32 and all(v == outs["211"] for v in outs.values()), [k for k, v in outs.items() if v != outs["211"]]


@pytest.mark.gpu
@pytest.mark.parametrize("paths,W,H,D,n", [(8, 300, 90, 64, 1), (5, 300, 90, 64, 3), (8, 233, 61, 128, 2), (5, 130, 40, 96, 1), (8, 1280, 200, 128, 2)])
def test_sgm_row_synchronous_sweep_is_what_runs(pkg, oracle, synth, paths, W, H, D, n):
    # the down / up directions run as row-synchronous sweeps (k_sgm_sweep: three directions per pass, strips of 32 columns that hand
    # their edge lines to the neighbours through a tagged ring): one sweep per call in MODE_SGBM, two in MODE_HH, none gives up;
    # widths that leave a partial last strip, one strip only (W1 = 34 -> two), many strips (36 at 1280), batches
    import torch
    Ls, Rs = synth.make_stream(77 + W, n, W, H, D)
    m = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, max_batch=n, paths=paths)
    dL, dR = torch.from_numpy(Ls).cuda(), torch.from_numpy(Rs).cuda()
    dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    for rep in range(3):                                     # the ring's tags change with every launch
        dD.zero_()
        m.compute_device(dL, dR, dD, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = dD.cpu().numpy()
        for i in range(n):
            want = oracle.sgm_compute(Ls[i], Rs[i], numDisparities=D, paths=paths)
            assert np.array_equal(got[i], want), (rep, i, int((got[i] != want).sum()))
    sweeps, gave_up = m.pass_stats()
    m.close()
    assert sweeps == 3 * (2 if paths == 8 else 1) and not gave_up


@pytest.mark.gpu
def test_two_sgm_handles_on_two_threads_and_a_clean_exit():
    # the sweeps of one process share one stream per device (their workgroups wait for each other, so two sweeps must never be
    # half resident each); the process must also END cleanly -- with hipLaunchCooperativeKernel called from worker threads it
    # died in the runtime's exit handlers, which is why the sweep is an ordinary launch (k_sgm.hip: launch_sweep_c)
    import json, subprocess, sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sgm_two_threads.py"), "3", "3"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stdout[-500:], p.stderr[-2000:])
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert not line["errors"] and line["same_as_one_thread"], line
    assert line["results"]["0"]["pass_stats"] == [6, False] and line["results"]["1"]["pass_stats"] == [3, False], line


_SGM_GIVEUP = r'''
import sys, time
sys.path.insert(0, %r)
import importlib
import numpy as np
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
orc.build()
W, H, D = 200, 40, 32
L, R = pkg.synth.make_pair(pkg.synth.STREAM_SEED + 5, W, H, D)
m = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, paths=8)
t0 = time.time()
try:
    m.compute(L, R)
    print("NOERROR")
except Exception as e:
    print("ERROR", str(e)[:160].replace("\n", " "))
print("SECONDS", round(time.time() - t0, 2), "STATS", m.pass_stats())
got = m.compute(L, R)                       # from now on: one pass per direction
print("SECOND", bool(np.array_equal(got, orc.sgm_compute(L, R, numDisparities=D))), m.pass_stats())
m.close()
print("ok")
'''


@pytest.mark.gpu
def test_sgm_sweep_gives_up_instead_of_hanging():
    # the second line of defence of the row-synchronous sweep: a strip whose neighbour never publishes its edge (forced here:
    # RTDM_SGM_SWEEP_TEST_GIVEUP) stops waiting after its bound, the pass runs to its end, the call reports the failure ONCE and
    # the handle computes correct maps with one pass per direction from then on -- nothing hangs
    import subprocess, sys
    p = subprocess.run([sys.executable, "-c", _SGM_GIVEUP % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env=dict(os.environ, RTDM_SGM_SWEEP_TEST_GIVEUP="1"))
    out = p.stdout
    assert p.returncode == 0 and out.strip().endswith("ok"), (out[-600:], p.stderr[-2000:])
    assert "ERROR" in out and "gave up" in out and "NOERROR" not in out, out
    assert "SECOND True" in out and "True)" in out.split("SECOND True")[1], out        # pass_stats(): gave_up stays reported
    secs = float(out.split("SECONDS")[1].split()[0])
    assert secs < 60, secs


@pytest.mark.gpu
@pytest.mark.parametrize("paths", [8, 5])
def test_sgm_every_cost_saturated_is_no_winner(pkg, oracle, synth, paths):
    # a large P2 and eight paths drive the saturating sum (R5) to 32767 at EVERY disparity of some pixels: the library's search
    # for a cost below its initial SHRT_MAX then finds none (bestDisp stays -1: the invalid value, no vote).  Found by
    # tools/soak_sgm.py (seed 500155): the device took disparity 0 there, in every form since round 2.
    W, H, D = 88, 67, 48
    kw = dict(blockSize=11, uniquenessRatio=0, speckleWindowSize=20, speckleRange=2, disp12MaxDiff=1, P1=600, P2=20000, paths=paths)
    Ls, Rs = synth.make_stream(155, 1, W, H, D)
    L, R = (Ls[0] // 32 * 32).astype(np.uint8), (Rs[0] // 32 * 32).astype(np.uint8)
    want = oracle.sgm_compute(L, R, numDisparities=D, **kw)
    m = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, **kw)
    got = m.compute(L, R)
    m.close()
    assert np.array_equal(got, want), int((got != want).sum())

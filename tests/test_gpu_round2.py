"""GPU suite, round-2 additions (-m gpu): the BASELINE configurations that had not run at their stated size, the
timed path of bench.py at a batch that takes the autotuned + side-stream branch, config 4's scatter/gather on a real
RCCL process group, and the autotuner's cost for a caller that moves its ROI.  Everything goes through the C ABI and
is compared bit for bit with the CPU oracle (parity unpinned against OpenCV itself: oracle/rtdm_oracle.h)."""
import json
import os
import subprocess
import sys
import time

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


# ---- BASELINE config 5 at its stated size (sgbm-sw.cpp:12-37: blockSize from the ctor, P1 600, P2 2400) -------------
@pytest.mark.parametrize("paths", [8, 5])
def test_config5_sgm_1280x720_d128(pkg, oracle, synth, paths):
    W, H, D = 1280, 720, 128
    L, R = synth.make_pair(synth.STREAM_SEED + 40, W, H, D)
    sg = pkg.HIPSemiGlobalMatcher(blockSize=5, numOfDisparities=D, width=W, height=H, paths=paths)
    got = sg.compute(L, R)
    sg.close()
    want = oracle.sgm_compute(L, R, numDisparities=D, blockSize=5, paths=paths)
    assert_same(got, want)
    assert (got != -16).mean() > 0.3


# ---- the headline at a batch that takes bench.py's branch: autotuned strips + border kernel on the side stream --------
def test_headline_batch32_autotuned_side_stream(pkg, oracle, synth):
    import torch
    n, W, H, D = 32, 1280, 720, 64
    dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
    dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    pkg.synth_pairs_device(dL, dR, first_frame=100, numDisparities=D, stream=st)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=9, width=W, height=H, max_batch=n)
    outs = []
    for _ in range(3):                      # 1st call: model; 2nd: measures the strip count; 3rd: uses the measured one
        dD.zero_()
        m.compute_device(dL, dR, dD, st)
        torch.cuda.synchronize()
        outs.append(dD.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    for i in (0, 15, 31):
        L, R = dL[i].cpu().numpy(), dR[i].cpu().numpy()
        Ls, Rs = synth.make_pair(synth.STREAM_SEED + 100 + i, W, H, D)
        assert np.array_equal(L, Ls) and np.array_equal(R, Rs)          # device stream == CPU stream
        assert_same(outs[2][i].cpu().numpy(), oracle.bm_compute(L, R, numDisparities=D, blockSize=9, nthreads=16))
    m.close()


# ---- config 4 on a real RCCL process group (world 1 on this box) and bench.py's own launcher -------------------------
def _bench(args, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, env=e, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_rccl_stream_world1_nccl_equals_direct_call():
    # a process group of size 1 with backend nccl: the scatter / gather of config 4 run as real RCCL calls
    out = _bench(["--rccl-stream", "16", "--chunk", "6", "--steps", "1", "--warmup", "1"], env={"RTDM_DIST_BACKEND": "nccl"})
    assert out["n_gpus"] == 1 and out["frames"] == 16 and out["backend"] == "nccl"
    assert out["parity_ok"] is True and out["equals_direct_call"] is True


def test_bench_gpus2_launches_two_ranks_itself():
    # `python bench.py --gpus 2` with no torchrun environment: bench.py starts the ranks (gloo: both share this GPU)
    env = {"RTDM_DIST_BACKEND": "gloo"}
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "2", "--batch", "16", "--no-cpu-baseline", "--no-configs"], env=env)
    assert out["n_gpus"] == 2 and out["parity_ok"] is True and out["parity_checked_frames"] == 6
    # config 4 through the same launcher, two ranks
    out = _bench(["--gpus", "2", "--rccl-stream", "10", "--chunk", "2", "--steps", "1", "--warmup", "1"], env=env)
    assert out["n_gpus"] == 2 and out["parity_ok"] is True and out["equals_direct_call"] is True


# (the autotuner's cost for a caller that moves its ROI is asserted on the tuner's own counters now:
#  tests/test_gpu_round3.py::test_roi_moving_batched_caller_never_triggers_the_tuner)


# ---- the opt-in two-lane mode (RTDM_LANES=2): pieces on two workspace slices, row kernels on a back stream ------------
def test_two_lane_pipeline_matches_the_oracle():
    code = r'''
import importlib, sys, numpy as np, torch
sys.path.insert(0, %r)
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
orc.build()
n, W, H, D, w = 22, 400, 150, 64, 9
Ls, Rs = pkg.synth.make_stream(31, n, W, H, D)
dL, dR = torch.from_numpy(Ls).cuda(), torch.from_numpy(Rs).cuda()
dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=n)
st = torch.cuda.current_stream().cuda_stream
for rep in range(3):                       # the autotuned strip count comes in on the later calls
    dD.fill_(777)
    m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize()
got = dD.cpu().numpy()
m.close()
for i in range(n):
    want = orc.bm_compute(Ls[i], Rs[i], numDisparities=D, blockSize=w)
    assert np.array_equal(got[i], want), (i, int((got[i] != want).sum()))
print("ok")
''' % ROOT
    env = dict(os.environ, RTDM_LANES="2", RTDM_PIECES="5")
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stderr[-2000:]


# ---- rtdm_bm_compute_batch on page-locked host memory: three streams, two halves of the staging planes ------------------
@pytest.mark.parametrize("W,H,n,mb", [(333, 120, 11, 6), (640, 96, 9, 4), (400, 150, 5, 2), (200, 64, 3, 1)])
def test_host_batch_pipeline_on_page_locked_memory(pkg, oracle, synth, W, H, n, mb):
    import torch
    D, w = 32, 7
    Ls, Rs = synth.make_stream(77, n, W, H, D)
    pl, pr = torch.from_numpy(Ls).pin_memory(), torch.from_numpy(Rs).pin_memory()
    po = torch.full((n, H, W), 12345, dtype=torch.int16).pin_memory()
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=mb)
    got = m.compute_batch(pl.numpy(), pr.numpy(), po.numpy())
    again = m.compute_batch(Ls, Rs)                       # the same frames from pageable memory: the unpipelined path
    m.close()
    assert np.array_equal(got, again)
    for i in range(n):
        assert_same(got[i], oracle.bm_compute(Ls[i], Rs[i], numDisparities=D, blockSize=w))


# ---- rtdm_bm_compute on page-locked planes: DMA straight from / to the caller's (pitched) views -----------------------
def test_single_frame_from_page_locked_pitched_views(pkg, oracle, synth):
    import torch
    W, H, D, w = 233, 156, 32, 7                                # the reference's crop of a 320-wide frame (estimator.cpp:33,36)
    Lf, Rf = synth.make_pair(synth.STREAM_SEED + 5, 320, 240, D)
    pl, pr = torch.from_numpy(Lf).pin_memory(), torch.from_numpy(Rf).pin_memory()
    po = torch.full((240, 320), 777, dtype=torch.int16).pin_memory()
    L, R, O = pl.numpy()[40:40 + H, 50:50 + W], pr.numpy()[40:40 + H, 50:50 + W], po.numpy()[40:40 + H, 50:50 + W]
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H)
    m.compute(L, R, O)
    m.close()
    assert_same(O, oracle.bm_compute(np.ascontiguousarray(L), np.ascontiguousarray(R), numDisparities=D, blockSize=w))
    guard = po.numpy().copy(); guard[40:40 + H, 50:50 + W] = 777
    assert (guard == 777).all()                                 # nothing outside the view was written

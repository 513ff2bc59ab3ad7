#!/usr/bin/env python3
"""Times one iteration of Estimator::run without capture/decode/drawing (estimator.cpp:29-77) through
rtdm_estimate_frame on the reference's 1280x720 calibration, host frames in, per-object depths out, and the
object-detection part alone; checks both against the oracle chain.  Run on the GPU box."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
import rectify_util as ru
c, maps = ru.maps(orc, "1280x720")
W, H = c["W"], c["H"]; x, y, rw, rh = c["roi"]
D, w = 64, 9
left, right = ru.red_scene(pkg.synth, 1, W, H, D)
rect = pkg.HIPRectifier(*maps, roi=c["roi"]); m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=rw, height=rh)
det = pkg.HIPObjectDetector(rw, rh)
def timed(fn, reps=30):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps
col = orc.rectify_rgb(left, maps[0], maps[1], c["roi"])
t_frame = timed(lambda: pkg.estimate_frame(m, rect, det, left, right, c["Q"]))
t_det = timed(lambda: det.detect(col))
boxes, mean, cnt, disp = pkg.estimate_frame(m, rect, det, left, right, c["Q"], want_disp=True)
t0 = time.perf_counter()
gl = orc.rectify_gray(left, maps[0], maps[1], c["roi"]); gr = orc.rectify_gray(right, maps[2], maps[3], c["roi"])
fout = orc.morph_open_close(orc.hsv_inrange(col)); wb = orc.external_boxes(fout, 100, True)
wd = orc.bm_compute(gl, gr, numDisparities=D, blockSize=w, roi1=orc.union_box(wb), nthreads=16)
wm, wc = orc.depth_stats(wd, c["Q"], fout, wb[:64])
t_cpu = time.perf_counter() - t0
print(json.dumps({"frame": [W, H], "crop_roif": list(c["roi"]), "objects": len(boxes),
                  "estimate_frame_ms_host_to_host": round(t_frame * 1e3, 3), "frames_per_s": round(1 / t_frame, 1),
                  "detect_only_ms_host_to_host": round(t_det * 1e3, 3), "cpu_oracle_chain_ms_16_threads_matcher": round(t_cpu * 1e3, 1),
                  "bit_exact_boxes_disp_counts": bool(boxes == wb[:64] and np.array_equal(disp, wd) and np.array_equal(cnt, wc)),
                  "mean_depth_max_rel_err": float(np.max(np.abs(mean - wm) / np.maximum(np.abs(wm), 1e-30))) if len(wm) else 0.0}, indent=1))

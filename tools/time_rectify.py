#!/usr/bin/env python3
"""Times the rectification step in front of the matcher on device-resident RGB
frames with the reference's 1280x720 calibration (tests/golden/calib.json): gray+remap+crop alone, and chained
into the matcher (raw frames -> disparity).  Run on the GPU box."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
import rectify_util as ru
c, maps = ru.maps(orc, "1280x720")
W, H = c["W"], c["H"]; x, y, rw, rh = c["roi"]
n, D, w = 64, 64, 9
st = torch.cuda.current_stream().cuda_stream
left, right = ru.rgb_pair(pkg.synth, 3, W, H)
dL = torch.from_numpy(left).cuda()[None].repeat(n, 1, 1, 1).contiguous(); dR = torch.from_numpy(right).cuda()[None].repeat(n, 1, 1, 1).contiguous()
gl = torch.empty((n, rh, rw), dtype=torch.uint8, device="cuda"); gr = torch.empty_like(gl)
dD = torch.empty((n, rh, rw), dtype=torch.int16, device="cuda")
r = pkg.HIPRectifier(*maps, roi=c["roi"], max_batch=n)
m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=rw, height=rh, max_batch=n)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
t_rect = timed(lambda: r.gray_device(dL, dR, gl, gr, st))
t_bm = timed(lambda: m.compute_device(gl, gr, dD, st))
t_chain = timed(lambda: r.compute_device(m, dL, dR, dD, st))
t0 = time.perf_counter(); wl = orc.rectify_gray(left, maps[0], maps[1], c["roi"]); wr = orc.rectify_gray(right, maps[2], maps[3], c["roi"]); tc = time.perf_counter() - t0
want = orc.bm_compute(wl, wr, numDisparities=D, blockSize=w, nthreads=8)
# algorithmic bytes per pair: read 2 RGB frames' crop footprint (~3 B/px) + 6 B/px of maps, write 1 B/px, both cameras
alg = 2 * rw * rh * (3 + 6 + 1)
t0 = time.perf_counter(); hd = r.compute(m, left, right); th = time.perf_counter() - t0
t0 = time.perf_counter(); hd = r.compute(m, left, right); th = time.perf_counter() - t0
print(json.dumps({"frame": [W, H], "crop_roif": list(c["roi"]), "batch": n,
                  "rectify_gray_us_per_pair": round(t_rect / n * 1e6, 2), "rectify_algorithmic_GBps": round(alg * n / t_rect / 1e9, 1),
                  "rectify_hbm_frac_of_8TBps": round(alg * n / t_rect / 8e12, 4),
                  "matcher_us_per_pair_on_crop": round(t_bm / n * 1e6, 2), "raw_frames_to_disparity_us_per_pair": round(t_chain / n * 1e6, 2),
                  "raw_frames_to_disparity_pairs_per_s": round(n / t_chain), "host_to_host_ms_single_pair": round(th * 1e3, 3),
                  "cpu_oracle_rectify_ms_per_pair_1thread": round(tc * 1e3, 1),
                  "bit_exact_rectify": bool(np.array_equal(gl[0].cpu().numpy(), wl) and np.array_equal(gr[5].cpu().numpy(), wr)),
                  "bit_exact_chain": bool(np.array_equal(dD[7].cpu().numpy(), want) and np.array_equal(hd, want))}, indent=1))

#!/usr/bin/env python3
"""ms per 1280x720 D=128 pair of the StereoSGBM path (both modes), checked against the oracle on one frame.
    python tools/time_sgm.py [n=4]"""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H, D = 1280, 720, 128
st = torch.cuda.current_stream().cuda_stream
dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
pkg.synth_pairs_device(dL, dR, 0, D)
L, R = dL[n - 1].cpu().numpy(), dR[n - 1].cpu().numpy()
for paths in (8, 5):
    sg = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, max_batch=n, paths=paths)
    for _ in range(2): sg.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): sg.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    ok = bool(np.array_equal(dD[n - 1].cpu().numpy(), orc.sgm_compute(L, R, numDisparities=D, paths=paths))) if os.environ.get("SGM_CHECK", "1") == "1" else None
    print("paths %d: %.4f ms per pair (%d pairs per call) exact=%s" % (paths, dt / n * 1e3, n, ok), flush=True)
    sg.close()

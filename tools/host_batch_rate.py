#!/usr/bin/env python3
"""PCIe-inclusive rate: rtdm_bm_compute_batch on host-resident frames (pageable numpy), 1280x720 d=64 9x9."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
W, H, D, n = 1280, 720, 64, 128
dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
pkg.synth_pairs_device(dL, dR, 0, D)
L, R = dL.cpu().numpy(), dR.cpu().numpy()
m = pkg.HIPMatcher(numOfDisparities=D, blockSize=9, width=W, height=H, max_batch=64)
m.compute_batch(L[:8], R[:8])
t0 = time.perf_counter(); out = m.compute_batch(L, R); dt = time.perf_counter() - t0
print("host-to-host batch (pageable, PCIe inclusive): %.0f pairs/s, %.3f ms/pair, %.2f GB/s over PCIe" % (n / dt, dt / n * 1e3, 4 * W * H * n / dt / 1e9))

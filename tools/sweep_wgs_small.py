#!/usr/bin/env python3
"""search-stage time for the smaller BASELINE shapes against RTDM_FAST_WGS (set in the environment by the caller)."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("rt-depth-map_amd")
st = torch.cuda.current_stream().cuda_stream
for (W, H, D, w, n) in ((320, 240, 32, 7, 256), (640, 480, 64, 9, 128), (1280, 720, 128, 11, 64), (1280, 720, 64, 9, 128)):
    dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL); dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    pkg.synth_pairs_device(dL, dR, 0, D)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=n)
    for _ in range(3): m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize(); m.set_profiling(True); m.reset_stage_times()
    t0 = time.perf_counter()
    for _ in range(10): m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    s = m.stage_times()["search"]
    print(os.environ.get("RTDM_FAST_WGS", "default"), W, H, D, w, n, "pairs/s", round(n / dt), "search_ms", round(s["total_ms"] / max(s["launches"], 1), 3))
    m.close()

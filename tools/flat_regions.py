#!/usr/bin/env python3
"""Effect of the wave-level texture early-out of k_search_fast: the headline workload with the upper half of every
frame replaced by a flat (untextured) area, as walls or sky are in real scenes."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("rt-depth-map_amd")
n, W, H, D, w = 128, 1280, 720, 64, 9
st = torch.cuda.current_stream().cuda_stream
dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL); dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
pkg.synth_pairs_device(dL, dR, 0, D)
m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=n)
def run(tag):
    for _ in range(3): m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize(); m.set_profiling(True); m.reset_stage_times(); t0 = time.perf_counter()
    for _ in range(10): m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    s = m.stage_times()["search"]; m.set_profiling(False)
    print(tag, "pairs/s", round(n / dt), "search_ms", round(s["total_ms"] / s["launches"], 3), "valid", round(float((dD != -16).float().mean()), 3))
run("textured everywhere      ")
dL[:, :H // 2, :] = 128; dR[:, :H // 2, :] = 128
run("upper half of frames flat")

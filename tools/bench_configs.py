#!/usr/bin/env python3
"""Throughput of every BASELINE configuration on one MI355X next to the CPU oracle (run on the GPU box).

    python tools/bench_configs.py > gpurun_out/configs.json
"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc

CONFIGS = [
    ("config1 320x240 d=32 7x7", 320, 240, 32, 7, 256, None),
    ("config1 233x156 ROI crop of 320x240 d=32 7x7", 233, 156, 32, 7, 256, None),
    ("config2 640x480 d=64 9x9", 640, 480, 64, 9, 128, None),
    ("headline 1280x720 d=64 9x9", 1280, 720, 64, 9, 256, None),
    ("headline + ROI1 = 400x300 box (estimator.cpp:54)", 1280, 720, 64, 9, 64, (440, 210, 400, 300)),
    ("config3 1280x720 d=128 11x11", 1280, 720, 128, 11, 32, None),
    ("reference default 1280x720 d=192 13x13", 1280, 720, 192, 13, 32, None),
    # the reference's own parameter set is d=192 13x13 at EVERY frame size (main.cpp:134-135, cmdline-parser.cpp:22; nd is NOT
    # scaled with the width: cmdline-parser.h:85-89 divides by the parser's own width).  At 320 wide d=192 leaves 129 columns.
    ("reference default at 640x480: d=192 13x13", 640, 480, 192, 13, 128, None),
    ("reference default at 320x240: d=192 13x13", 320, 240, 192, 13, 256, None),
    ("(96,13) ring form at 640x480 -- not a reference setting", 640, 480, 96, 13, 128, None),
]
out = []
cores = min(os.cpu_count() or 1, 64)
for name, W, H, D, w, B, roi in CONFIGS:
    dL = torch.empty((B, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
    dD = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
    pkg.synth_pairs_device(dL, dR, 0, D)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=B)
    if roi: m.setROI1(roi)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 10
    for _ in range(K): m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    m.set_profiling(True); m.reset_stage_times(); m.compute_device(dL, dR, dD, st); torch.cuda.synchronize()
    stages = {k: round(v["total_ms"] / max(1, v["launches"]), 4) for k, v in m.stage_times().items()}
    # single-frame host-to-host latency through rtdm_bm_compute (PCIe inclusive)
    L, R = dL[0].cpu().numpy(), dR[0].cpu().numpy()
    m1 = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=1)
    if roi: m1.setROI1(roi)
    for _ in range(3): got = m1.compute(L, R)
    t0 = time.perf_counter()
    for _ in range(20): got = m1.compute(L, R)
    lat = (time.perf_counter() - t0) / 20
    # oracle: parity on this very frame + timing
    kw = dict(numDisparities=D, blockSize=w, roi1=roi)
    want = orc.bm_compute(L, R, nthreads=cores, **kw)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        orc.bm_compute(L, R, nthreads=cores, **kw); n += 1
    cpu = n / (time.perf_counter() - t0)
    out.append({"config": name, "batch": B, "gpu_pairs_per_s": round(B * K / dt, 1), "ms_per_pair": round(dt / (B * K) * 1e3, 5),
                "variant": m.search_variant, "stage_ms_per_batch": stages, "host_to_host_ms_single_frame": round(lat * 1e3, 3),
                "cpu_oracle_pairs_per_s": round(cpu, 2), "cpu_threads": cores, "bit_exact_vs_oracle": bool(np.array_equal(got, want)),
                "algorithmic_GBps": round(4 * W * H * B * K / dt / 1e9, 2)})
    m.close(); m1.close()
print(json.dumps(out, indent=1))

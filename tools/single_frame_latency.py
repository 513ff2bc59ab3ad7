#!/usr/bin/env python3
"""Single-frame host-to-host latency through rtdm_bm_compute (what the reference's per-frame loop would see), with the
device-side stage times, for both search kernels (run on the GPU box)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
lib = pkg.binding.lib()
synth = pkg.synth
for (W, H, D, w, roi) in ((233, 156, 32, 7, None), (320, 240, 32, 7, None), (534, 378, 64, 9, None), (640, 480, 64, 9, None),
                          (934, 404, 64, 9, None), (1280, 720, 64, 9, None), (1280, 720, 64, 9, (440, 210, 400, 300)),
                          (934, 404, 192, 13, None)):
    L, R = synth.make_pair(1, W, H, D)
    for mode in (0, 1):
        lib.rtdm_debug_search_kernel(mode)
        m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H)
        if roi: m.setROI1(roi)
        for _ in range(5): m.compute(L, R)
        m.set_profiling(True); m.reset_stage_times()
        t0 = time.perf_counter()
        for _ in range(50): m.compute(L, R)
        dt = (time.perf_counter() - t0) / 50
        st = {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in m.stage_times().items()}
        print(W, H, "d=%d w=%d roi=%s" % (D, w, roi), "%.3f ms host to host" % (dt * 1e3), st, m.search_variant, flush=True)
        m.close()
lib.rtdm_debug_search_kernel(-1)

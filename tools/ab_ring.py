#!/usr/bin/env python3
"""A/B of the hand-written search kernels (run on the GPU box): search-stage time per launch of k_search_fast (mode 0) and
k_search_ring with two, four and eight lanes per pixel (modes 2, 4, 8) for every configuration the ring kernel is instantiated for."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("rt-depth-map_amd")
lib = pkg.binding.lib()
st = torch.cuda.current_stream().cuda_stream
out = {}
for (W, H, B) in ((1280, 720, 64), (640, 480, 128), (320, 240, 256)):
    for (D, w) in ((64, 9), (64, 7), (64, 5), (32, 7), (32, 9), (32, 11), (32, 13), (48, 7), (48, 9), (16, 5), (16, 7), (16, 9),
                   (128, 7), (128, 9), (128, 11), (64, 11), (64, 13), (128, 13), (96, 7), (96, 9), (96, 11), (96, 13), (48, 11), (48, 13),
                   (16, 11), (16, 13), (32, 5), (32, 15), (48, 5), (64, 15), (128, 15)):
        dL = torch.empty((B, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
        dD = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
        pkg.synth_pairs_device(dL, dR, first_frame=0, numDisparities=D, stream=st)
        res = {}
        ref = None
        for mode in (0, 2, 4, 8):
            lib.rtdm_debug_search_kernel(mode)
            m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=B)
            for _ in range(3): m.compute_device(dL, dR, dD, st)
            torch.cuda.synchronize()
            m.set_profiling(True); m.reset_stage_times()
            for _ in range(5): m.compute_device(dL, dR, dD, st)
            torch.cuda.synchronize()
            t = m.stage_times()
            if m.search_variant in res: m.close(); continue     # no four-lane form: the same kernel again
            res[m.search_variant] = round(t["search"]["total_ms"] / t["search"]["launches"], 4)
            if ref is None: ref = dD.clone()
            else: res["same_bytes"] = res.get("same_bytes", True) and bool(torch.equal(ref, dD))
            m.close()
        lib.rtdm_debug_search_kernel(-1)
        best = min(v for k, v in res.items() if k.startswith("fast_ring"))
        res["ring_speedup"] = round(res.get("fast_qsad", 0) / best, 3)
        out["%dx%d_b%d_d%d_w%d" % (W, H, B, D, w)] = res
        print("%dx%d b%d d=%d w=%d %s" % (W, H, B, D, w, res), flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "ab_ring.json"), "w"), indent=1)

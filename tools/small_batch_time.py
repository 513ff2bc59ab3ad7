#!/usr/bin/env python3
"""Device-resident call time of the headline configuration for small batches (run on the GPU box; environment switches apply):
    python3 tools/small_batch_time.py [batches=1,2,4,8,16,32] [W H D w]"""
import importlib, json, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
batches = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8,16,32").split(",")]
W, H, D, w = [int(v) for v in sys.argv[2:6]] if len(sys.argv) > 5 else (1280, 720, 64, 9)
st = torch.cuda.current_stream().cuda_stream
out = {}
for n in batches:
    dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
    dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    pkg.synth_pairs_device(dL, dR, 0, D)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=n)
    for _ in range(5): m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 200 if n <= 4 else 60
    for _ in range(reps): m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        m.compute_device(dL, dR, dD, st); torch.cuda.synchronize()
    dts = (time.perf_counter() - t0) / reps
    out[n] = {"us_per_call": round(dt * 1e6, 1), "us_per_pair": round(dt * 1e6 / n, 2), "us_per_call_synchronised": round(dts * 1e6, 1),
              "crc": zlib.crc32(dD.cpu().numpy().tobytes())}
    m.close()
    print(n, out[n], flush=True)
print(json.dumps(out))

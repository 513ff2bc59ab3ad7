#!/usr/bin/env python3
"""Two StereoSGBM handles driven from two host threads on two streams, each asking for more than half of the device with its
row-synchronous sweeps (run on the GPU box under a timeout): do cooperative launches of one process get in each other's way?
    python3 tools/sgm_two_threads.py [pairs per call=16] [calls=4]"""
import importlib, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H, D = 1280, 720, 128
res, errs = {}, []

def run(k):
    try:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
            dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
            pkg.synth_pairs_device(dL, dR, 100 * k, D, stream=s.cuda_stream)
            torch.cuda.synchronize()
            sg = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, max_batch=n, paths=8 if k == 0 else 5)
            t0 = time.perf_counter()
            for _ in range(calls): sg.compute_device(dL, dR, dD, s.cuda_stream)
            s.synchronize()
            dt = time.perf_counter() - t0
            res[k] = {"ms_per_pair": round(dt / (n * calls) * 1e3, 3), "pass_stats": sg.pass_stats(), "crc": int(np.uint32(dD[0].cpu().numpy().astype(np.int64).sum() & 0xffffffff))}
            sg.close()
    except Exception as e:      # noqa: BLE001
        errs.append((k, repr(e)))

ts = [threading.Thread(target=run, args=(k,)) for k in range(2)]
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
both = round(time.perf_counter() - t0, 3)
threaded = {k: dict(v) for k, v in res.items()}
for k in range(2): run(k)                      # the same two jobs one after the other on this thread: same bytes?
same = all(k in threaded and k in res and threaded[k]["crc"] == res[k]["crc"] for k in range(2))
print(json.dumps({"both_threads_s": both, "results": threaded, "same_as_one_thread": same, "errors": errs}))

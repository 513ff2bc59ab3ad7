#!/usr/bin/env python3
"""StereoSGBM ms per pair against pairs per call (run on the GPU box): 1280x720 D=128 blockSize 5, both modes.
    python3 tools/sgm_batch_sweep.py [batches=1,2,4,8,16] [paths=8,5]
The cost volumes of a pair are 0.53 GB (pix u8 + C u16 + S u16), so a batch of 16 holds 8.5 GB."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
batches = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8,16").split(",")]
modes = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "8,5").split(",")]
W, H, D = 1280, 720, 128
st = torch.cuda.current_stream().cuda_stream
out = {}
ref = {}
for paths in modes:
    for n in batches:
        dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
        dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
        pkg.synth_pairs_device(dL, dR, 0, D)
        sg = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, max_batch=n, paths=paths)
        for _ in range(2): sg.compute_device(dL, dR, dD, st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = max(3, 16 // n)
        for _ in range(reps): sg.compute_device(dL, dR, dD, st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        got = dD[0].cpu().numpy()
        if paths not in ref: ref[paths] = got
        out["paths%d_n%d" % (paths, n)] = {"ms_per_pair": round(dt / n * 1e3, 4), "same_first_frame": bool(np.array_equal(got, ref[paths]))}
        sg.close(); del dL, dR, dD
        print("paths", paths, "n", n, out["paths%d_n%d" % (paths, n)], flush=True)
print(json.dumps(out))

#!/usr/bin/env python3
"""PCIe-inclusive rate: rtdm_bm_compute_batch on host-resident frames, 1280x720 d=64 9x9 -- pageable numpy arrays and
page-locked ones (torch pin_memory), checked against the device-resident result (run on the GPU box)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
W, H, D, n = 1280, 720, 64, 256
dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
pkg.synth_pairs_device(dL, dR, 0, D)
m = pkg.HIPMatcher(numOfDisparities=D, blockSize=9, width=W, height=H, max_batch=64)
dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for i in range(0, n, 64): m.compute_device(dL[i:i + 64], dR[i:i + 64], dD[i:i + 64], st)
torch.cuda.synchronize()
want = dD.cpu().numpy()
for kind in ("pageable", "page-locked"):
    if kind == "pageable":
        L, R = dL.cpu().numpy(), dR.cpu().numpy()
    else:
        pl, pr = dL.cpu().pin_memory(), dR.cpu().pin_memory()
        L, R = pl.numpy(), pr.numpy()
    po = torch.empty((n, H, W), dtype=torch.int16)
    if kind != "pageable": po = po.pin_memory()
    out = po.numpy()
    m.compute_batch(L[:64], R[:64], out[:64])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); m.compute_batch(L, R, out); best = min(best, time.perf_counter() - t0)
    print("host-to-host batch (%s in and out, PCIe inclusive): %.0f pairs/s, %.3f ms/pair, %.2f GB/s over PCIe, same as device path: %s"
          % (kind, n / best, best / n * 1e3, 4 * W * H * n / best / 1e9, bool(np.array_equal(out, want))), flush=True)

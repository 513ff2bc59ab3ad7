#!/usr/bin/env python3
"""Where a single 1280x720 frame's host-to-host time goes (run on the GPU box): the device-resident call alone, pinned and
pageable copies of the frame's bytes alone, and rtdm_bm_compute end to end."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
W, H, D, w = 1280, 720, 64, 9
L, R = pkg.synth.make_pair(1, W, H, D)
m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H)
def t(f, n=200):
    for _ in range(10): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("rtdm_bm_compute host to host      %.3f ms (fresh output array per call)" % t(lambda: m.compute(L, R)))
out = np.empty((H, W), np.int16)
print("rtdm_bm_compute host to host      %.3f ms (caller keeps its output array)" % t(lambda: m.compute(L, R, out)))
dL, dR = torch.from_numpy(L).cuda()[None], torch.from_numpy(R).cuda()[None]
dD = torch.empty((1, H, W), dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def dev():
    m.compute_device(dL, dR, dD, st); torch.cuda.synchronize()
print("device-resident call + sync       %.3f ms" % t(dev))
pl, pd = torch.empty((2, H, W), dtype=torch.uint8).pin_memory(), torch.empty((H, W), dtype=torch.int16).pin_memory()
gl = torch.empty((2, H, W), dtype=torch.uint8, device="cuda")
def copies():
    gl.copy_(pl, non_blocking=True); pd.copy_(dD[0], non_blocking=True); torch.cuda.synchronize()
print("pinned H2D 1.84 MB + D2H 1.84 MB  %.3f ms" % t(copies))
a = np.empty((2, H, W), np.uint8); b = np.empty((H, W), np.int16); pn, pdn = pl.numpy(), pd.numpy()
Ls = np.stack([L, R])
def hostcopy():
    pn[...] = Ls; b[...] = pdn
print("host memcpy in 1.84 MB + out 1.84 MB %.3f ms" % t(hostcopy))
m.close()

#!/usr/bin/env python3
"""Single 1280x720 frame through rtdm_bm_compute, 30 times (to be run under rocprofv3 --kernel-trace --stats)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("rt-depth-map_amd")
W, H, D, w = 1280, 720, 64, 9
L, R = pkg.synth.make_pair(1, W, H, D)
m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H)
out = np.empty((H, W), np.int16)
for _ in range(30): m.compute(L, R, out)
m.close()

#!/usr/bin/env python3
"""One-off extended fuzz: N random block-matching configurations (same generator as tests/test_gpu_fuzz.py, other seeds)
through the C ABI against the oracle.  Usage: python tools/soak_fuzz.py [first_seed] [count]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
assert torch.cuda.is_available()
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
import test_gpu_fuzz as tf
orc.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
variants = {}
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    W, H, kw, roi1, roi2 = tf._case(rng)
    L, R = pkg.synth.make_pair(seed, W, H, kw["numDisparities"])
    if rng.random() < 0.2:                       # plateaus: many exact ties
        L = (L // 32 * 32).astype(np.uint8); R = (R // 32 * 32).astype(np.uint8)
    want = orc.bm_compute(L, R, roi1=roi1, roi2=roi2, nthreads=4, **kw)
    D, w = kw["numDisparities"], kw["blockSize"]
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H,
                       **{k: v for k, v in kw.items() if k not in ("numDisparities", "blockSize")})
    if roi1: m.setROI1(roi1)
    if roi2: m.setROI2(roi2)
    got = m.compute(L, R)
    variants[m.search_variant] = variants.get(m.search_variant, 0) + 1
    m.close()
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH seed", seed, W, H, kw, roi1, roi2, int((got != want).sum()))
print("checked", count, "mismatches", bad, variants)
sys.exit(1 if bad else 0)

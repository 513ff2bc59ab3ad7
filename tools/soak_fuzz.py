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
    if (seed - first) % 500 == 0: print("case", seed - first, "mismatches so far", bad, flush=True)   # (a silent run looks hung)
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
print("checked", count, "mismatches", bad, variants, flush=True)

# ---- second part: the ring kernel's own table on taller frames and device batches (strip boundaries, the 16-bit cap of
# ---- a strip, the side-stream border kernel, the autotuned strip count on the second and third call)
RING = [(64, 9), (64, 7), (64, 5), (32, 7), (32, 9), (32, 11), (32, 13), (48, 7), (48, 9), (16, 5), (16, 7), (16, 9),
        (128, 7), (128, 9), (128, 11), (64, 9), (64, 7), (128, 11), (96, 9), (96, 13), (48, 13), (64, 13), (128, 13), (96, 11),
        (64, 15), (128, 15), (32, 15), (48, 5), (32, 5), (16, 13), (16, 11), (48, 11), (64, 11), (96, 7),
        (192, 13), (192, 9), (192, 11), (192, 15), (256, 15), (256, 9), (256, 11), (256, 13), (192, 13), (64, 9)]
st = torch.cuda.current_stream().cuda_stream
for seed in range(first, first + max(1, count // 10)):
    if (seed - first) % 100 == 0: print("ring case", seed - first, "mismatches so far", bad, flush=True)
    rng = np.random.default_rng(seed + 777)
    D, w = RING[int(rng.integers(0, len(RING)))]
    W, H = int(rng.integers(D + w + 40, D + w + 700)), int(rng.integers(w + 40, 420))
    n = int(rng.choice([1, 2, 5, 16, 20]))
    cap = int(rng.choice([31, 31, 63 if 126 * w * w <= 32766 else 31, 7]))
    kw = dict(numDisparities=D, blockSize=w, preFilterCap=cap, minDisparity=int(rng.choice([0, 0, -6, 4])),
              uniquenessRatio=int(rng.choice([10, 0, 25])), textureThreshold=int(rng.choice([10, 0, 100])),
              speckleWindowSize=int(rng.choice([100, 0, 30])), speckleRange=int(rng.choice([32, 2])), disp12MaxDiff=int(rng.choice([1, -1, 0])))
    Ls, Rs = pkg.synth.make_stream(seed % 1000, n, W, H, D)
    dL, dR = torch.from_numpy(Ls).cuda(), torch.from_numpy(Rs).cuda()
    dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=n,
                       **{k: v for k, v in kw.items() if k not in ("numDisparities", "blockSize")})
    for rep in range(3):
        dD.fill_(12345)
        m.compute_device(dL, dR, dD, st)
        torch.cuda.synchronize()
    variants[m.search_variant] = variants.get(m.search_variant, 0) + 1
    got = dD.cpu().numpy()
    m.close()
    for i in sorted({0, n // 2, n - 1}):
        want = orc.bm_compute(Ls[i], Rs[i], nthreads=8, **kw)
        if not np.array_equal(got[i], want):
            bad += 1
            print("MISMATCH ring seed", seed, W, H, n, i, kw, int((got[i] != want).sum()))
print("ring part done, total mismatches", bad, variants)
sys.exit(1 if bad else 0)

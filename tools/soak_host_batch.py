#!/usr/bin/env python3
"""Random soak of the host entry points (run on the GPU box): rtdm_bm_compute_batch on pageable and page-locked frames, contiguous
and as pitched views of wider planes (guard bands must stay untouched), batches below / at / above max_batch, against
rtdm_bm_compute frame by frame and the oracle.    python tools/soak_host_batch.py [first_seed=9000] [count=120]"""
import importlib, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
orc.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
for seed in range(first, first + count):
    if (seed - first) % 20 == 0: print("case", seed - first, "mismatches so far", bad, flush=True)
    rng = np.random.default_rng(seed)
    D = int(rng.choice([16, 32, 64, 96, 128])); w = int(rng.choice([5, 7, 9, 11, 15]))
    W = D + w + int(rng.integers(20, 300)); H = int(rng.integers(w + 10, 140))
    n = int(rng.choice([1, 2, 3, 7, 16, 17, 33])); mb = int(rng.choice([1, 4, 16, 40]))
    padx, pady = int(rng.choice([0, 0, 7, 16])), int(rng.choice([0, 3]))
    pinned = bool(rng.random() < 0.5)
    kw = dict(minDisparity=int(rng.choice([0, 0, -4, 3])), uniquenessRatio=int(rng.choice([10, 0])), speckleWindowSize=int(rng.choice([100, 0])),
              disp12MaxDiff=int(rng.choice([1, -1])))
    try:
        Ls, Rs = pkg.synth.make_stream(seed % 5000, n, W, H, D)
        def host(shape, dtype, fill):
            t = torch.full(shape, fill, dtype=dtype)
            return (t.pin_memory() if pinned else t).numpy()
        bigL = host((n, H + pady, W + padx), torch.uint8, 77); bigR = host((n, H + pady, W + padx), torch.uint8, 78)
        bigD = host((n, H + pady, W + padx), torch.int16, 12345)
        vL, vR, vD = bigL[:, :H, :W], bigR[:, :H, :W], bigD[:, :H, :W]
        vL[...] = Ls; vR[...] = Rs
        m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=mb, **kw)
        m.compute_batch(vL, vR, out=vD)
        one = m.compute(Ls[n - 1], Rs[n - 1])
        m.close()
        ok = np.array_equal(vD[n - 1], one)
        for i in sorted({0, n // 2, n - 1}):
            ok = ok and np.array_equal(vD[i], orc.bm_compute(Ls[i], Rs[i], numDisparities=D, blockSize=w, nthreads=8, **kw))
        guard = (padx == 0 or np.all(bigD[:, :, W:] == 12345)) and (pady == 0 or np.all(bigD[:, H:, :] == 12345))
        if not (ok and guard):
            bad += 1; print("MISMATCH seed", seed, W, H, D, w, n, mb, padx, pady, pinned, kw, "values", ok, "guard band", guard, flush=True)
    except Exception:          # noqa: BLE001
        bad += 1
        print("ERROR seed", seed, W, H, D, w, n, mb, padx, pady, pinned, traceback.format_exc()[-500:], flush=True)
print("checked", count, "host batches, mismatches", bad, flush=True)
sys.exit(1 if bad else 0)

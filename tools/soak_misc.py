#!/usr/bin/env python3
"""Extended random parity soak of the rows either side of the matcher (run on the GPU box): the object detection (HSV range,
open/close, external components, boxes, union ROI) and the morphology filter alone, with the generators of
tests/test_gpu_objects.py / tests/test_gpu_fuzz.py on other seeds.   python tools/soak_misc.py [first_seed=1000] [count=200]"""
import importlib, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
assert torch.cuda.is_available()
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
orc.build()
import test_gpu_objects as tobj
import test_gpu_fuzz as tf
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(first, first + count):
    if (seed - first) % 50 == 0: print("case", seed - first, "mismatches so far", bad, flush=True)
    for name, fn in (("objects", lambda s: tobj.test_detection_matches_the_oracle.__wrapped__(pkg, orc, s) if hasattr(tobj.test_detection_matches_the_oracle, "__wrapped__") else tobj.test_detection_matches_the_oracle(pkg, orc, s)),
                     ("morphology", lambda s: tf.test_random_morphology_shapes(pkg, orc, s))):
        try:
            fn(seed)
        except AssertionError as e:
            bad += 1
            print("MISMATCH", name, "seed", seed, str(e)[:200], flush=True)
        except Exception:          # noqa: BLE001
            bad += 1
            print("ERROR", name, "seed", seed, traceback.format_exc()[-400:], flush=True)
# rectification: random frame sizes, maps that point far outside / at the edges / at every fractional offset, random crops, batches
for seed in range(first, first + max(1, count // 2)):
    if (seed - first) % 50 == 0: print("rectify case", seed - first, "mismatches so far", bad, flush=True)
    rng = np.random.default_rng(seed + 31337)
    W, H = int(rng.integers(1, 300)), int(rng.integers(1, 160))
    maps = []
    for k in range(2):
        maps.append(np.stack([rng.integers(-4, W + 3, (H, W)), rng.integers(-4, H + 3, (H, W))], -1).astype(np.int16))
        maps.append(rng.integers(0, 1024, (H, W)).astype(np.uint16))
    n = int(rng.choice([1, 2, 3]))
    L = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8); R = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    rw, rh = int(rng.integers(1, W + 1)), int(rng.integers(1, H + 1))
    roi = (int(rng.integers(0, W - rw + 1)), int(rng.integers(0, H - rh + 1)), rw, rh)
    try:
        r = pkg.HIPRectifier(*maps, roi=roi, max_batch=n)
        dl = torch.empty((n, roi[3], roi[2]), dtype=torch.uint8, device="cuda"); dr = torch.empty_like(dl)
        r.gray_device(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), dl, dr)
        torch.cuda.synchronize()
        ok = all(np.array_equal(dl[i].cpu().numpy(), orc.rectify_gray(L[i], maps[0], maps[1], roi)) and
                 np.array_equal(dr[i].cpu().numpy(), orc.rectify_gray(R[i], maps[2], maps[3], roi)) for i in range(n))
        ok = ok and np.array_equal(r.rgb(L[0], 1), orc.rectify_rgb(L[0], maps[2], maps[3], roi))
        r.close()
        if not ok:
            bad += 1; print("MISMATCH rectify seed", seed, W, H, n, roi, flush=True)
    except Exception:          # noqa: BLE001
        bad += 1
        print("ERROR rectify seed", seed, W, H, n, roi, traceback.format_exc()[-400:], flush=True)
# depth statistics behind the matcher (/16 + reprojection + per-region mean Z): random disparity maps (with invalid pixels), Q, masks, regions
for seed in range(first, first + max(1, count // 2)):
    if (seed - first) % 50 == 0: print("depth case", seed - first, "mismatches so far", bad, flush=True)
    rng = np.random.default_rng(seed + 4242)
    W, H = int(rng.integers(8, 400)), int(rng.integers(4, 200))
    d = rng.integers(-16, 64 * 16, (H, W)).astype(np.int16)
    d[rng.random((H, W)) < 0.3] = -16                                   # FILTERED
    d[rng.random((H, W)) < 0.05] = 0                                    # disparity 0: Z = inf in the library, skipped
    mask = ((rng.random((H, W)) < float(rng.choice([0.2, 0.6, 1.0]))) * 255).astype(np.uint8)
    Q = np.array([[1, 0, 0, -W / 2 - 0.3], [0, 1, 0, -H / 2 - 0.8], [0, 0, 0, float(rng.uniform(100, 900))], [0, 0, 1 / float(rng.uniform(1, 20)), float(rng.choice([0.0, 0.0, 0.37]))]])
    regions = []
    for _ in range(int(rng.integers(1, 7))):
        rw, rh = int(rng.integers(1, W + 1)), int(rng.integers(0, H + 1))
        regions.append((int(rng.integers(0, W - rw + 1)), int(rng.integers(0, H - max(rh, 1) + 1)), rw, rh))
    unit = float(rng.choice([25.0, 1.0, 108.0]))
    try:
        mean, cnt = pkg.depth_stats_device(torch.from_numpy(d).cuda(), Q, torch.from_numpy(mask).cuda(), regions, unit)
        wm, wc = orc.depth_stats(d, Q, mask, regions, unit)
        if not (np.array_equal(cnt, wc) and np.allclose(mean, wm, rtol=1e-9, atol=0, equal_nan=True)):
            bad += 1; print("MISMATCH depth seed", seed, W, H, regions, list(cnt), list(wc), list(mean), list(wm), flush=True)
    except Exception:          # noqa: BLE001
        bad += 1
        print("ERROR depth seed", seed, W, H, regions, traceback.format_exc()[-500:], flush=True)
print("checked", count, "seeds x (objects, morphology) +", max(1, count // 2), "x (rectification, depth statistics), mismatches", bad, flush=True)
sys.exit(1 if bad else 0)

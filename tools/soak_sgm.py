#!/usr/bin/env python3
"""Extended random parity soak of the StereoSGBM path (run on the GPU box): random frame shapes (partial / single / many strips
of the row-synchronous sweeps), every numDisparities class, both modes, block sizes on both sides of the fused pixel-cost +
block-sum kernel, the library's parameter coercions, device batches of 1-5 pairs -- against oracle/sgm_oracle.c, tolerance 0.
    python tools/soak_sgm.py [first_seed=500000] [count=300]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
orc.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"       # python tools/soak_sgm.py FIRST COUNT wide: frames 640-1280 wide
big = len(sys.argv) > 3 and sys.argv[3] == "big"         # python tools/soak_sgm.py FIRST COUNT big: windows > 17
st = torch.cuda.current_stream().cuda_stream
bad = 0
sweeps = 0
for seed in range(first, first + count):
    if (seed - first) % 50 == 0: print("case", seed - first, "mismatches so far", bad, flush=True)
    rng = np.random.default_rng(seed)
    D = int(rng.choice([16, 32, 48, 64, 80, 96, 128, 160, 192, 256]))
    bs = int(rng.choice([1, 3, 5, 5, 7, 9, 11, 4, 6, 13]))
    minD = int(rng.choice([0, 0, 0, 3, -5]))
    W = D + abs(minD) + int(rng.choice([9, 17, 33, 40, 64, 65, 97, 130, 200, 333]))
    H = int(rng.integers(3, 70))
    n = int(rng.choice([1, 1, 2, 3, 5]))
    if wide:                                     # many strips per frame, batches large enough to change the strip width
        W = int(rng.choice([640, 901, 1280, 1279])); H = int(rng.integers(8, 48)); n = int(rng.choice([1, 2, 6, 17]))
    kw = dict(blockSize=bs, minDisparity=minD, uniquenessRatio=int(rng.choice([10, 0, 25, -1])),
              speckleWindowSize=int(rng.choice([100, 0, 20])), speckleRange=int(rng.choice([32, 1, 2])),
              disp12MaxDiff=int(rng.choice([1, -1, 2])), P1=int(rng.choice([600, 8, 100, 0])),
              P2=int(rng.choice([2400, 700, 3000, 0, 20000])), paths=int(rng.choice([8, 5])))
    if big:                                      # windows > 17: k_sgm_box_any with the cost check; a frame whose costs would wrap is refused by both
        kw["blockSize"] = int(rng.choice([19, 21, 25, 18, 33]))
    Ls, Rs = pkg.synth.make_stream(seed % 100000, n, W, H, D)
    if rng.random() < 0.2:                       # plateaus: many exact ties
        Ls = (Ls // 32 * 32).astype(np.uint8); Rs = (Rs // 32 * 32).astype(np.uint8)
    try:
        m = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, max_batch=n, **kw)
    except Exception as e:          # noqa: BLE001 -- a refused parameter set must be refused by the oracle too
        try:
            orc.sgm_compute(Ls[0], Rs[0], numDisparities=D, **kw)
            bad += 1; print("REFUSED ONLY BY THE DEVICE seed", seed, W, H, D, kw, repr(e)[:100])
        except Exception:           # noqa: BLE001
            pass
        continue
    dL, dR = torch.from_numpy(Ls).cuda(), torch.from_numpy(Rs).cuda()
    dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    refused = False
    for rep in range(2):
        dD.fill_(12345)
        try:
            m.compute_device(dL, dR, dD, st)
        except Exception as e:       # noqa: BLE001 -- a frame whose block cost + P2 passes 32767: the oracle must refuse it as well
            refused = True
            torch.cuda.synchronize()
            try:
                for i in range(n): orc.sgm_compute(Ls[i], Rs[i], numDisparities=D, **kw)
                bad += 1; print("FRAME REFUSED ONLY BY THE DEVICE seed", seed, W, H, D, n, kw, repr(e)[:80])
            except ValueError:
                pass
            break
        torch.cuda.synchronize()
    if refused:
        m.close(); continue
    got = dD.cpu().numpy()
    sw, gave_up = m.pass_stats()
    sweeps += sw
    m.close()
    if gave_up:
        bad += 1; print("SWEEP GAVE UP seed", seed, W, H, D, n, kw)
    for i in range(n):
        try:
            want = orc.sgm_compute(Ls[i], Rs[i], numDisparities=D, **kw)
        except ValueError:
            bad += 1; print("FRAME REFUSED ONLY BY THE ORACLE seed", seed, W, H, D, n, i, kw); continue
        if not np.array_equal(got[i], want):
            bad += 1
            print("MISMATCH seed", seed, W, H, D, n, i, kw, int((got[i] != want).sum()))
print("checked", count, "configurations, mismatches", bad, "row-synchronous sweeps launched", sweeps, flush=True)
sys.exit(1 if bad else 0)

"""Shared inputs of the rectification tests: the reference's calibration numbers (tests/golden/calib.json, data
extracted from /root/reference/backup/*/{intrinsics,extrinsics}.yml by tests/golden/make_calib.py), the crop the
reference derives from them (main.cpp:80-85) and a synthetic RGB sensor frame."""
import json
import os

import numpy as np

_CAL = None


def calib(res):
    global _CAL
    if _CAL is None:
        _CAL = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "calib.json")))
    c = _CAL[res]
    m = lambda k: np.array(c[k]["data"], np.float64).reshape(c[k]["rows"], c[k]["cols"])  # noqa: E731
    out = {k: m(k) for k in ("M1", "D1", "M2", "D2", "R1", "R2", "P1", "P2", "Q")}
    out["W"], out["H"] = int(c["Width"]), int(c["Height"])
    r1, r2 = c["ROI1"], c["ROI2"]
    # roif (main.cpp:80-85): max of the origins, min of the sizes
    out["roi"] = (max(r1[0], r2[0]), max(r1[1], r2[1]), min(r1[2], r2[2]), min(r1[3], r2[3]))
    return out


def maps(oracle, res):
    c = calib(res)
    l1, l2 = oracle.init_undistort_rectify_map(c["M1"], c["D1"], c["R1"], c["P1"], c["W"], c["H"])
    r1, r2 = oracle.init_undistort_rectify_map(c["M2"], c["D2"], c["R2"], c["P2"], c["W"], c["H"])
    return c, (l1, l2, r1, r2)


def rgb_pair(synth, seed, W, H):
    """Two textured RGB frames (channels = differently seeded noise images)."""
    a, b = synth.make_pair(synth.STREAM_SEED + 5000 + seed, W, H, 16)
    c, d = synth.make_pair(synth.STREAM_SEED + 5100 + seed, W, H, 16)
    left = np.ascontiguousarray(np.stack([a, c, ((a.astype(np.int32) + d) // 2).astype(np.uint8)], -1))
    right = np.ascontiguousarray(np.stack([b, d, ((b.astype(np.int32) + c) // 2).astype(np.uint8)], -1))
    return left, right


def red_scene(synth, seed, W, H, D=32, nblobs=6):
    """Sensor RGB frames for the whole per-frame chain: a synthetic stereo pair as the gray texture, with saturated red
    objects (hue 0, S > 150: what estimator.cpp:110-115 filters for) painted over it -- ellipses, one ring with a
    nested blob, one object touching the frame edge, a few specks that the 10x10 opening must erase."""
    L, R = synth.make_pair(synth.STREAM_SEED + 7000 + seed, W, H, D)
    rng = np.random.default_rng(900 + seed)
    yy, xx = np.mgrid[0:H, 0:W]
    m = np.zeros((H, W), bool)
    for _ in range(nblobs):
        cx, cy = rng.integers(W // 5, 4 * W // 5), rng.integers(H // 4, 3 * H // 4)
        a, b = rng.integers(W // 30 + 6, W // 10 + 8), rng.integers(H // 30 + 6, H // 8 + 8)
        m |= ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1
    cx, cy, r = W // 2, H // 2, min(W, H) // 6
    d2 = (xx - cx) ** 2 + (yy - cy) ** 2
    m &= ~(d2 <= (r + 14) ** 2)
    m |= (d2 <= (r + 12) ** 2) & (d2 >= (r - 2) ** 2)            # ring ...
    m |= d2 <= (r // 3) ** 2                                      # ... with a nested blob
    m[H // 3:H // 3 + 40, :W // 4] |= xx[H // 3:H // 3 + 40, :W // 4] < W // 5    # reaches the left frame edge
    for _ in range(30):
        x, y = rng.integers(0, W - 4), rng.integers(0, H - 4)
        m[y:y + 3, x:x + 3] = True                                # specks
    out = []
    for g in (L, R):
        g16 = g.astype(np.int32)
        rgb = np.stack([g, g, g], -1)
        red = np.stack([150 + g16 // 3, g16 // 6, g16 // 6], -1).astype(np.uint8)
        rgb[m] = red[m]
        out.append(np.ascontiguousarray(rgb))
    return out[0], out[1]

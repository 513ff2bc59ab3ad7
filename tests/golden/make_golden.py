"""Regenerates tests/golden/*.npz.

The reference holds no fixtures for this path (no tests, no images; its arithmetic lives in OpenCV,
which is absent here), so these vectors come from this repository's own oracle (oracle/bm_oracle.c,
oracle/morph_oracle.c) after it passed the brute-force cross-check.  They freeze the oracle's
behaviour so that (a) an edit to the oracle cannot silently move the goal posts and (b) the HIP
path can be checked on the GPU box against data rather than against code.  PARITY UNPINNED against
the real SWMatcherKonolige.

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc  # noqa: E402

synth = importlib.import_module("rt-depth-map_amd.synth")

BM_CASES = {
    # name: (W, H, seed offset, params)
    "bm_64x48_d16_w5": (64, 48, 1, dict(numDisparities=16, blockSize=5)),
    "bm_96x64_d32_w7": (96, 64, 2, dict(numDisparities=32, blockSize=7)),
    "bm_97x65_d16_w9_odd": (97, 65, 3, dict(numDisparities=16, blockSize=9)),
    "bm_128x96_d32_w13_ref_literals": (128, 96, 4, dict(numDisparities=32, blockSize=13)),
    "bm_96x64_d16_w7_raw": (96, 64, 5, dict(numDisparities=16, blockSize=7, disp12MaxDiff=-1, speckleWindowSize=0)),
    "bm_96x64_d16_w7_mind4": (96, 64, 6, dict(numDisparities=16, blockSize=7, minDisparity=4)),
    "bm_96x64_d16_w7_mindneg": (96, 64, 7, dict(numDisparities=16, blockSize=7, minDisparity=-5)),
    "bm_160x120_d64_w9": (160, 120, 8, dict(numDisparities=64, blockSize=9)),
    "bm_120x80_d16_w7_roi": (120, 80, 9, dict(numDisparities=16, blockSize=7, roi1=(30, 20, 70, 40))),
}


def main():
    for name, (W, H, so, kw) in BM_CASES.items():
        D = kw["numDisparities"]
        L, R = synth.make_pair(synth.STREAM_SEED + 1000 + so, W, H, D)
        disp = orc.bm_compute(L, R, **kw)
        keys = sorted(kw)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), left=L, right=R, disp=disp,
                            param_names=np.array(keys), param_values=np.array([str(kw[k]) for k in keys]))
        print(name, "valid=%.3f" % (disp != (kw.get("minDisparity", 0) - 1) * 16).mean())
    rng = np.random.default_rng(42)
    for name, (W, H) in {"morph_64x48": (64, 48), "morph_233x156": (233, 156)}.items():
        mask = ((rng.random((H, W)) < 0.55) * 255).astype(np.uint8)
        mask[H // 4:H // 2, W // 4:W // 2] = 255
        gray = rng.integers(0, 256, (H, W), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), mask=mask, mask_out=orc.morph_open_close(mask),
                            gray=gray, gray_out=orc.morph_open_close(gray))
        print(name)


if __name__ == "__main__":
    main()

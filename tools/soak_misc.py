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
print("checked", count, "seeds x (objects, morphology), mismatches", bad, flush=True)
sys.exit(1 if bad else 0)

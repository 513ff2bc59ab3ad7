#!/usr/bin/env python3
"""One configuration's batched device call a few times (to be run under rocprofv3): run_config.py W H D w batch [reps]."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("rt-depth-map_amd")
W, H, D, w, B = [int(a) for a in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 4
dL = torch.empty((B, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
dD = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
pkg.synth_pairs_device(dL, dR, 0, D)
m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=B)
st = torch.cuda.current_stream().cuda_stream
m.set_profiling(True)
for _ in range(reps): m.compute_device(dL, dR, dD, st)
torch.cuda.synchronize()
t = m.stage_times()
print(m.search_variant, {k: round(v["total_ms"] / max(1, v["launches"]), 4) for k, v in t.items()})
m.close()

"""Workload for rocprofv3 (tools/prof_sgm.sh): BASELINE config 5, 1280x720 D=128 blockSize 5, n pairs per call, `calls` calls.
    python3 tools/prof_sgm.py [paths=8] [n=4] [calls=3]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("rt-depth-map_amd")
paths = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
W, H, D = 1280, 720, 128
dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
pkg.synth_pairs_device(dL, dR, 0, D)
sg = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, max_batch=n, paths=paths)
st = torch.cuda.current_stream().cuda_stream
for _ in range(calls): sg.compute_device(dL, dR, dD, st)
torch.cuda.synchronize()

import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("rt-depth-map_amd")
n, W, H, D = 4, 1280, 720, 128
dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
pkg.synth_pairs_device(dL, dR, 0, D)
sg = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, max_batch=n)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): sg.compute_device(dL, dR, dD, st)
torch.cuda.synchronize()

"""Workload for rocprofv3 (tools/prof_misc.sh): the kernels either side of the matcher at their stated sizes --
k_morph_open_close on 64 x 1280x720 masks, k_rectify_gray on 64 raw 1280x720 RGB pairs (reference calibration), and the
object detector (k_hsv_inrange, k_morph_open_close on ONE crop, k_cc_*) on the 934x404 crop -- two parts, profiled in separate
runs because k_morph_open_close appears in both with different shapes.    python3 tools/prof_misc.py filters|objects [calls=3]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
import rectify_util as ru
part = sys.argv[1] if len(sys.argv) > 1 else "filters"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
st = torch.cuda.current_stream().cuda_stream
n, W, H = 64, 1280, 720
d_in = (torch.rand((n, H, W), device="cuda") < 0.5).to(torch.uint8) * 255
d_out = torch.empty_like(d_in)
if part == "filters":
    mf = pkg.HIPMorphologicalFilter(W, H, 8, max_batch=n)
    for _ in range(calls): mf.run_device(d_in, d_out, st)
c, maps = ru.maps(orc, "1280x720")
x, y, rw, rh = c["roi"]
left, right = ru.rgb_pair(pkg.synth, 3, W, H)
dL = torch.from_numpy(left).cuda()[None].repeat(n, 1, 1, 1).contiguous(); dR = torch.from_numpy(right).cuda()[None].repeat(n, 1, 1, 1).contiguous()
gl = torch.empty((n, rh, rw), dtype=torch.uint8, device="cuda"); gr = torch.empty_like(gl)
if part == "filters":
    r = pkg.HIPRectifier(*maps, roi=c["roi"], max_batch=n)
    for _ in range(calls): r.gray_device(dL, dR, gl, gr, st)
else:
    sl, sr = ru.red_scene(pkg.synth, 1, W, H, 64)
    col = orc.rectify_rgb(sl, maps[0], maps[1], c["roi"])
    det = pkg.HIPObjectDetector(rw, rh)
    for _ in range(calls * 4): det.detect(col)
torch.cuda.synchronize()

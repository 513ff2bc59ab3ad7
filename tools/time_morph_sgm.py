#!/usr/bin/env python3
"""Times the fused morphology kernel and the SGM-8 path on device-resident batches (run on the GPU box)."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
out = {}
st = torch.cuda.current_stream().cuda_stream
# --- morphology, BASELINE config 3's post-filter -------------------------------------------------
n, W, H = 64, 1280, 720
d_in = (torch.rand((n, H, W), device="cuda") < 0.5).to(torch.uint8) * 255
d_out = torch.empty_like(d_in)
mf = pkg.HIPMorphologicalFilter(W, H, 8, max_batch=n)
for _ in range(3): mf.run_device(d_in, d_out, st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): mf.run_device(d_in, d_out, st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
m0 = d_in[0].cpu().numpy()
t0 = time.perf_counter(); want = orc.morph_open_close(m0); tc = time.perf_counter() - t0
out["morph_open_close_1280x720"] = {"us_per_frame": round(dt / n * 1e6, 2), "frames_per_s": round(n / dt), "algorithmic_GBps": round(2 * W * H * n / dt / 1e9, 1),
                                     "hbm_frac_of_8TBps": round(2 * W * H * n / dt / 8e12, 4), "cpu_oracle_ms_per_frame_1thread": round(tc * 1e3, 1),
                                     "bit_exact": bool(np.array_equal(d_out[0].cpu().numpy(), want))}
# --- StereoSGBM, BASELINE config 5: MODE_HH (8 paths) and the reference's own MODE_SGBM (5), at 4 and at 16 pairs per call -----------
D = 128
for n in (4, 16):
    dL = torch.empty((n, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
    dD = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
    pkg.synth_pairs_device(dL, dR, 0, D)
    L, R = dL[0].cpu().numpy(), dR[0].cpu().numpy()
    for paths in (8, 5):
        sg = pkg.HIPSemiGlobalMatcher(numOfDisparities=D, width=W, height=H, max_batch=n, paths=paths)
        for _ in range(2): sg.compute_device(dL, dR, dD, st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 5 if n == 4 else 3
        for _ in range(reps): sg.compute_device(dL, dR, dD, st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        got = dD[0].cpu().numpy()
        rec = {"ms_per_pair": round(dt / n * 1e3, 3), "pairs_per_s": round(n / dt, 1), "pairs_per_call": n}
        if n == 4:
            t0 = time.perf_counter(); want = orc.sgm_compute(L, R, numDisparities=D, paths=paths); tc = time.perf_counter() - t0
            rec.update({"cpu_oracle_s_per_pair_1thread": round(tc, 2), "bit_exact_vs_oracle": bool(np.array_equal(got, want)),
                        "pixels_differing_pct": float((got != want).mean() * 100), "valid_fraction": float((got != -16).mean())})
        sweeps, gave_up = sg.pass_stats()
        rec.update({"row_synchronous_sweeps_per_call": sweeps // (2 + reps), "sweep_gave_up": gave_up})
        sg.close()
        out["sgm%d_1280x720_d128_bs5%s" % (paths, "" if n == 4 else "_batch%d" % n)] = rec
    del dL, dR, dD
print(json.dumps(out, indent=1))

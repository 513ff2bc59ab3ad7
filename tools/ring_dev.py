#!/usr/bin/env python3
"""Development loop for the search kernels on the GPU box: for each library build named on the command line ("default" or a
`make variant NAME=...` build, loaded through RTDM_LIB_VARIANT in a process of its own) check a few frames of a configuration
against the oracle and time the search stage with HIP events.

    python tools/ring_dev.py [--cfg D,w[,W,H]] [--batch 256] [--steps 10] default dev ...
"""
import argparse, importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(a):
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    pkg = importlib.import_module("rt-depth-map_amd")
    from oracle import oracle as orc
    orc.build()
    out = []
    for cfg in a.cfg:
        v = [int(x) for x in cfg.split(",")]
        D, w = v[0], v[1]
        W, H = (v[2], v[3]) if len(v) >= 4 else (1280, 720)
        B = a.batch
        dL = torch.empty((B, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
        dD = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for i0 in range(0, B, 256):
            n = min(256, B - i0)
            pkg.synth_pairs_device(dL[i0:i0 + n], dR[i0:i0 + n], i0, D, stream=st)
        m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=B)
        for _ in range(3): m.compute_device(dL, dR, dD, st)
        torch.cuda.synchronize(); m.set_profiling(True); m.reset_stage_times()
        import time
        t0 = time.perf_counter()
        for _ in range(a.steps): m.compute_device(dL, dR, dD, st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        t = m.stage_times()
        idx = sorted({0, B // 2, B - 1})
        ok = all(np.array_equal(dD[i].cpu().numpy(), orc.bm_compute(dL[i].cpu().numpy(), dR[i].cpu().numpy(), numDisparities=D, blockSize=w, nthreads=32))
                 for i in idx)
        out.append(dict(cfg=cfg, variant=m.search_variant, exact=bool(ok), pairs_per_s=round(B * a.steps / dt, 1),
                        stage_ms={k: round(x["total_ms"] / max(1, x["launches"]), 4) for k, x in t.items()}))
        m.close()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", action="append")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    a.cfg = a.cfg or ["64,9"]
    if a.child:
        child(a); sys.exit(0)
    for lib in a.libs or ["default"]:
        env = dict(os.environ)
        env.pop("RTDM_LIB_VARIANT", None)
        if lib != "default": env["RTDM_LIB_VARIANT"] = lib
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--batch", str(a.batch), "--steps", str(a.steps)] + sum([["--cfg", c] for c in a.cfg], [])
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        res = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if p.returncode or not res:
            print("%-12s FAILED rc=%d %s" % (lib, p.returncode, p.stderr[-600:]), flush=True); continue
        for r in json.loads(res[0][7:]):
            print("%-12s cfg %-16s %-16s exact=%-5s %9.1f pairs/s  search %.4f lr %.4f spk %.4f pre %.4f ms" % (
                lib, r["cfg"], r["variant"], r["exact"], r["pairs_per_s"], r["stage_ms"]["search"], r["stage_ms"]["lrcheck"],
                r["stage_ms"]["speckle"], r["stage_ms"]["prefilter"]), flush=True)

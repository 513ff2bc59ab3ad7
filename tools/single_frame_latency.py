import importlib, sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
pkg = importlib.import_module("rt-depth-map_amd")
synth = pkg.synth
for (W,H,D,w) in ((233,156,32,7),(232,156,32,7),(240,156,32,7),(320,240,32,7),(934,404,64,9),(936,404,64,9)):
    L,R = synth.make_pair(1, W, H, D)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H)
    for _ in range(5): m.compute(L,R)
    m.set_profiling(True); m.reset_stage_times()
    t0=time.perf_counter()
    for _ in range(50): m.compute(L,R)
    dt=(time.perf_counter()-t0)/50
    st={k: round(v["total_ms"]/max(v["launches"],1),4) for k,v in m.stage_times().items()}
    print(W,H, round(dt*1e3,3),"ms", st, m.search_variant)
    m.close()

import importlib, sys, os, json, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
pkg = importlib.import_module("rt-depth-map_amd")
st = torch.cuda.current_stream().cuda_stream
for (W,H) in ((934,404),(936,404),(1280,720),(1024,404)):
    n, D, w = 128, 64, 9
    dL = torch.empty((n,H,W),dtype=torch.uint8,device="cuda"); dR=torch.empty_like(dL); dD=torch.empty((n,H,W),dtype=torch.int16,device="cuda")
    pkg.synth_pairs_device(dL,dR,0,D)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=n)
    for _ in range(3): m.compute_device(dL,dR,dD,st)
    torch.cuda.synchronize(); m.set_profiling(True); m.reset_stage_times()
    t0=time.perf_counter()
    for _ in range(10): m.compute_device(dL,dR,dD,st)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print(W,H, round(dt/n*1e6,2),"us/pair", {k: round(v["total_ms"]/max(v["launches"],1),3) for k,v in m.stage_times().items()})
    m.close()

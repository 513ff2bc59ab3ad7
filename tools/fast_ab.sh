#!/bin/bash
# A/B of k_search_fast build variants on the large-D configurations (run on the GPU box): each flag set is built as a VARIANT
# library (`make variant`; the shipped library is never touched) and times the search stage of config 3 (d=128, 11x11) and of
# d=192 / d=256 with RTDM_RING=0 (so that k_search_fast runs), checking the bytes.
R=$GRAFT_REPO_ROOT
cd $R/rt-depth-map_amd
K=0
for V in "$@"; do
    K=$((K + 1)); F=$(echo "$V" | tr ',' ' ')
    make -s variant NAME=fast$K VSRC=k_search_fast VFLAGS="$F" 2> /dev/null || { echo "build failed: $V"; continue; }
    export RTDM_LIB_VARIANT=fast$K RTDM_RING=0
    python - "$V" <<'PY'
import importlib, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
pkg = importlib.import_module("rt-depth-map_amd")
from oracle import oracle as orc
st = torch.cuda.current_stream().cuda_stream
for (D, w) in ((128, 11), (192, 13), (256, 15)):
    W, H, B = 1280, 720, 32
    dL = torch.empty((B, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
    dD = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
    pkg.synth_pairs_device(dL, dR, 0, D)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=w, width=W, height=H, max_batch=B)
    for _ in range(3): m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize(); m.set_profiling(True); m.reset_stage_times()
    for _ in range(5): m.compute_device(dL, dR, dD, st)
    torch.cuda.synchronize(); t = m.stage_times()
    ok = np.array_equal(dD[3].cpu().numpy(), orc.bm_compute(dL[3].cpu().numpy(), dR[3].cpu().numpy(), numDisparities=D, blockSize=w, nthreads=32))
    print("%-24s d=%d w=%d search %.4f ms per %d pairs  %s exact=%s" % (sys.argv[1], D, w, t["search"]["total_ms"] / t["search"]["launches"], B, m.search_variant, ok), flush=True)
    m.close()
PY
done
rm -rf $R/rt-depth-map_amd/build_fast* $R/rt-depth-map_amd/lib/variants/librtdm_hip_fast*.so

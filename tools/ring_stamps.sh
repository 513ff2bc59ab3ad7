#!/bin/bash
# Diagnostic build of k_search_ring with s_memtime stamps (run on the GPU box), as a VARIANT library (never the shipped one):
# prints the share of each segment of a row group.  Extra hipcc flags may follow.
R=$GRAFT_REPO_ROOT
cd $R/rt-depth-map_amd && make -s variant NAME=stamps VSRC=k_search_ring VFLAGS="-DRING_STAMPS $*" 2> /dev/null || exit 1
cd $R && RTDM_LIB_VARIANT=stamps python - <<'PY'
import ctypes, importlib, torch
pkg = importlib.import_module("rt-depth-map_amd")
lib = pkg.binding.lib()
W, H, D, B = 1280, 720, 64, 64
dL = torch.empty((B, H, W), dtype=torch.uint8, device="cuda"); dR = torch.empty_like(dL)
dD = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
pkg.synth_pairs_device(dL, dR, first_frame=0, numDisparities=D, stream=st)
m = pkg.HIPMatcher(numOfDisparities=D, blockSize=9, width=W, height=H, max_batch=B)
for _ in range(3): m.compute_device(dL, dR, dD, st)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
lib.rtdm_debug_ring_stamps(out, 1)
m.compute_device(dL, dR, dD, st); torch.cuda.synchronize()
lib.rtdm_debug_ring_stamps(out, 0)
v = list(out); tot = sum(v)
names = ["0 before group", "1 LDS reads of the rows", "2 row steps (SAD+commit)", "3", "4", "5 selection", "6 stores", "7"]
for n, x in zip(names, v): print("%-26s %14d  %5.1f %%" % (n, x, 100.0 * x / max(1, tot)))
PY
rm -rf $R/rt-depth-map_amd/build_stamps $R/rt-depth-map_amd/lib/variants/librtdm_hip_stamps.so

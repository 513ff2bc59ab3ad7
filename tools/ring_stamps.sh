#!/bin/bash
# Diagnostic build of k_search_ring with s_memtime stamps (run on the GPU box): prints the share of each segment of a row pair.
R=$GRAFT_REPO_ROOT
cd $R/rt-depth-map_amd
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -DRING_STAMPS "$@" -c csrc/k_search_ring.hip -o build/k_search_ring.o 2> /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/librtdm_hip.so build/*.o
cd $R && python - <<'PY'
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
names = ["0 before pair", "1 LDS reads row A", "2 step A (SAD+commit)", "3 LDS reads row B", "4 step B (SAD+commit)", "5 swaps", "6 selection+stores", "7"]
for n, x in zip(names, v): print("%-26s %14d  %5.1f %%" % (n, x, 100.0 * x / tot))
PY

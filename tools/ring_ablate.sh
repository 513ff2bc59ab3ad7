#!/bin/bash
# Timing-only ablations of k_search_ring (run on the GPU box): rebuilds the kernel with -DRING_ABL=n / other flags and times
# the search stage with bench.py.  Outputs are WRONG for n != 0 (bench.py's parity check fails by design); only the stage time matters.
R=$GRAFT_REPO_ROOT
cd $R/rt-depth-map_amd
for V in "$@"; do
    F=$(echo "$V" | tr ',' ' ')
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $F -c csrc/k_search_ring.hip -o build/k_search_ring.o 2> /dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/librtdm_hip.so build/*.o
    python $R/bench.py --no-cpu-baseline --steps 10 > /tmp/abl.json 2> /dev/null
    python - "$V" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/abl.json"))
    print("%-40s search %.4f ms  total %.1f pairs/s parity=%s" % (sys.argv[1], d["stage_ms_per_launch"]["search"], d["value"], d["parity_ok"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done

#!/bin/bash
# A/B of the two-lane ring kernel's selection (run on the GPU box): -DRING_SPLIT_SELECT=0 = the per-configuration default
# (transposing selection for D = 16 and D = 32 except w = 9, GroupSelect elsewhere), =1 = GroupSelect everywhere.
R=$GRAFT_REPO_ROOT
cd $R/rt-depth-map_amd
for V in "-DRING_SPLIT_SELECT=0" "-DRING_SPLIT_SELECT=1"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $V -c csrc/k_search_ring.hip -o build/k_search_ring.o 2> /dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/librtdm_hip.so build/*.o
    echo "== $V"
    python $R/tools/ab_ring.py 2>/dev/null | grep -v amdgpu.ids
done

#!/bin/bash
# A/B of the two-lane ring kernel's selection (run on the GPU box): -DRING_SPLIT_SELECT=0 = the per-configuration default
# (transposing selection for D = 16 and D = 32 except w = 9, GroupSelect / GroupSelectRec elsewhere), =1 = everywhere.
# Built as VARIANT libraries; the shipped library is never touched.
R=$GRAFT_REPO_ROOT
cd $R/rt-depth-map_amd
for V in 0 1; do
    make -s variant NAME=split$V VSRC=k_search_ring VFLAGS="-DRING_SPLIT_SELECT=$V" 2> /dev/null || exit 1
    echo "== -DRING_SPLIT_SELECT=$V"
    RTDM_LIB_VARIANT=split$V python $R/tools/ab_ring.py 2>/dev/null | grep -v amdgpu.ids
done
rm -rf build_split* lib/variants/librtdm_hip_split*.so

#!/bin/bash
# Timing-only ablations of k_search_ring (run on the GPU box): each flag set is built as a VARIANT library
# (`make variant`: lib/variants/librtdm_hip_ablK.so; the shipped lib/librtdm_hip.so and build/ are never touched) and timed
# with tools/ring_dev.py.  Outputs are WRONG for RING_ABL != 0 (ring_dev prints exact=False by design): only the stage time
# matters.  usage: tools/ring_ablate.sh "-DRING_ABL=1" "-DRING_ABL=4,-DRING_STATIC_SLOTS=0" ...
R=$GRAFT_REPO_ROOT
cd $R/rt-depth-map_amd
K=0; NAMES="default"
for V in "$@"; do
    K=$((K + 1)); F=$(echo "$V" | tr ',' ' ')
    make -s variant NAME=abl$K VSRC=k_search_ring VFLAGS="$F" 2> /dev/null || { echo "build failed: $V"; continue; }
    echo "abl$K = $V"; NAMES="$NAMES abl$K"
done
python $R/tools/ring_dev.py --batch ${ABL_BATCH:-256} --steps 10 --cfg ${ABL_CFG:-64,9} $NAMES
rm -rf build_abl* lib/variants/librtdm_hip_abl*.so

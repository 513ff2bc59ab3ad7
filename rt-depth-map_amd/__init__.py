"""rt-depth-map_amd: MI355X-native stereo block matching behind rt-depth-map's BlockMatcher.

Import through importlib (the directory name carries the reference's hyphen):
    pkg = importlib.import_module("rt-depth-map_amd")
The compute path is the HIP library rt-depth-map_amd/lib/librtdm_hip.so (C ABI: include/rtdm.h);
nothing in this package falls back to the CPU.
"""
from . import binding, synth  # noqa: F401
from .matcher import (HIPMatcher, HIPMorphologicalFilter, HIPObjectDetector, HIPRectifier,  # noqa: F401
                      HIPSemiGlobalMatcher, depth_stats_device, estimate_frame, synth_pairs_device)

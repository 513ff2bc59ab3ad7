"""Host side of the drop-in boundary (no GPU needed): the OpenCV-free C++ core builds, fails loudly
without a device, and the real adapters still match the reference's abstract classes."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "rt-depth-map_amd", "host")
LIBDIR = os.path.join(ROOT, "rt-depth-map_amd", "lib")
REF = "/root/reference"


def test_host_selftest_contract():
    exe = os.path.join(LIBDIR, "host_selftest")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "rt-depth-map_amd")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    import torch
    if not torch.cuda.is_available():
        assert "ctor_status=-3" in out.stdout and "compute_status=-3" in out.stdout
        assert "no CPU fallback" in out.stderr


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only mounted in the build container")
def test_adapters_match_reference_interfaces(tmp_path):
    # lay the adapter files out the way INTEGRATION.md installs them, then syntax-check them against
    # the reference's own BlockMatcher / VideoFilterDevice headers and the OpenCV type shim
    inc = tmp_path / "include"
    (inc / "stereo-matcher").mkdir(parents=True)
    (inc / "filter").mkdir()
    shutil.copy(os.path.join(HOST, "bm-hip.h"), inc / "stereo-matcher" / "bm-hip.h")
    shutil.copy(os.path.join(HOST, "mf-hip.h"), inc / "filter" / "mf-hip.h")
    shutil.copy(os.path.join(HOST, "sgbm-hip.h"), inc / "stereo-matcher" / "sgbm-hip.h")
    for src in ("bm-hip.cpp", "mf-hip.cpp", "sgbm-hip.cpp"):
        cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "shims"),
               "-I", str(inc), "-I", os.path.join(REF, "include"), "-I", HOST, os.path.join(HOST, src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_adapter_keeps_the_reference_constructor_shape():
    text = open(os.path.join(HOST, "bm-hip.h")).read()
    # SWMatcherKonolige's twelve arguments, same order (include/stereo-matcher/bm-sw.h:28-30)
    order = ["roi1", "roi2", "preFilterCap", "blockSize", "minDisparity", "textureThreshold", "numOfDisparities",
             "maxDisparity", "uniquenessRatio", "speckleWindowSize", "speckleRange", "disp12MaxDiff"]
    pos = [text.index(n) for n in order]
    assert pos == sorted(pos)
    assert "public BlockMatcher" in text
    assert "public VideoFilterDevice" in open(os.path.join(HOST, "mf-hip.h")).read()

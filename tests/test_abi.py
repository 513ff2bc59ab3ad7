"""No-GPU checks of the drop-in boundary: the library loads, exports every symbol that include/rtdm.h
declares, reports errors the way the header says, and refuses to run without a device."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, load


def header_functions():
    text = open(os.path.join(ROOT, "include", "rtdm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtdm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    B = load("binding")
    L = C.CDLL(B.LIB_PATH)
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "librtdm_hip.so lacks " + n
    assert sorted(B.EXPORTS) == names


def test_abi_version_and_strerror():
    B = load("binding")
    L = B.lib()
    assert L.rtdm_abi_version() == 3
    assert L.rtdm_strerror(0) == b"ok"
    assert b"no CPU fallback" in L.rtdm_strerror(-3)
    p = B.BMParams()
    L.rtdm_bm_default_params(C.byref(p), 192)
    # main.cpp:134-135 literals
    assert (p.preFilterCap, p.blockSize, p.minDisparity, p.textureThreshold, p.numDisparities, p.uniquenessRatio,
            p.speckleWindowSize, p.speckleRange, p.disp12MaxDiff) == (31, 13, 0, 10, 192, 10, 100, 32, 1)


def test_parameter_validation_precedes_device_use():
    B = load("binding")
    L = B.lib()
    h = C.c_void_p()
    for bad in (dict(numDisparities=20), dict(blockSize=8), dict(blockSize=3), dict(preFilterCap=0),
                dict(preFilterCap=64), dict(textureThreshold=-1), dict(uniquenessRatio=-1), dict(minDisparity=-2048),
                dict(minDisparity=2000, numDisparities=64)):
        p = B.make_params(**bad)
        assert L.rtdm_bm_create(C.byref(p), 64, 48, 1, 0, C.byref(h)) == -1, bad
    assert L.rtdm_bm_create(None, 64, 48, 1, 0, C.byref(h)) == -7
    p = B.make_params()
    assert L.rtdm_bm_create(C.byref(p), 0, 48, 1, 0, C.byref(h)) == -2


def test_no_device_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    pkg = load()
    with pytest.raises(pkg.binding.RtdmError) as e:
        pkg.HIPMatcher(width=64, height=48)
    assert e.value.status == -3
    with pytest.raises(pkg.binding.RtdmError):
        pkg.HIPMorphologicalFilter(64, 48)


def test_product_package_never_touches_the_oracle():
    pkgdir = os.path.join(ROOT, "rt-depth-map_amd")
    for dp, _, fs in os.walk(pkgdir):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "rtdm_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_header_is_plain_c_and_cpp():
    # the boundary is a C ABI: the header must compile as C99 (what cgo / a C caller sees) and as C++11 (the
    # reference's own dialect, Makefile.build: -std=c++11) without warnings
    import subprocess
    hdr = os.path.join(ROOT, "include", "rtdm.h")
    for cc, std, lang in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "c++")):
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", lang, hdr], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def _build_c_caller(tmp_path):
    import subprocess
    exe = str(tmp_path / "c_caller")
    libdir = os.path.join(ROOT, "rt-depth-map_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_caller.c"), "-L", libdir, "-lrtdm_hip", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_plain_c_program_links_and_fails_loudly_without_a_device(tmp_path):
    import subprocess
    import torch
    out = subprocess.run([_build_c_caller(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    if not torch.cuda.is_available():
        assert "create: -3" in out.stdout


@pytest.mark.gpu
def test_plain_c_program_matches_the_oracle(tmp_path):
    import subprocess
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    synth = load("synth")
    W, H, D, w = 320, 200, 32, 7
    L, R = synth.make_pair(synth.STREAM_SEED + 77, W, H, D)
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    src.write_bytes(L.tobytes() + R.tobytes())
    rc = subprocess.run([_build_c_caller(tmp_path), str(src), str(dst), str(W), str(H), str(D), str(w)], capture_output=True, text=True)
    assert rc.returncode == 0, rc.stdout + rc.stderr
    got = np.frombuffer(dst.read_bytes(), np.int16).reshape(H, W)
    assert np.array_equal(got, orc.bm_compute(L, R, numDisparities=D, blockSize=w, nthreads=8))

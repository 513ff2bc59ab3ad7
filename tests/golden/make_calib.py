"""Extracts the calibration NUMBERS (data, no code) of the reference's three camera set-ups into calib.json.

Run in the build container only:  python tests/golden/make_calib.py
Source: /root/reference/backup/<WxH>/{intrinsics,extrinsics}.yml (OpenCV FileStorage YAML 1.0).  The reference
reads M1 D1 M2 D2 Width Height ROI1 ROI2 R T from them (main.cpp:61-78) and recomputes R1 R2 P1 P2 Q with
stereoRectify; the files also hold the R1 R2 P1 P2 Q the calibration tool stored, which is what the tests use.
"""
import json
import os
import re

ROOT = "/root/reference/backup"


def parse(path):
    text = open(path).read()
    out = {}
    for m in re.finditer(r"^(\w+): !!opencv-matrix\s+rows: (\d+)\s+cols: (\d+)\s+dt: \w+\s+data: \[(.*?)\]", text, re.S | re.M):
        out[m.group(1)] = {"rows": int(m.group(2)), "cols": int(m.group(3)),
                           "data": [float(v) for v in m.group(4).replace("\n", " ").split(",")]}
    for m in re.finditer(r"^(\w+): \[(.*?)\]", text, re.S | re.M):
        out[m.group(1)] = [int(v) for v in m.group(2).split(",")]
    for m in re.finditer(r"^(\w+): ([-+0-9.eE]+)\s*$", text, re.M):
        out[m.group(1)] = float(m.group(2))
    return out


def main():
    res = {}
    for d in sorted(os.listdir(ROOT)):
        c = parse(os.path.join(ROOT, d, "intrinsics.yml"))
        c.update(parse(os.path.join(ROOT, d, "extrinsics.yml")))
        res[d] = c
    with open(os.path.join(os.path.dirname(__file__), "calib.json"), "w") as f:
        json.dump(res, f, indent=1)
    print({k: sorted(v) for k, v in res.items()})


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Run HERE after `gpurun -- bash tools/pmc_traffic.sh`: copies gpurun_out/pmc_traffic/{traffic.json,summary.txt} to
profiles/rNN_pmc_traffic.json / rNN_pmc_traffic_rocprofv3.txt, adds the git commit, and refuses if the device sources have
changed since the measurement (bench.py would mark the file stale anyway)."""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", "pmc_traffic")
d = json.load(open(os.path.join(src, "traffic.json")))
if d.get("kernel_source_sha") != bench.kernel_source_sha():
    sys.exit("device sources changed since the counters were collected: %s vs %s" % (d.get("kernel_source_sha"), bench.kernel_source_sha()))
d["git_commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
d["git_dirty"] = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "rt-depth-map_amd/csrc"], text=True).strip())
json.dump(d, open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(ROOT, "profiles", tag + "_pmc_traffic_rocprofv3.txt"))
print("wrote profiles/%s_pmc_traffic.json (commit %s, sources %s)" % (tag, d["git_commit"][:12], d["kernel_source_sha"]))

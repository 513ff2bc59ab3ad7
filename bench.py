#!/usr/bin/env python3
"""bench.py -- stereo-pairs/sec of the BlockMatcher hot path on MI355X.

Headline workload (BASELINE.json metric): 1280x720 rectified pairs, numDisparities 64, 9x9 SAD, every other
StereoBM knob at the reference's literals (main.cpp:134-135: cap 31, texture 10, uniqueness 10,
disp12MaxDiff 1, speckle 100/32), i.e. the whole cv::StereoBM::compute pipeline that
SWMatcherKonolige::compute (bm-sw.cpp:33-38) runs.  A "step" is one rtdm_bm_compute_device call
over a batch of --batch synthetic pairs (default 1024, the stream length of BASELINE config 4) that are
already resident in HBM.  `value` is measured over EXACTLY --steps steps.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

After the headline region, on rank 0 and outside `value`:
  * "sustained": the same step repeated until >= 2 s have passed (a region long enough for an outside sampler to see);
  * "configs": every other size north_star names -- 320x240 d=32 7x7, 640x480 d=64 9x9, 1280x720 d=128 11x11 + the
    morphological open/close (frame sizes: backup/{320x240,640x480,1280x720}/extrinsics.yml:56-57), and the reference's
    own parameter set d=192 13x13 (main.cpp:134-135, utils/cmdline-parser.cpp:22) -- each on a device-resident batch,
    >= 20 steps, three frames checked against the CPU oracle, with the oracle timed beside it on every host core; and
    BASELINE config 5, the SWSemiGlobalMatcher counterpart (cv::StereoSGBM restated, 1280x720 d=128, 16 pairs per call,
    MODE_HH and MODE_SGBM, one frame of each checked against the oracle, integer arithmetic: tolerance 0);
  * "single_frame": ms per frame for a caller that hands over one frame per call (estimator.cpp:56), host to host;
  * "cpu_baseline": the oracle (a port: bm-sw.cpp itself needs OpenCV) on the box's host cores, frame-parallel and
    row-striped, the better of the two reported.

--gpus N > 1 without a torch.distributed environment: this process touches no GPU; it starts N
ranks of itself (python -m torch.distributed.run, one process per GPU, RCCL), relays rank 0's JSON
line and exits with the ranks' status.  Under torchrun (WORLD_SIZE set) it is one of the ranks.

Multi-GPU: frames are independent (estimator.cpp:18-82 carries nothing across iterations), so each
rank synthesises and processes its own shard of the stream (weak scaling, no data-path collective);
timing is barrier + synchronize on both sides and the MAX over ranks.  --rccl-stream FRAMES is
BASELINE config 4 instead: rank 0 owns the frames, scatter -> compute -> gather over RCCL.

Three frames of the last step's output of every timed workload are compared with the CPU oracle
("parity_ok"); a mismatch exits non-zero.
"""
import argparse
import hashlib
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, D, BLOCK = 1280, 720, 64, 9
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
ALGO_BYTES_PER_PAIR = 4 * W * H  # read L + read R (u8) + write int16 disparity
PMC_CANDIDATES = ("r03_pmc_traffic.json", "r02_pmc_traffic.json")
# the quad-SAD issue floor of the packed search kernels: one v_qsad_pk_u16_u8 = 4 disparities x 4 window
# bytes of one pixel per lane, a quarter-rate instruction measured at 16.4 SIMD cycles per wave (tools/ubench_valu.hip,
# profiles/r02_ubench_valu_instruction_costs.txt); 256 CUs x 4 SIMDs
QSAD_CYCLES, N_SIMD = 16.4, 1024

# the other sizes north_star names (BASELINE.json configs 1-3) + the reference's real parameter set, which is d=192 13x13
# at EVERY resolution (utils/cmdline-parser.cpp:22; scale_to_width divides by the parser's own width, cmdline-parser.h:85-89)
CONFIGS = [
    dict(key="config1", workload="320x240 d=32 7x7", W=320, H=240, D=32, w=7, batch=2048),
    dict(key="config2", workload="640x480 d=64 9x9", W=640, H=480, D=64, w=9, batch=512),
    dict(key="config3", workload="1280x720 d=128 11x11 + 10x10-ellipse open/close (mf-sw.cpp:19-28)", W=1280, H=720, D=128, w=11,
         batch=128, morph=True),
    dict(key="reference_default", workload="1280x720 d=192 13x13 (main.cpp:134-135, cmdline-parser.cpp:22)", W=1280, H=720,
         D=192, w=13, batch=64),
]
# the reference's real call (estimator.cpp:33-36,54-56): views of size roif at its origin inside full-pitch planes;
# roif = (max x, max y, min w, min h) of ROI1/ROI2 of backup/1280x720/extrinsics.yml:56-57 (main.cpp:80-85)
REF_CROP = (192, 177, 934, 404)
REF_ROI1 = (300, 90, 320, 200)   # a union-of-objects box as find_relevant_matching_region would return (estimator.cpp:53)


def kernel_source_sha():
    """Content hash of the device sources: ties profiles/*_pmc_traffic.json to the kernels that were measured."""
    d = os.path.join(ROOT, "rt-depth-map_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def usable_cores():
    """Host cores this process may really use: the scheduler affinity, cut down to the cgroup's CPU quota (a GPU box hands
    a one-GPU job a share of its cores; 256 busy threads on a 16-CPU quota only throttle each other)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                  # cgroup v2: "max 100000" | "1600000 100000"
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())              # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(math.ceil(quota))))
    return n


def launch_ranks(n):
    """--gpus N without WORLD_SIZE: run N ranks of this script as child processes (no exec, no GPU use here)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout:                       # relay everything; remember the JSON line of rank 0
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line is not None:
        sys.stdout.write(line); sys.stdout.flush()
    if rc == 0 and line is None:
        rc = 3
    sys.exit(rc)


def oracle_ready(rank, dist):
    """Build the oracle library once per job (rank 0; the others wait): concurrent builds would race on the same file."""
    from oracle import oracle as orc
    if rank == 0:
        orc.build()
    if dist is not None:
        dist.barrier()


def oracle_frames(dL, dR, idx, threads, **kw):
    """The CPU oracle's disparity for frames idx of the device batch (the checker, never the thing measured)."""
    from oracle import oracle as orc
    kw.setdefault("numDisparities", D); kw.setdefault("blockSize", BLOCK)
    return [orc.bm_compute(dL[i].cpu().numpy(), dR[i].cpu().numpy(), nthreads=threads, **kw) for i in idx]


def cpu_frame_parallel(frames, kw, threads, budget_s):
    """One frame per thread (the way the reference's loop would be scaled out on a host, estimator.cpp:18-82 being stateless per
    frame): every worker runs whole single-threaded oracle calls on frames of the pool until the deadline.  The ctypes call
    releases the GIL.  Returns (pairs/s, frames done)."""
    from oracle import oracle as orc
    try:
        # the oracle mallocs its planes per call; above glibc's mmap threshold every call would be an mmap + page faults +
        # munmap under the process-wide mm lock, which serialises the threads (8 cores: 431 -> 971 pairs/s at 320x240).
        # Keep those blocks in the per-thread arenas instead.  M_MMAP_THRESHOLD = -3, M_TRIM_THRESHOLD = -1.
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 1 << 30); libc.mallopt(-1, 1 << 30)
    except (OSError, AttributeError):
        pass
    counts = [0] * threads
    go = threading.Event()
    deadline = [0.0]

    def work(j):
        L, R = frames[j % len(frames)]
        orc.bm_compute(L, R, nthreads=1, **kw)                 # warm: page the buffers in, before the clock starts
        go.wait()
        k = j
        while time.perf_counter() < deadline[0]:
            L, R = frames[k % len(frames)]
            orc.bm_compute(L, R, nthreads=1, **kw)
            counts[j] += 1
            k += threads

    ts = [threading.Thread(target=work, args=(j,)) for j in range(threads)]
    for t in ts: t.start()
    time.sleep(0.05)
    t0 = time.perf_counter()
    deadline[0] = t0 + budget_s
    go.set()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    n = sum(counts)
    return (n / dt if n else 0.0), n


def cpu_baseline(pkg, frames, budget_s=16.0, width=W, height=H, nd=D, block=BLOCK, modes=("frames", "rows", "single")):
    """The oracle (a port, not the reference: bm-sw.cpp needs OpenCV) on the host cores of this box, timed two ways:
    "frames" = one whole frame per thread on every core this process may run on; "rows" = one frame at a time, its rows
    striped over <= 64 threads (how cv::StereoBM itself parallelises).  The better one is reported."""
    from oracle import oracle as orc
    so = orc.build_native()                   # -march=native for THIS box's CPU; the portable build otherwise
    host = os.cpu_count() or 1
    usable = usable_cores()
    kw = dict(numDisparities=nd, blockSize=block)
    L, R = frames[0]
    res = {}
    share = budget_s / (2.2 if "rows" in modes else 1.2)
    if "frames" in modes:
        # the affinity mask can promise more cores than the box's share really runs at once (and a cgroup quota is not always
        # visible): probe a few thread counts and keep the best
        cands = sorted({usable, max(1, usable // 4), min(usable, 16)}, reverse=True)
        for t in cands:
            v, n = cpu_frame_parallel(frames, kw, t, share / len(cands))
            if "frames" not in res or v > res["frames"]["value"]:
                res["frames"] = dict(value=v, threads=t, frames=n)
            res.setdefault("_probe", {})[str(t)] = round(v, 2)
    if "rows" in modes:
        threads = max(1, min(usable, 64))     # the oracle stripes rows over at most 64 threads
        orc.bm_compute(L, R, nthreads=threads, **kw)
        n, t0 = 0, time.perf_counter()
        while True:
            orc.bm_compute(L, R, nthreads=threads, **kw); n += 1
            dt = time.perf_counter() - t0
            if dt > share or n >= 400:
                break
        res["rows"] = dict(value=n / dt, threads=threads, frames=n)
    single = None
    if "single" in modes:
        m, t0 = 0, time.perf_counter()
        while True:
            orc.bm_compute(L, R, nthreads=1, **kw); m += 1
            d1 = time.perf_counter() - t0
            if d1 > budget_s * 0.12 or m >= 100:
                break
        single = m / d1
    probe = res.pop("_probe", None)
    mode = max(res, key=lambda k: res[k]["value"])
    best = res[mode]
    return {"value": round(best["value"], 2), "unit": "stereo-pairs/s", "cores": best["threads"], "threads": best["threads"],
            "host_cores": host, "usable_cores": usable, "kind": "port", "mode": mode, "build": os.path.basename(so),
            "modes": {k: {"pairs_per_s": round(v["value"], 2), "threads": v["threads"], "frames": v["frames"]} for k, v in res.items()},
            "frames_mode_pairs_per_s_by_threads": probe,
            "single_thread_pairs_per_s": round(single, 2) if single else None,
            "sample": "%d x %dx%d d=%d %dx%d full pipeline, oracle/bm_oracle.c (scalar C, not OpenCV's SIMD StereoBM); mode "
                      "'frames' = one frame per thread (best of a few thread counts up to the %d usable cores), 'rows' = one frame at a "
                      "time row-striped over <= 64 threads" % (best["frames"], width, height, nd, block, block, usable)}


def sad_issue_floor(n_pixels, nd, block, clock_ghz, measured_ms):
    """Time the search's quad-SADs alone would take at their measured issue cost, and the share of it the kernel reaches:
    every searched pixel needs nd/4 * ceil(block/4) v_qsad_pk_u16_u8 per lane (each row's SADs are computed once)."""
    per_pixel = nd // 4 * ((block + 3) // 4)
    floor_ms = n_pixels / 64.0 * per_pixel * QSAD_CYCLES / (N_SIMD * clock_ghz * 1e9) * 1e3
    return {"qsad_per_pixel": per_pixel, "cycles_per_qsad_wave_instruction": QSAD_CYCLES, "simds": N_SIMD, "clock_ghz": round(clock_ghz, 3),
            "floor_ms_per_launch": round(floor_ms, 4), "frac_of_floor": round(floor_ms / measured_ms, 4) if measured_ms > 0 else None}


def searched_pixels(width, height, nd, block):
    # columns [nd-1, W) x rows [r, H-r) of a full frame (minDisparity 0, no ROI): what the search kernels visit
    return (width - nd + 1) * (height - 2 * (block // 2))


def time_steps(run, sync, min_steps, min_seconds):
    """>= min_steps calls of run(), continued until min_seconds have passed; returns (steps, seconds)."""
    sync()
    t0 = time.perf_counter()
    n = 0
    while True:
        for _ in range(min_steps if n == 0 else max(1, min_steps // 4)):
            run()
        n += min_steps if n == 0 else max(1, min_steps // 4)
        sync()
        dt = time.perf_counter() - t0
        if dt >= min_seconds:
            return n, dt


def run_config(cfg, pkg, torch, dev, local_rank, clock_ghz, cpu_budget_s, min_steps=20, min_seconds=0.6):
    """One BASELINE configuration on a device-resident batch: throughput, stage times, parity of three frames, CPU beside it."""
    import numpy as np
    cw, ch, cd, cb, B = cfg["W"], cfg["H"], cfg["D"], cfg["w"], cfg["batch"]
    stream = torch.cuda.current_stream().cuda_stream
    dL = torch.empty((B, ch, cw), dtype=torch.uint8, device=dev); dR = torch.empty_like(dL)
    dD = torch.empty((B, ch, cw), dtype=torch.int16, device=dev)
    for i0 in range(0, B, 256):
        n = min(256, B - i0)
        pkg.synth_pairs_device(dL[i0:i0 + n], dR[i0:i0 + n], first_frame=i0, numDisparities=cd, device=local_rank, stream=stream)
    m = pkg.HIPMatcher(numOfDisparities=cd, blockSize=cb, width=cw, height=ch, max_batch=B, device=local_rank)
    mf = dM = dMo = None
    if cfg.get("morph"):
        # config 3's post-filter on an 8UC1 frame of the same size: the valid-disparity mask of each map
        mf = pkg.HIPMorphologicalFilter(cw, ch, 8, max_batch=B, device=local_rank)
        dM = torch.empty((B, ch, cw), dtype=torch.uint8, device=dev); dMo = torch.empty_like(dM)

    def step():
        m.compute_device(dL, dR, dD, stream)
        if mf is not None:
            torch.ne(dD, m.filtered, out=dM.view(torch.bool))          # 0 / 1 bytes ...
            dM.mul_(255)                                               # ... -> the 0 / 255 mask the reference filters (estimator.cpp:43)
            mf.run_device(dM, dMo, stream)

    for _ in range(3):                       # first: model; second: the strip count is measured; third: uses it
        step()
    torch.cuda.synchronize()
    m.set_profiling(True); m.reset_stage_times()
    steps, dt = time_steps(step, torch.cuda.synchronize, min_steps, min_seconds)
    stages = m.stage_times()
    m.set_profiling(False)
    srch = stages["search"]
    s_ms = srch["total_ms"] / max(1, srch["launches"])
    idx = sorted({0, B // 2, B - 1})
    kw = dict(numDisparities=cd, blockSize=cb)
    want = oracle_frames(dL, dR, idx, max(1, min(usable_cores(), 64)), **kw)
    ok = all(np.array_equal(dD[i].cpu().numpy(), wv) for i, wv in zip(idx, want))
    morph_ok = None
    if mf is not None:
        from oracle import oracle as orc
        i = idx[1]
        mask = ((want[1] != m.filtered).astype(np.uint8)) * 255
        morph_ok = bool(np.array_equal(dMo[i].cpu().numpy(), orc.morph_open_close(mask)))
        ok = ok and morph_ok
    nf = min(B, 32)
    frames = [(dL[i].cpu().numpy(), dR[i].cpu().numpy()) for i in range(nf)]
    cpu = cpu_baseline(pkg, frames, budget_s=cpu_budget_s, width=cw, height=ch, nd=cd, block=cb, modes=("frames",)) if cpu_budget_s > 0 else None
    pps = B * steps / dt
    algo = 4 * cw * ch + (2 * cw * ch if mf is not None else 0)
    out = {"key": cfg["key"], "workload": cfg["workload"], "batch": B, "steps": steps, "seconds": round(dt, 3),
           "pairs_per_s": round(pps, 1), "ms_per_frame": round(dt / (B * steps) * 1e3, 5),
           "algorithmic_bytes_per_pair": algo, "frac_of_hbm": round(pps * algo / 1e9 / HBM_PEAK_GBS, 5),
           "search_kernel": m.search_variant, "search_ms_per_launch": round(s_ms, 4),
           "search_frac_of_hbm": round(4 * cw * ch * B / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if s_ms > 0 else None,
           "search_frac_of_qsad_issue_floor": sad_issue_floor(searched_pixels(cw, ch, cd, cb) * B, cd, cb, clock_ghz, s_ms)["frac_of_floor"]
           if m.search_variant.startswith("fast") else None,
           "stage_ms_per_launch": {k: round(v["total_ms"] / max(1, v["launches"]), 4) for k, v in stages.items()},
           "parity_checked_frames": len(idx), "parity_ok": bool(ok),
           "cpu_pairs_per_s": cpu["value"] if cpu else None, "cpu_threads": cpu["threads"] if cpu else None}
    if morph_ok is not None:
        out["morph_parity_ok"] = morph_ok
    m.close()
    if mf is not None:
        mf.close()
    return out


def run_sgm_config(pkg, torch, dev, local_rank, batch=16, min_steps=3, min_seconds=0.5, cpu=True):
    """BASELINE config 5: the SWSemiGlobalMatcher counterpart (cv::StereoSGBM restated) at 1280x720 d=128 blockSize 5 on a
    device-resident batch: MODE_HH (the "8-path" of the config) and MODE_SGBM (what sgbm-sw.cpp:15 creates), one frame of each
    checked against the oracle (integer algorithm: tolerance 0), the oracle timed beside it (one thread, one frame)."""
    import numpy as np
    from oracle import oracle as orc
    cd = 128
    dL = torch.empty((batch, H, W), dtype=torch.uint8, device=dev); dR = torch.empty_like(dL)
    dD = torch.empty((batch, H, W), dtype=torch.int16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    pkg.synth_pairs_device(dL, dR, first_frame=0, numDisparities=cd, device=local_rank, stream=st)
    out = {"key": "config5", "workload": "1280x720 d=128 StereoSGBM blockSize 5 (sgbm-sw.cpp:12-37), %d pairs per call" % batch, "batch": batch,
           "algorithmic_bytes_per_pair": 4 * W * H, "tolerance": 0}
    L0, R0 = dL[batch - 1].cpu().numpy(), dR[batch - 1].cpu().numpy()
    ok = True
    for paths, name in ((8, "mode_hh_8_paths"), (5, "mode_sgbm_5_paths")):
        sg = pkg.HIPSemiGlobalMatcher(numOfDisparities=cd, width=W, height=H, max_batch=batch, paths=paths, device=local_rank)
        def step():
            sg.compute_device(dL, dR, dD, st)
        for _ in range(2): step()
        torch.cuda.synchronize()
        steps, dt = time_steps(step, torch.cuda.synchronize, min_steps, min_seconds)
        got = dD[batch - 1].cpu().numpy()
        rec = {"pairs_per_s": round(batch * steps / dt, 1), "ms_per_pair": round(dt / (batch * steps) * 1e3, 4), "steps": steps}
        sweeps, gave_up = sg.pass_stats()
        rec["row_synchronous_sweeps_per_call"] = sweeps // (2 + steps)
        rec["sweep_gave_up"] = bool(gave_up)
        if cpu:
            t0 = time.perf_counter(); want = orc.sgm_compute(L0, R0, numDisparities=cd, paths=paths); tc = time.perf_counter() - t0
            rec["parity_ok"] = bool(np.array_equal(got, want)); rec["cpu_pairs_per_s_1_thread"] = round(1.0 / tc, 3)
            ok = ok and rec["parity_ok"]
        sg.close()
        out[name] = rec
    out["pairs_per_s"] = out["mode_hh_8_paths"]["pairs_per_s"]
    out["ms_per_frame"] = out["mode_hh_8_paths"]["ms_per_pair"]
    out["frac_of_hbm"] = round(out["pairs_per_s"] * 4 * W * H / 1e9 / HBM_PEAK_GBS, 6)
    out["parity_ok"] = bool(ok) if cpu else None
    out["parity_checked_frames"] = 1 if cpu else 0
    return out


def single_frame_latency(pkg, torch, dL, dR, dD, want0, local_rank, stream):
    """"ms/frame" for a caller that hands over ONE frame at a time (the reference's loop, estimator.cpp:56): host to host
    through rtdm_bm_compute -- pageable frames in, pageable map out, PCIe inclusive -- and the device-resident call alone.
    Outside the timed region; reported beside the throughput figure, never instead of it."""
    import numpy as np
    L1, R1 = dL[0].cpu().numpy(), dR[0].cpu().numpy()
    out1 = np.empty((H, W), np.int16)
    m1 = pkg.HIPMatcher(numOfDisparities=D, blockSize=BLOCK, width=W, height=H, max_batch=1, device=local_rank)
    for _ in range(5): m1.compute(L1, R1, out1)
    t1 = time.perf_counter()
    for _ in range(50): m1.compute(L1, R1, out1)
    h2h = (time.perf_counter() - t1) / 50
    for _ in range(5): m1.compute_device(dL[:1], dR[:1], dD[:1], stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(50):
        m1.compute_device(dL[:1], dR[:1], dD[:1], stream); torch.cuda.synchronize()
    dev1 = (time.perf_counter() - t1) / 50
    pin = [torch.from_numpy(a).pin_memory() for a in (L1, R1, np.zeros((H, W), np.int16))]
    pl1, pr1, po1 = [t.numpy() for t in pin]
    for _ in range(5): m1.compute(pl1, pr1, po1)
    t1 = time.perf_counter()
    for _ in range(50): m1.compute(pl1, pr1, po1)
    h2hp = (time.perf_counter() - t1) / 50
    m1.close()
    single = {"host_to_host_ms": round(h2h * 1e3, 4), "host_to_host_page_locked_ms": round(h2hp * 1e3, 4),
              "device_resident_ms": round(dev1 * 1e3, 4),
              "same_as_batched": bool(np.array_equal(out1, want0) and np.array_equal(po1, want0)),
              "note": "one 1280x720 pair per call; host_to_host = rtdm_bm_compute, PCIe inclusive, from / to pageable "
                      "memory (gathered through a staging area) or page-locked memory (DMA straight from / to the caller's planes)"}
    # the call the reference really makes (estimator.cpp:33-36,54-56; main.cpp:80-85,134-135): 934x404 views at the crop
    # origin of 1280-pitch planes, ROI1 set before every compute, d=192 13x13
    from oracle import oracle as orc
    cx, cy, cw, ch = REF_CROP
    Lv, Rv = L1[cy:cy + ch, cx:cx + cw], R1[cy:cy + ch, cx:cx + cw]
    disp = np.empty((H, W), np.int16)                      # the caller's reused left_disp (estimator.h:100), as a view too
    Ov = disp[cy:cy + ch, cx:cx + cw]
    mr = pkg.HIPMatcher(numOfDisparities=192, blockSize=13, width=cw, height=ch, max_batch=1, device=local_rank)
    res = {}
    for name, roi in (("roi1_set", REF_ROI1), ("no_roi", None)):
        mr.setROI1(roi if roi else (0, 0, 0, 0))
        for _ in range(5): mr.compute(Lv, Rv, Ov)
        t1 = time.perf_counter()
        for _ in range(40):
            if roi: mr.setROI1(roi)
            mr.compute(Lv, Rv, Ov)
        res[name] = (time.perf_counter() - t1) / 40
        wantv = orc.bm_compute(np.ascontiguousarray(Lv), np.ascontiguousarray(Rv), nthreads=min(usable_cores(), 64),
                               numDisparities=192, blockSize=13, roi1=roi)
        res[name + "_ok"] = bool(np.array_equal(Ov, wantv))
    single["reference_call"] = {"workload": "934x404 views at (192,177) of 1280-pitch pageable planes, d=192 13x13, setROI1 + compute per frame "
                                            "(estimator.cpp:33-36,54-56; main.cpp:80-85,134-135)",
                                "host_to_host_ms_roi1_set": round(res["roi1_set"] * 1e3, 4), "host_to_host_ms_no_roi": round(res["no_roi"] * 1e3, 4),
                                "roi1": list(REF_ROI1), "search_kernel": mr.search_variant,
                                "parity_ok": bool(res["roi1_set_ok"] and res["no_roi_ok"])}
    mr.close()
    return single


def rccl_stream(args, pkg, torch, dist, rank, local_rank, world, backend):
    """BASELINE config 4: a root-sourced stream of independent pairs, block-partitioned over the ranks."""
    sh = importlib.import_module("rt-depth-map_amd.sharding")
    N = args.rccl_stream
    dev = torch.device("cuda", local_rank)
    cap = max(sh.shard_sizes(N, world))
    chunk = max(1, min(args.chunk, cap))
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=BLOCK, width=W, height=H, max_batch=chunk, device=local_rank)
    left = right = None
    if rank == 0:
        st0 = torch.cuda.current_stream().cuda_stream
        left = torch.empty((N, H, W), dtype=torch.uint8, device=dev); right = torch.empty_like(left)
        for i0 in range(0, N, 128):
            n = min(128, N - i0)
            pkg.synth_pairs_device(left[i0:i0 + n], right[i0:i0 + n], first_frame=i0, numDisparities=D, device=local_rank, stream=st0)

    def compute_into(L, R, out):
        m.compute_device(L, R, out, torch.cuda.current_stream().cuda_stream)

    def run(timings=None):
        return sh.scatter_compute_gather(dist, left, right, N, (H, W), None, dev, chunk=chunk, compute_into=compute_into, timings=timings)

    def sync_all():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        out = run()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    sync_all()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())
    # one more pass with events around every scatter / compute / gather (outside the timed region: the scatter's end event makes
    # the side stream wait for it, which the timed passes do not): what overlaps on the first real xGMI run shows up here
    tm = sh.new_timings()
    t1 = time.perf_counter()
    out = run(tm)
    sync_all()
    wall_ms = (time.perf_counter() - t1) * 1e3
    phases = sh.phase_ms(tm)
    phases["wall_ms"] = round(wall_ms, 3)
    phases["note"] = ("sums of per-chunk HIP-event intervals on this rank (rank 0): scatter and gather on the side stream, compute on the "
                      "caller's stream; scatter + compute + gather > wall means they overlapped")
    rc = 0
    oracle_ready(rank, dist)
    if rank == 0:
        idx = sorted({0, N // 2, N - 1})
        want = oracle_frames(left, right, idx, min(usable_cores(), 64))
        import numpy as np
        ok = all(np.array_equal(out[i].cpu().numpy(), w) for i, w in zip(idx, want))
        # the direct call on the same frames (no collectives) must give the same bytes
        chk = torch.empty((len(idx), H, W), dtype=torch.int16, device=dev)
        m.compute_device(left[idx].contiguous(), right[idx].contiguous(), chk, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        same = bool(torch.equal(chk, out[idx]))
        print(json.dumps({"metric": "root-sourced stereo-pairs/sec (BASELINE config 4), 1280x720 d=64 9x9", "value": round(N * args.steps / el, 1),
                          "unit": "stereo-pairs/s", "n_gpus": world, "ranks_seen": dist.get_world_size(), "steps": args.steps, "warmup": args.warmup, "frames": N,
                          "chunk_frames": chunk, "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
                          "data": "synthetic", "dtype": "u8", "backend": backend,
                          "collectives": "scatter + gather per chunk (torch.distributed, backend nccl = RCCL), double-buffered beside the compute",
                          "phase_ms_per_pass": phases,
                          "valid_fraction": round(float((out != m.filtered).float().mean().item()), 4),
                          "parity_checked_frames": len(idx), "parity_ok": bool(ok), "equals_direct_call": same}))
        rc = 0 if (ok and same) else 4
    dist.barrier()
    dist.destroy_process_group()
    if rc:
        sys.exit(rc)


def load_pmc(m, frames_per_launch):
    """HBM bytes of the dominant kernel from the rocprofv3 PMC passes (tools/pmc_traffic.sh; FETCH_SIZE x2 + WRITE_SIZE, per
    the gfx950 correction); collected offline because --pmc cannot run inside this process.  The file records the hash
    of the device sources it was measured on: if the kernels have changed since, the figures are withheld."""
    traffic = compute_view = None
    state, used = "absent", None
    for name in PMC_CANDIDATES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        used = path
        try:
            pm = json.load(open(path))
            if pm.get("kernel_source_sha") != kernel_source_sha():
                state = "stale"
                break
            if not m.search_variant.startswith("fast"):
                break
            state = "current"
            ks = pm["kernels"]
            main_k = [k for k in ks if k.startswith("k_search_") and "border" not in k and "generic" not in k]
            k = ks[max(main_k, key=lambda n: ks[n].get("SQ_INSTS_VALU", 0))]
            kb = [ks[n] for n in ks if n.startswith("k_search_border")]          # runs beside the tile kernel on a side stream
            per_pair = k["hbm_bytes_per_pair"] + sum(b.get("hbm_bytes_per_pair", 0) for b in kb)
            traffic = int(per_pair * frames_per_launch)
            # the kernel is integer-VALU bound, not HBM bound: what the SQ counters of the same
            # offline rocprofv3 run say about it
            compute_view = {"bound": "valu", "valu_busy_raw": k.get("valu_busy_raw", k.get("valu_busy_frac_of_simd_cycles")),
                            "valu_busy_note": k.get("valu_busy_note"),
                            "valu_insts_per_pixel": k.get("valu_insts_per_pixel"), "wave_cycle_split": k.get("wave_cycle_split"),
                            "pairs_per_launch_when_measured": pm.get("pairs_per_launch"),
                            "source": "rocprofv3 --pmc SQ_* pass, " + os.path.relpath(path, ROOT)}
        except Exception:
            traffic = None
        break
    return traffic, compute_view, state, (os.path.relpath(used, ROOT) if used else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024,
                    help="pairs per step per GPU (BASELINE config 4's stream length; 256 is 4 percent slower: tail effects of the search grid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-frame", action="store_true",
                    help="skip the single-frame latency measurement (profiling runs: its launches would mix into per-kernel averages)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations")
    ap.add_argument("--headline-only", action="store_true",
                    help="the K timed steps and nothing else (rocprofv3 runs: no sustained region, configs, single frames or CPU baseline)")
    ap.add_argument("--rccl-stream", type=int, default=0, metavar="FRAMES",
                    help="BASELINE config 4 instead of the headline: rank 0 owns FRAMES pairs, scatter -> compute -> gather "
                         "over torch.distributed (RCCL); reports root-sourced pairs/s")
    ap.add_argument("--chunk", type=int, default=128, help="--rccl-stream: frames per scatter/gather per rank (128: 60 k pairs/s at world size 1, 64: 55 k)")
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu_baseline = args.no_single_frame = args.no_configs = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)               # does not return

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    # one process per GPU; RTDM_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals
    backend = os.environ.get("RTDM_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.rccl_stream:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # a checkout without built libraries: build them (what __graft_entry__.build() does) -- rank 0 only, the others wait.
    # This is building the product, not falling back: without the HIP library nothing below can run.
    if not os.path.exists(os.path.join(ROOT, "rt-depth-map_amd", "lib", "librtdm_hip.so")):
        if rank == 0:
            subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "rt-depth-map_amd")])
        if dist is not None:
            dist.barrier()
    pkg = importlib.import_module("rt-depth-map_amd")
    if args.rccl_stream:
        return rccl_stream(args, pkg, torch, dist, rank, local_rank, world, backend)
    B = args.batch
    dev = torch.device("cuda", local_rank)
    clock_ghz = (getattr(torch.cuda.get_device_properties(local_rank), "clock_rate", 0) or 2400000) / 1e6
    dL = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    dR = torch.empty_like(dL)
    dD = torch.empty((B, H, W), dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    # rank r owns frames [r*B, (r+1)*B) of the synthetic stream; no communication
    pkg.synth_pairs_device(dL, dR, first_frame=rank * B, numDisparities=D, device=local_rank, stream=stream)
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=BLOCK, width=W, height=H, max_batch=B, device=local_rank)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def allmax(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        m.compute_device(dL, dR, dD, stream)
    sync_all()
    m.set_profiling(True)
    m.reset_stage_times()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.compute_device(dL, dR, dD, stream)
    sync_all()
    elapsed = allmax(time.perf_counter() - t0)
    stages = m.stage_times()
    m.set_profiling(False)

    # the same step, repeated until the GPU has been busy for >= 2 s in one stretch: long enough for a sampler outside this
    # process to see it.  Its own figure ("sustained"); `value` stays the K-step region above.
    sustained = None
    if not args.headline_only:
        n_s = max(args.steps, int(math.ceil(2.0 / max(elapsed / max(1, args.steps), 1e-6))))   # from the all-reduced time: same on every rank
        t0 = time.perf_counter()
        for _ in range(n_s):
            m.compute_device(dL, dR, dD, stream)
        sync_all()
        el_s = allmax(time.perf_counter() - t0)
        sustained = {"steps": n_s, "seconds": round(el_s, 3), "pairs_per_s": round(world * B * n_s / el_s, 1)}

    # parity self-check of the timed path (this batch size, autotuned strips, side-stream border kernel): three frames of
    # the last step's output against the CPU oracle, on every rank's own shard
    import numpy as np
    oracle_ready(rank, dist)
    idx = sorted({0, B // 2, B - 1})
    want = oracle_frames(dL, dR, idx, max(1, min(usable_cores() // max(1, min(world, torch.cuda.device_count())), 64)))
    bad = [i for i, w in zip(idx, want) if not np.array_equal(dD[i].cpu().numpy(), w)]
    parity_ok = not bad
    if dist is not None:
        flag = torch.tensor([0 if parity_ok else 1], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        parity_ok = int(flag.item()) == 0

    single = None
    if rank == 0 and world == 1 and not args.no_single_frame:
        single = single_frame_latency(pkg, torch, dL, dR, dD, want[0], local_rank, stream)
    valid_frac = float((dD != m.filtered).float().mean().item())
    total_pairs = world * B * args.steps
    value = total_pairs / elapsed
    srch = stages["search"]
    avg_ms = srch["total_ms"] / max(1, srch["launches"])
    frames_per_launch = srch["frames"] / max(1, srch["launches"])
    achieved = ALGO_BYTES_PER_PAIR * frames_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, compute_view, pmc_state, pmc_file = load_pmc(m, frames_per_launch)
    floor = sad_issue_floor(searched_pixels(W, H, D, BLOCK) * frames_per_launch, D, BLOCK, clock_ghz, avg_ms) \
        if m.search_variant.startswith("fast") else None
    variant = m.search_variant
    cpu_frames = [(dL[i].cpu().numpy(), dR[i].cpu().numpy()) for i in range(min(B, 32))] if rank == 0 else None
    filtered = m.filtered

    # the other BASELINE configurations (rank 0; at world > 1 the other ranks wait at the final barrier)
    configs = None
    if rank == 0 and not args.no_configs:
        m.close(); del dL, dR, dD
        torch.cuda.empty_cache()
        cpu_each = 0.0 if args.no_cpu_baseline else (2.5 if world == 1 else 1.5)
        configs = [run_config(c, pkg, torch, dev, local_rank, clock_ghz, cpu_each) for c in CONFIGS]
        configs.append(run_sgm_config(pkg, torch, dev, local_rank, cpu=cpu_each > 0))
        if not all(c["parity_ok"] is not False for c in configs):
            parity_ok = False
            bad = bad + [c["key"] for c in configs if not c["parity_ok"]]

    out = {
        "metric": "stereo-pairs/sec, 1280x720 d=64 9x9 SAD", "value": round(value, 1), "unit": "stereo-pairs/s",
        "n_gpus": world, "ranks_seen": dist.get_world_size() if dist is not None else 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "ms_per_frame": round(elapsed / (B * args.steps) * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "1280x720 rectified pairs, numDisparities=64, blockSize=9, full StereoBM pipeline "
                               "(x-Sobel prefilter, SAD search, uniqueness/texture, left-right check, speckle filter)",
                   "pairs_per_step_per_gpu": B, "search_kernel": variant,
                   "parallelism": "whole-frame sharding, %d rank(s), no data-path collective" % world,
                   "valid_fraction": round(valid_frac, 4)},
        "parity_checked_frames": len(idx) * world, "parity_ok": bool(parity_ok),
        "timed_region_s": round(elapsed, 4), "sustained": sustained,
        "roofline": {"bound": "hbm", "kernel": "SAD search (%s)" % variant,
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_state": pmc_state,
                     "traffic_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE(x2)/WRITE_SIZE, %s; null + "
                                     "\"stale\" when that file was measured on other kernel sources" % pmc_file,
                     "avg_launch_ms": round(avg_ms, 4), "pairs_per_launch": frames_per_launch,
                     "algorithmic_bytes_per_pair": ALGO_BYTES_PER_PAIR,
                     "qsad_issue_floor": floor,
                     "note": "the kernel is integer-VALU bound (DESIGN.md section 4): frac is its share of the HBM peak as north_star asks; "
                             "qsad_issue_floor.frac_of_floor is its share of its own quad-SAD issue roofline, the number that can still move"},
        "stage_ms_per_launch": {k: round(v["total_ms"] / max(1, v["launches"]), 4) for k, v in stages.items()},
        "compute_view": compute_view,
        "configs": configs,
        "single_frame": single,
    }
    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, cpu_frames, budget_s=16.0 if world == 1 else 8.0)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not parity_ok:
        sys.stderr.write("bench.py: PARITY MISMATCH against the oracle on %s of rank %d\n" % (bad, rank))
        sys.exit(4)
    del filtered


if __name__ == "__main__":
    main()

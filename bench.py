#!/usr/bin/env python3
"""bench.py -- stereo-pairs/sec of the BlockMatcher hot path on MI355X.

Workload (BASELINE.json metric): 1280x720 rectified pairs, numDisparities 64, 9x9 SAD, every other
StereoBM knob at the reference's literals (main.cpp:134-135: cap 31, texture 10, uniqueness 10,
disp12MaxDiff 1, speckle 100/32), i.e. the whole cv::StereoBM::compute pipeline that
SWMatcherKonolige::compute (bm-sw.cpp:33-38) runs.  A "step" is one rtdm_bm_compute_device call
over a batch of --batch synthetic pairs (default 1024, the stream length of BASELINE config 4) that are
already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

--gpus N > 1 without a torch.distributed environment: this process touches no GPU; it starts N
ranks of itself (python -m torch.distributed.run, one process per GPU, RCCL), relays rank 0's JSON
line and exits with the ranks' status.  Under torchrun (WORLD_SIZE set) it is one of the ranks.

Multi-GPU: frames are independent (estimator.cpp:18-82 carries nothing across iterations), so each
rank synthesises and processes its own shard of the stream (weak scaling, no data-path collective);
timing is barrier + synchronize on both sides and the MAX over ranks.  --rccl-stream FRAMES is
BASELINE config 4 instead: rank 0 owns the frames, scatter -> compute -> gather over RCCL.

After the timed region three frames of the last step's output are compared with the CPU oracle
("parity_ok"); a mismatch exits non-zero.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, D, BLOCK = 1280, 720, 64, 9
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
ALGO_BYTES_PER_PAIR = 4 * W * H  # read L + read R (u8) + write int16 disparity
PMC_JSON = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")


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


def oracle_frames(dL, dR, idx, threads):
    """The CPU oracle's disparity for frames idx of the device batch (the checker, never the thing measured)."""
    from oracle import oracle as orc
    return [orc.bm_compute(dL[i].cpu().numpy(), dR[i].cpu().numpy(), nthreads=threads, numDisparities=D, blockSize=BLOCK)
            for i in idx]


def cpu_baseline(pkg, budget_s=12.0):
    """The oracle (a port, not the reference: bm-sw.cpp needs OpenCV) on every host core of this box."""
    from oracle import oracle as orc
    so = orc.build_native()                   # -march=native for THIS box's CPU; the portable build otherwise
    cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = cores
    threads = max(1, min(usable, 64))         # the oracle stripes rows over at most 64 threads
    L, R = pkg.synth.make_pair(pkg.synth.STREAM_SEED, W, H, D)
    kw = dict(numDisparities=D, blockSize=BLOCK)
    orc.bm_compute(L, R, nthreads=threads, **kw)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        orc.bm_compute(L, R, nthreads=threads, **kw)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s * 0.6 or n >= 400:
            break
    multi = n / dt
    m, t0 = 0, time.perf_counter()
    while True:
        orc.bm_compute(L, R, nthreads=1, **kw)
        m += 1
        d1 = time.perf_counter() - t0
        if d1 > budget_s * 0.4 or m >= 100:
            break
    return {"value": round(multi, 2), "unit": "stereo-pairs/s", "cores": cores, "threads": threads, "kind": "port",
            "build": os.path.basename(so),
            "sample": "%d x 1280x720 d=64 9x9 full pipeline, oracle/bm_oracle.c (scalar C, not OpenCV's SIMD StereoBM) row-striped "
                      "over %d threads of %d host cores (single thread: %.2f pairs/s over %d frames)" % (n, threads, cores, m / d1, m)}


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

    def run():
        return sh.scatter_compute_gather(dist, left, right, N, (H, W), None, dev, chunk=chunk, compute_into=compute_into)

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
    rc = 0
    oracle_ready(rank, dist)
    if rank == 0:
        idx = sorted({0, N // 2, N - 1})
        want = oracle_frames(left, right, idx, min(os.cpu_count() or 1, 64))
        import numpy as np
        ok = all(np.array_equal(out[i].cpu().numpy(), w) for i, w in zip(idx, want))
        # the direct call on the same frames (no collectives) must give the same bytes
        chk = torch.empty((len(idx), H, W), dtype=torch.int16, device=dev)
        m.compute_device(left[idx].contiguous(), right[idx].contiguous(), chk, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        same = bool(torch.equal(chk, out[idx]))
        print(json.dumps({"metric": "root-sourced stereo-pairs/sec (BASELINE config 4), 1280x720 d=64 9x9", "value": round(N * args.steps / el, 1),
                          "unit": "stereo-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "frames": N,
                          "chunk_frames": chunk, "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
                          "data": "synthetic", "dtype": "u8", "backend": backend,
                          "collectives": "scatter + gather per chunk (torch.distributed, backend nccl = RCCL), double-buffered beside the compute",
                          "valid_fraction": round(float((out != m.filtered).float().mean().item()), 4),
                          "parity_checked_frames": len(idx), "parity_ok": bool(ok), "equals_direct_call": same}))
        rc = 0 if (ok and same) else 4
    dist.barrier()
    dist.destroy_process_group()
    if rc:
        sys.exit(rc)


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
    ap.add_argument("--rccl-stream", type=int, default=0, metavar="FRAMES",
                    help="BASELINE config 4 instead of the headline: rank 0 owns FRAMES pairs, scatter -> compute -> gather "
                         "over torch.distributed (RCCL); reports root-sourced pairs/s")
    ap.add_argument("--chunk", type=int, default=32, help="--rccl-stream: frames per scatter/gather per rank")
    args = ap.parse_args()

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

    for _ in range(args.warmup):
        m.compute_device(dL, dR, dD, stream)
    sync_all()
    m.set_profiling(True)
    m.reset_stage_times()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.compute_device(dL, dR, dD, stream)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stages = m.stage_times()
    m.set_profiling(False)

    # parity self-check of the timed path (this batch size, autotuned strips, side-stream border kernel): three frames of
    # the last step's output against the CPU oracle, on every rank's own shard
    import numpy as np
    oracle_ready(rank, dist)
    idx = sorted({0, B // 2, B - 1})
    want = oracle_frames(dL, dR, idx, max(1, min((os.cpu_count() or 1) // max(1, min(world, torch.cuda.device_count())), 64)))
    bad = [i for i, w in zip(idx, want) if not np.array_equal(dD[i].cpu().numpy(), w)]
    parity_ok = not bad
    if dist is not None:
        flag = torch.tensor([0 if parity_ok else 1], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        parity_ok = int(flag.item()) == 0

    # "ms/frame" for a caller that hands over ONE frame at a time (the reference's loop, estimator.cpp:56): host to host
    # through rtdm_bm_compute -- pageable frames in, pageable map out, PCIe inclusive -- and the device-resident call alone.
    # Outside the timed region; reported beside the throughput figure, never instead of it.
    single = None
    if rank == 0 and world == 1 and not args.no_single_frame:
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
        single = {"host_to_host_ms": round(h2h * 1e3, 4), "host_to_host_page_locked_ms": round(h2hp * 1e3, 4),
                  "device_resident_ms": round(dev1 * 1e3, 4),
                  "same_as_batched": bool(np.array_equal(out1, want[0]) and np.array_equal(po1, want[0])),
                  "note": "one 1280x720 pair per call; host_to_host = rtdm_bm_compute, PCIe inclusive, from / to pageable "
                          "memory (gathered through a staging area) or page-locked memory (DMA straight from / to the caller's planes)"}
        m1.close()
    valid_frac = float((dD != m.filtered).float().mean().item())
    total_pairs = world * B * args.steps
    value = total_pairs / elapsed
    srch = stages["search"]
    avg_ms = srch["total_ms"] / max(1, srch["launches"])
    frames_per_launch = srch["frames"] / max(1, srch["launches"])
    achieved = ALGO_BYTES_PER_PAIR * frames_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes of the dominant kernel from the rocprofv3 PMC passes (tools/pmc_traffic.sh; FETCH_SIZE x2 + WRITE_SIZE, per
    # the gfx950 correction); collected offline because --pmc cannot run inside this process.  The file records the hash
    # of the device sources it was measured on: if the kernels have changed since, the figures are withheld.
    traffic = None
    compute_view = None
    pmc_state = "absent"
    try:
        pm = json.load(open(PMC_JSON))
        if pm.get("kernel_source_sha") != kernel_source_sha():
            pmc_state = "stale"
        elif m.search_variant.startswith("fast"):
            pmc_state = "current"
            ks = pm["kernels"]
            main_k = [k for k in ks if k.startswith("k_search_") and "border" not in k and "generic" not in k]
            k = ks[max(main_k, key=lambda n: ks[n].get("SQ_INSTS_VALU", 0))]
            kb = [ks[n] for n in ks if n.startswith("k_search_border")]          # runs beside the tile kernel on a side stream
            per_pair = k["hbm_bytes_per_pair"] + sum(b.get("hbm_bytes_per_pair", 0) for b in kb)
            traffic = int(per_pair * frames_per_launch)
            # the kernel is integer-VALU bound, not HBM bound: what the SQ counters of the same
            # offline rocprofv3 run say about it
            compute_view = {"bound": "valu", "valu_busy_frac_of_simd_cycles": k.get("valu_busy_frac_of_simd_cycles"),
                            "valu_insts_per_pixel": k.get("valu_insts_per_pixel"), "wave_cycle_split": k.get("wave_cycle_split"),
                            "pairs_per_launch_when_measured": pm.get("pairs_per_launch"),
                            "source": "rocprofv3 --pmc SQ_* pass, " + os.path.relpath(PMC_JSON, ROOT)}
    except Exception:
        traffic = None
    out = {
        "metric": "stereo-pairs/sec, 1280x720 d=64 9x9 SAD", "value": round(value, 1), "unit": "stereo-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "ms_per_frame": round(elapsed / (B * args.steps) * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "1280x720 rectified pairs, numDisparities=64, blockSize=9, full StereoBM pipeline "
                               "(x-Sobel prefilter, SAD search, uniqueness/texture, left-right check, speckle filter)",
                   "pairs_per_step_per_gpu": B, "search_kernel": m.search_variant,
                   "parallelism": "whole-frame sharding, %d rank(s), no data-path collective" % world,
                   "valid_fraction": round(valid_frac, 4)},
        "parity_checked_frames": len(idx) * world, "parity_ok": bool(parity_ok),
        "roofline": {"bound": "hbm", "kernel": "SAD search (%s)" % m.search_variant,
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_state": pmc_state,
                     "traffic_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE(x2)/WRITE_SIZE, %s; null + "
                                     "\"stale\" when that file was measured on other kernel sources" % os.path.relpath(PMC_JSON, ROOT),
                     "avg_launch_ms": round(avg_ms, 4), "pairs_per_launch": frames_per_launch,
                     "algorithmic_bytes_per_pair": ALGO_BYTES_PER_PAIR},
        "stage_ms_per_launch": {k: round(v["total_ms"] / max(1, v["launches"]), 4) for k, v in stages.items()},
        "compute_view": compute_view,
        "single_frame": single,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not parity_ok:
        sys.stderr.write("bench.py: PARITY MISMATCH against the oracle on frames %s of rank %d\n" % (bad, rank))
        sys.exit(4)


if __name__ == "__main__":
    main()

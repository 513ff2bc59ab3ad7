#!/usr/bin/env python3
"""bench.py -- stereo-pairs/sec of the BlockMatcher hot path on MI355X.

Workload (BASELINE.json metric): 1280x720 rectified pairs, numDisparities 64, 9x9 SAD, every other
StereoBM knob at the reference's literals (main.cpp:134-135: cap 31, texture 10, uniqueness 10,
disp12MaxDiff 1, speckle 100/32), i.e. the whole cv::StereoBM::compute pipeline that
SWMatcherKonolige::compute (bm-sw.cpp:33-38) runs.  A "step" is one rtdm_bm_compute_device call
over a batch of --batch synthetic pairs that are already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: frames are independent, so each rank synthesises and processes its own shard of the
stream (weak scaling, no data-path collective); timing is barrier + synchronize on both sides and
the MAX over ranks.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, D, BLOCK = 1280, 720, 64, 9
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
ALGO_BYTES_PER_PAIR = 4 * W * H  # read L + read R (u8) + write int16 disparity


def cpu_baseline(pkg, budget_s=12.0):
    """The oracle (a port, not the reference: bm-sw.cpp needs OpenCV) on this box's host cores."""
    from oracle import oracle as orc
    import numpy as np
    orc.build()
    cores = min(os.cpu_count() or 1, 16)
    L, R = pkg.synth.make_pair(pkg.synth.STREAM_SEED, W, H, D)
    kw = dict(numDisparities=D, blockSize=BLOCK)
    orc.bm_compute(L, R, nthreads=cores, **kw)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        orc.bm_compute(L, R, nthreads=cores, **kw)
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
    return {"value": round(multi, 2), "unit": "stereo-pairs/s", "cores": cores, "kind": "port",
            "sample": "%d x 1280x720 d=64 9x9 full pipeline, oracle/bm_oracle.c row-striped over %d threads "
                      "(single thread: %.2f pairs/s over %d frames)" % (n, cores, m / d1, m)}


def rccl_stream(args, pkg, torch, dist, rank, local_rank, world):
    """BASELINE config 4: a root-sourced stream of independent pairs, block-partitioned over the ranks."""
    sh = importlib.import_module("rt-depth-map_amd.sharding")
    N = args.rccl_stream
    dev = torch.device("cuda", local_rank)
    st = torch.cuda.current_stream().cuda_stream
    cap = max(sh.shard_sizes(N, world))
    m = pkg.HIPMatcher(numOfDisparities=D, blockSize=BLOCK, width=W, height=H, max_batch=min(cap, 128), device=local_rank)
    left = right = None
    if rank == 0:
        left = torch.empty((N, H, W), dtype=torch.uint8, device=dev); right = torch.empty_like(left)
        for i0 in range(0, N, 128):
            n = min(128, N - i0)
            pkg.synth_pairs_device(left[i0:i0 + n], right[i0:i0 + n], first_frame=i0, numDisparities=D, device=local_rank, stream=st)

    def compute(L, R):
        out = torch.empty(L.shape, dtype=torch.int16, device=dev)
        m.compute_device(L.contiguous(), R.contiguous(), out, st)
        return out

    class _Solo:      # world == 1: same code path without a process group
        @staticmethod
        def get_world_size(): return 1
        @staticmethod
        def get_rank(): return 0
        @staticmethod
        def scatter(t, lst, src=0): t.copy_(lst[0])
        @staticmethod
        def gather(t, lst, dst=0): lst[0].copy_(t)
    d = dist if dist is not None else _Solo
    for _ in range(max(1, args.warmup)):
        sh.scatter_compute_gather(d, left, right, N, (H, W), compute, dev, chunk=64)
    torch.cuda.synchronize()
    if dist is not None: dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = sh.scatter_compute_gather(d, left, right, N, (H, W), compute, dev, chunk=64)
    torch.cuda.synchronize()
    if dist is not None: dist.barrier()
    el = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "root-sourced stereo-pairs/sec (BASELINE config 4), 1280x720 d=64 9x9", "value": round(N * args.steps / el, 1),
                          "unit": "stereo-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "frames": N,
                          "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
                          "data": "synthetic", "dtype": "u8", "collectives": "scatter + gather (torch.distributed, backend nccl = RCCL)",
                          "valid_fraction": round(float((out != m.filtered).float().mean().item()), 4)}))
    if dist is not None: dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="pairs per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rccl-stream", type=int, default=0, metavar="FRAMES",
                    help="BASELINE config 4 instead of the headline: rank 0 owns FRAMES pairs, scatter -> compute -> gather "
                         "over torch.distributed (RCCL); reports root-sourced pairs/s")
    args = ap.parse_args()

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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # a checkout without built libraries: build them (what __graft_entry__.build() does) -- rank 0 only, the others wait.
    # This is building the product, not falling back: without the HIP library nothing below can run.
    if not os.path.exists(os.path.join(ROOT, "rt-depth-map_amd", "lib", "librtdm_hip.so")):
        if rank == 0:
            import subprocess
            subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "rt-depth-map_amd")])
        if dist is not None:
            dist.barrier()
    pkg = importlib.import_module("rt-depth-map_amd")
    if args.rccl_stream:
        return rccl_stream(args, pkg, torch, dist, rank, local_rank, world)
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

    valid_frac = float((dD != m.filtered).float().mean().item())
    total_pairs = world * B * args.steps
    value = total_pairs / elapsed
    srch = stages["search"]
    avg_ms = srch["total_ms"] / max(1, srch["launches"])
    frames_per_launch = srch["frames"] / max(1, srch["launches"])
    achieved = ALGO_BYTES_PER_PAIR * frames_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes of the dominant kernel from the rocprofv3 PMC passes (tools/pmc_traffic.sh; FETCH_SIZE x2 +
    # WRITE_SIZE, per the gfx950 correction); collected offline because --pmc cannot run inside this process
    traffic = None
    compute_view = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        if m.search_variant == "fast_qsad":
            k = pm["kernels"]["k_search_fast<64,3>"]
            kb = pm["kernels"].get("k_search_border<1>", {})               # runs beside the tile kernel on a side stream
            traffic = int((k["hbm_bytes_per_pair"] + kb.get("hbm_bytes_per_pair", 0)) * frames_per_launch)
            # the kernel is integer-VALU bound, not HBM bound: what the SQ counters of the same
            # offline rocprofv3 run say about it
            compute_view = {"bound": "valu", "valu_busy_frac_of_simd_cycles": k.get("valu_busy_frac_of_simd_cycles"),
                            "valu_insts_per_pixel": k.get("valu_insts_per_pixel"), "wave_cycle_split": k.get("wave_cycle_split"),
                            "source": "rocprofv3 --pmc SQ_* pass, profiles/r01_pmc_traffic.json"}
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
        "roofline": {"bound": "hbm", "kernel": "SAD search (%s)" % m.search_variant,
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "traffic_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE(x2)/WRITE_SIZE, profiles/r01_pmc_traffic.json",
                     "avg_launch_ms": round(avg_ms, 4), "pairs_per_launch": frames_per_launch,
                     "algorithmic_bytes_per_pair": ALGO_BYTES_PER_PAIR},
        "stage_ms_per_launch": {k: round(v["total_ms"] / max(1, v["launches"]), 4) for k, v in stages.items()},
        "compute_view": compute_view,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

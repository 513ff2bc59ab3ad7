"""Whole-frame sharding of an independent-pair stream across the GPUs of one node.

The reference is stateless per frame (estimator.cpp:18-82 carries nothing across iterations), so a
stream of N pairs partitions into contiguous blocks, one per rank, with no exchange inside the
path.  Two modes:

  * comm-free (bench.py, weak scaling): every rank synthesises its own block from
    (seed, global frame index) directly in HBM -- no collective at all.
  * root-sourced (BASELINE config 4, "batched stream"): rank 0 owns the frames; inputs are
    scattered and disparities gathered with torch.distributed collectives (backend "nccl" is RCCL
    over xGMI on ROCm; "gloo" on CPU for the tests).  Only scatter/gather: there is no reduction.

One process per GPU; nothing here touches the compute path itself (`compute` is a callable).
"""
import torch


def partition(n_frames, world, rank):
    """Contiguous block partition: rank r gets frames [start, start+count)."""
    start = n_frames * rank // world
    end = n_frames * (rank + 1) // world
    return start, end - start


def shard_sizes(n_frames, world):
    return [partition(n_frames, world, r)[1] for r in range(world)]


def scatter_compute_gather(dist, left, right, n_frames, frame_shape, compute, device, out_dtype=torch.int16,
                           chunk=None):
    """left/right: uint8 [n_frames, H, W] on rank 0 (ignored elsewhere).  Returns the gathered
    [n_frames, H, W] disparities on rank 0 (None elsewhere).  `compute(L, R) -> D` runs on this
    rank's shard (tensors on `device`).  Shards are padded to the largest block so that the
    collectives see equal sizes; `chunk` (frames) bounds the size of each scatter/gather so that
    communication of chunk k+1 can overlap the compute of chunk k on a separate stream."""
    world, rank = dist.get_world_size(), dist.get_rank()
    H, W = frame_shape
    counts = shard_sizes(n_frames, world)
    mine = counts[rank]
    cap = max(counts)
    step = cap if not chunk else min(chunk, cap)
    out_local = torch.empty((cap, H, W), dtype=out_dtype, device=device)
    gathered = [torch.empty((cap, H, W), dtype=out_dtype, device=device) for _ in range(world)] if rank == 0 else None
    for c0 in range(0, cap, step):
        c1 = min(c0 + step, cap)
        m = c1 - c0
        lbuf = torch.empty((m, H, W), dtype=torch.uint8, device=device)
        rbuf = torch.empty((m, H, W), dtype=torch.uint8, device=device)
        lsrc = rsrc = None
        if rank == 0:
            lsrc, rsrc = [], []
            for r in range(world):
                s, cnt = partition(n_frames, world, r)
                lo, hi = min(c0, cnt), min(c1, cnt)
                lp = torch.zeros((m, H, W), dtype=torch.uint8, device=device)
                rp = torch.zeros((m, H, W), dtype=torch.uint8, device=device)
                if hi > lo:
                    lp[:hi - lo] = left[s + lo:s + hi].to(device)
                    rp[:hi - lo] = right[s + lo:s + hi].to(device)
                lsrc.append(lp); rsrc.append(rp)
        dist.scatter(lbuf, lsrc, src=0)
        dist.scatter(rbuf, rsrc, src=0)
        live = max(0, min(c1, mine) - c0)
        if live > 0:
            out_local[c0:c0 + live] = compute(lbuf[:live], rbuf[:live])
    # collectives move raw bytes: gloo has no int16 kernels, and the payload is opaque anyway
    dist.gather(out_local.view(torch.uint8), [g.view(torch.uint8) for g in gathered] if rank == 0 else None, dst=0)
    if rank != 0:
        return None
    out = torch.empty((n_frames, H, W), dtype=out_dtype, device=device)
    for r in range(world):
        s, cnt = partition(n_frames, world, r)
        out[s:s + cnt] = gathered[r][:cnt]
    return out


def local_stream(first_frame, n_frames, world, rank):
    """Comm-free mode: the global frame indices this rank synthesises and processes."""
    s, cnt = partition(n_frames, world, rank)
    return first_frame + s, cnt

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
import contextlib
import time

import torch


def partition(n_frames, world, rank):
    """Contiguous block partition: rank r gets frames [start, start+count)."""
    start = n_frames * rank // world
    end = n_frames * (rank + 1) // world
    return start, end - start


def shard_sizes(n_frames, world):
    return [partition(n_frames, world, r)[1] for r in range(world)]


class _Streams:
    """The two queues of the pipeline: `main` (the caller's current stream: compute) and `comm` (a side
    stream from which the collectives are issued).  On a CPU device both are the program order."""

    def __init__(self, device):
        self.cuda = device.type == "cuda"
        if self.cuda:
            self.main = torch.cuda.current_stream(device)
            self.comm = torch.cuda.Stream(device)
            self.comm.wait_stream(self.main)          # the caller's inputs are ready on `main`

    def on_comm(self):
        return torch.cuda.stream(self.comm) if self.cuda else contextlib.nullcontext()

    def mark(self, comm=False, timing=False):
        """An event at the current tail of `main` (or `comm`); None on CPU (timing: the host clock there)."""
        if not self.cuda:
            return time.perf_counter() if timing else None
        ev = torch.cuda.Event(enable_timing=timing)
        ev.record(self.comm if comm else self.main)
        return ev

    def comm_waits(self, ev):
        if ev is not None:
            self.comm.wait_event(ev)

    def main_waits(self, ev):
        if ev is not None:
            self.main.wait_event(ev)

    def join(self):
        if self.cuda:
            self.main.wait_stream(self.comm)


def new_timings():
    """Collector for scatter_compute_gather(timings=...): per-chunk (begin, end) marks of the three phases."""
    return {"scatter": [], "compute": [], "gather": []}


def phase_ms(timings):
    """Milliseconds per phase, summed over the chunks of one pass (call after the device has been synchronised): HIP-event
    intervals on the stream the phase was issued from -- scatter and gather on the side stream, compute on the caller's -- or
    host clock differences on a CPU device, where every phase blocks."""
    out = {}
    for k, marks in timings.items():
        tot = 0.0
        for a, b in marks:
            tot += a.elapsed_time(b) if hasattr(a, "elapsed_time") else (b - a) * 1e3
        out[k + "_ms"] = round(tot, 3)
    out["chunks"] = len(timings["compute"])
    return out


def scatter_compute_gather(dist, left, right, n_frames, frame_shape, compute, device, out_dtype=torch.int16,
                           chunk=None, compute_into=None, timings=None):
    """left/right: uint8 [n_frames, H, W] on rank 0 (ignored elsewhere).  Returns the gathered
    [n_frames, H, W] disparities on rank 0 (None elsewhere).  `compute(L, R) -> D` runs on this rank's
    shard (tensors on `device`); `compute_into(L, R, out)`, if given, is used instead and writes `out`.

    The shard is cut into chunks of `chunk` frames and pipelined over two buffer sets: while chunk k is
    computed on the caller's stream, the scatter of chunk k+1 and the gather of chunk k-1 run from a side
    stream (collectives are issued asynchronously from it; their completion is joined to the compute
    stream with Work.wait(), which does not block the host on a GPU).  The root passes VIEWS of its frames
    to the scatter and views of the result to the gather; only a ragged last chunk (a rank whose block is
    one frame shorter) goes through a padded temporary.

    timings (new_timings()): every phase of every chunk is bracketed by marks on the stream it is issued from.  The end mark
    of a collective needs the issuing stream to wait for it (Work.wait()), which serialises the side stream's collectives --
    they are in order on RCCL's own stream anyway -- so a timed pass is for diagnosis, not for the throughput figure."""
    world, rank = dist.get_world_size(), dist.get_rank()
    H, W = frame_shape
    counts = shard_sizes(n_frames, world)
    starts = [partition(n_frames, world, r)[0] for r in range(world)]
    mine, cap = counts[rank], max(counts)
    step = cap if not chunk else max(1, min(chunk, cap))
    chunks = [(c0, min(c0 + step, cap)) for c0 in range(0, cap, step)]
    st = _Streams(device)
    NB = 2
    lbuf = [torch.empty((step, H, W), dtype=torch.uint8, device=device) for _ in range(NB)]
    rbuf = [torch.empty((step, H, W), dtype=torch.uint8, device=device) for _ in range(NB)]
    obuf = [torch.empty((step, H, W), dtype=out_dtype, device=device) for _ in range(NB)]
    out = torch.empty((n_frames, H, W), dtype=out_dtype, device=device) if rank == 0 else None
    in_free = [None] * NB       # event on main: the compute that read lbuf/rbuf[b] has been enqueued and finished
    out_busy = [None] * NB      # Work of the gather that reads obuf[b]
    keep = []                   # temporaries referenced by in-flight collectives

    def src_views(frames, c0, c1):
        """What the root hands to the scatter for chunk [c0, c1): one [m, H, W] tensor per rank."""
        m, lst = c1 - c0, []
        for r in range(world):
            lo, hi = min(c0, counts[r]), min(c1, counts[r])
            v = frames[starts[r] + lo:starts[r] + hi]
            if v.device != device:
                v = v.to(device, non_blocking=True)
            if hi - lo < m:                         # ragged: this rank's block ends inside the chunk
                p = torch.zeros((m, H, W), dtype=torch.uint8, device=device)
                if hi > lo:
                    p[:hi - lo] = v
                v = p
            lst.append(v.contiguous())
        keep.extend(lst)
        return lst

    def issue_scatter(k):
        b, (c0, c1) = k % NB, chunks[k]
        m = c1 - c0
        with st.on_comm():
            st.comm_waits(in_free[b])
            ls = src_views(left, c0, c1) if rank == 0 else None
            rs = src_views(right, c0, c1) if rank == 0 else None
            t_a = st.mark(comm=True, timing=True) if timings is not None else None
            wks = (dist.scatter(lbuf[b][:m], ls, src=0, async_op=True),
                   dist.scatter(rbuf[b][:m], rs, src=0, async_op=True))
            if timings is not None:
                for wk in wks:
                    wk.wait()
                timings["scatter"].append((t_a, st.mark(comm=True, timing=True)))
            return wks

    def issue_gather(k, done):
        b, (c0, c1) = k % NB, chunks[k]
        m = c1 - c0
        with st.on_comm():
            st.comm_waits(done)
            dst, fix = None, []
            if rank == 0:
                dst = []
                for r in range(world):
                    lo, hi = min(c0, counts[r]), min(c1, counts[r])
                    if hi - lo == m:
                        dst.append(out[starts[r] + lo:starts[r] + hi])
                    else:
                        t = torch.empty((m, H, W), dtype=out_dtype, device=device)
                        dst.append(t); fix.append((t, starts[r] + lo, hi - lo))
                keep.extend(dst)
            # collectives move raw bytes: gloo has no int16 kernels, and the payload is opaque anyway
            t_a = st.mark(comm=True, timing=True) if timings is not None else None
            wk = dist.gather(obuf[b][:m].view(torch.uint8),
                             [d.view(torch.uint8) for d in dst] if rank == 0 else None, dst=0, async_op=True)
            if timings is not None:
                wk.wait()
                timings["gather"].append((t_a, st.mark(comm=True, timing=True)))
            if fix:
                wk.wait()                              # comm stream waits (host too on CPU); then the ragged tails
                for t, s0, cnt in fix:
                    if cnt > 0:
                        out[s0:s0 + cnt] = t[:cnt]
            return wk

    pend = issue_scatter(0)
    for k, (c0, c1) in enumerate(chunks):
        b = k % NB
        nxt = issue_scatter(k + 1) if k + 1 < len(chunks) else None
        for wk in pend:
            wk.wait()                                  # main waits for chunk k's inputs
        if out_busy[b] is not None:
            out_busy[b].wait()                         # obuf[b] is free again (gather of chunk k-2)
        live = max(0, min(c1, mine) - c0)
        t_a = st.mark(timing=True) if timings is not None else None
        if live > 0:
            if compute_into is not None:
                compute_into(lbuf[b][:live], rbuf[b][:live], obuf[b][:live])
            else:
                obuf[b][:live] = compute(lbuf[b][:live], rbuf[b][:live])
        if timings is not None:
            timings["compute"].append((t_a, st.mark(timing=True)))
        done = st.mark()
        in_free[b] = done
        out_busy[b] = issue_gather(k, done)
        pend = nxt
    for wk in out_busy:
        if wk is not None:
            wk.wait()
    st.join()
    return out


def local_stream(first_frame, n_frames, world, rank):
    """Comm-free mode: the global frame indices this rank synthesises and processes."""
    s, cnt = partition(n_frames, world, rank)
    return first_frame + s, cnt

"""Multi-GPU path on CPU: world_size-2 (and 3) gloo runs of the whole-frame sharding.  The compute
callable is the oracle here (tests may use it); on the GPU box the same function wraps the HIP
matcher.  Property: the gathered N-rank output is byte-identical to the 1-rank output."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load

W, H, D, NF = 96, 64, 16, 7


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _oracle_compute(L, R):
    from oracle import oracle as orc
    out = [orc.bm_compute(l.numpy(), r.numpy(), numDisparities=D, blockSize=7) for l, r in zip(L, R)]
    return torch.from_numpy(np.stack(out))


def _worker(rank, world, port, chunk, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load()
    sh = load("sharding")
    left = right = None
    if rank == 0:
        L, R = pkg.synth.make_stream(0, NF, W, H, D)
        left, right = torch.from_numpy(L), torch.from_numpy(R)
    tm = sh.new_timings() if chunk == 2 else None
    out = sh.scatter_compute_gather(dist, left, right, NF, (H, W), _oracle_compute, torch.device("cpu"), chunk=chunk, timings=tm)
    if tm is not None:                       # per-phase marks: one triple per chunk, host clock on a CPU device
        ph = sh.phase_ms(tm)
        nchunks = -(-max(sh.shard_sizes(NF, world)) // chunk)
        assert ph["chunks"] == nchunks and len(tm["scatter"]) == nchunks and len(tm["gather"]) == nchunks
        assert ph["compute_ms"] > 0 and ph["scatter_ms"] >= 0 and ph["gather_ms"] >= 0
    # comm-free mode: every rank makes its own frames; rank 0 checks its block against the stream
    first, cnt = sh.local_stream(0, NF, world, rank)
    Lr, Rr = pkg.synth.make_stream(first, cnt, W, H, D)
    mine = _oracle_compute(torch.from_numpy(Lr), torch.from_numpy(Rr))
    blocks = [None] * world
    dist.all_gather_object(blocks, mine.numpy())
    if rank == 0:
        q.put((out.numpy(), np.concatenate(blocks)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunk", [(2, None), (2, 2), (2, 1), (3, None), (3, 2)])
def test_sharded_equals_single_rank(world, chunk):
    pkg = load()
    L, R = pkg.synth.make_stream(0, NF, W, H, D)
    ref = _oracle_compute(torch.from_numpy(L), torch.from_numpy(R)).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, chunk, q)) for r in range(world)]
    [p.start() for p in procs]
    gathered, commfree = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert np.array_equal(gathered, ref)
    assert np.array_equal(commfree, ref)


def test_partition_is_a_contiguous_cover():
    sh = load("sharding")
    for n in (1, 7, 128, 1024):
        for world in (1, 2, 3, 8):
            spans = [sh.partition(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
    assert sh.shard_sizes(1024, 8) == [128] * 8


def test_bench_launcher_relays_the_ranks_failure_without_a_gpu():
    """`python bench.py --gpus 2` with no torch.distributed environment starts its own ranks (the parent touches no GPU and
    does not exec).  In this container there is no HIP device: every rank must fail loudly ("no CPU fallback") and the
    launcher must hand the failure on instead of printing a result line."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered on the GPU box by tests/test_gpu_round2.py::test_bench_gpus2_launches_two_ranks_itself")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "needs a HIP device" in p.stderr

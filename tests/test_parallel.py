"""CPU, world_size 2 over gloo: the clip sharding + all-gather that bench.py / inference use on RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gcd_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = parallel.clips_for_rank(num_clips, rank, world)
        # "denoise" each clip: a deterministic function of the clip id
        local = [torch.full((14, 4, 3, 5), float(c)) + torch.arange(5.0) for c in mine]
        got = parallel.gather_ragged_clips(local, num_clips, dist)
        ok = all(torch.equal(got[c], torch.full((14, 4, 3, 5), float(c)) + torch.arange(5.0))
                 for c in range(num_clips))
        eq = parallel.gather_clips(torch.full((2, 3), float(rank)), dist)
        ok = ok and eq.shape == (world, 2, 3) and all(float(eq[r, 0, 0]) == r for r in range(world))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_clips", [2, 5])
def test_gloo_world2_gather(num_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}


def test_clip_assignment_covers_everything():
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 8, 9, 64):
            seen = sorted(c for r in range(world) for c in parallel.clips_for_rank(n, r, world))
            assert seen == list(range(n))
    with pytest.raises(ValueError):
        parallel.clips_for_rank(4, 2, 2)


def test_single_process_passthrough():
    t = torch.randn(3, 4)
    assert torch.equal(parallel.gather_clips(t)[0], t)
    assert parallel.gather_ragged_clips([t], 1)[0] is t

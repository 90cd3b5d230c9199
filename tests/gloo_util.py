"""World-size-N gloo harness for the CPU tests of the multi-GPU paths (clip sharding, gradient exchange).

Rendezvous is a FileStore in a fresh temp directory: no TCP port is reserved in the parent and re-bound in the
children (the close-then-bind race of a "free port" helper made these tests flaky under load)."""
import os
import tempfile

import torch.distributed as dist
import torch.multiprocessing as mp


def init(rank: int, world: int, store_path: str, backend: str = "gloo") -> None:
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # the container hostname may not resolve
    dist.init_process_group(backend, store=dist.FileStore(store_path, world), rank=rank, world_size=world)


def run_world(target, world: int, *args, timeout: float = 600.0):
    """spawn `world` ranks of target(rank, world, store_path, *args, q); every rank must q.put((rank, payload)).
    Returns {rank: payload}.  Results are drained BEFORE join: a child blocks at exit until its queue
    items have been consumed."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    path = os.path.join(tempfile.mkdtemp(prefix="gcd_gloo_"), "store")
    procs = [ctx.Process(target=target, args=(r, world, path) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = dict(q.get(timeout=timeout) for _ in range(world))
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    return res

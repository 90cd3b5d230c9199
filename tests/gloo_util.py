"""World-size-N gloo harness for the CPU tests of the multi-GPU paths (clip sharding, gradient exchange).

Rendezvous is a FileStore in a fresh temp directory: no TCP port is reserved in the parent and re-bound in the
children (the close-then-bind race of a "free port" helper made these tests flaky under load)."""
import os
import tempfile

import torch.distributed as dist
import torch.multiprocessing as mp


def backend_name() -> str:
    """gloo (default; CPU tests and the two-ranks-on-one-GPU tests) or, on a multi-GPU node, GCD_DIST_BACKEND=nccl
    (= RCCL; tools/first_multi_gpu.sh)."""
    return os.environ.get("GCD_DIST_BACKEND", "gloo")


def rank_device(rank: int):
    """cuda:0 for every rank (one-GPU lease) unless GCD_TEST_GPUS_PER_RANK=1 gives each rank its own GPU."""
    import torch
    if os.environ.get("GCD_TEST_GPUS_PER_RANK") == "1":
        return torch.device(f"cuda:{rank % torch.cuda.device_count()}")
    return torch.device("cuda:0")


def init(rank: int, world: int, store_path: str, backend: str = None) -> None:
    backend = backend or backend_name()
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

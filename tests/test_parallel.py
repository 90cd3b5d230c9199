"""CPU, world_size 2 over gloo: the clip sharding + all-gather that bench.py / inference use on RCCL."""
import pytest
import torch
import torch.distributed as dist
from gcd_amd import parallel
from gloo_util import init as _init, run_world as _run_world


def _worker(rank, world, port, num_clips, q):
    _init(rank, world, port)
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
    assert _run_world(_worker, 2, num_clips) == {0: True, 1: True}


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


# ---------------------------------------------------------------------------------------------
# sample_clips: strided clips -> per-clip seeded EulerEDM loop -> ragged gather (scripts/test.py:1051-1090)
# ---------------------------------------------------------------------------------------------
class _StubDenoiser:
    """Stands in for the UNet stack: a cheap deterministic function of (input, sigma, cond)."""

    def __call__(self, x, sigma, c):
        s = sigma.reshape(-1, 1, 1, 1)
        return x / (1.0 + s * s) + 0.1 * torch.tanh(c["concat"]) + 0.01 * c["vector"].mean(1).reshape(-1, 1, 1, 1)


def _sample_all(num_clips, dist_mod):
    from gcd_amd.sampling import EulerEDMSampler
    T = 14
    sampler = EulerEDMSampler(
        discretization_config={"target": "gcd_amd.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
        num_steps=6,
        guider_config={"target": "gcd_amd.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": 1.5, "min_scale": 1.0}},
        device="cpu")
    den = _StubDenoiser()

    def one(i, g):
        noise = torch.randn(T, 4, 6, 8, generator=g)
        c = {"concat": torch.full((T, 4, 6, 8), 0.1 * i), "vector": torch.full((T, 8), float(i)),
             "crossattn": torch.zeros(T, 1, 4)}
        uc = {"concat": torch.zeros_like(c["concat"]), "vector": c["vector"], "crossattn": c["crossattn"]}
        return sampler(den, noise, cond=c, uc=uc)

    return parallel.sample_clips(one, num_clips, dist_mod, base_seed=1000)


def _sample_worker(rank, world, port, num_clips, q):
    _init(rank, world, port)
    try:
        got = _sample_all(num_clips, dist)
        q.put((rank, [t.clone() for t in got]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_clips", [1, 5])
def test_sample_clips_world2_equals_single_process(num_clips):
    """1 clip: rank 1 holds nothing (the empty-rank pad must match dtype / device); 5 clips: ragged."""
    want = _sample_all(num_clips, None)
    assert len(want) == num_clips and all(t.shape == (14, 4, 6, 8) for t in want)
    res = _run_world(_sample_worker, 2, num_clips)
    for r in (0, 1):
        assert len(res[r]) == num_clips
        for a, b in zip(res[r], want):
            assert torch.equal(a, b), "a clip's result depends on the number of ranks"
    if num_clips > 1:
        assert not torch.equal(want[0], want[1])


def test_sample_clips_without_gather_keeps_local_results():
    got = parallel.sample_clips(lambda i, g: torch.full((2,), float(i)), 3, None, gather=False)
    assert [float(t[0]) for t in got] == [0.0, 1.0, 2.0]

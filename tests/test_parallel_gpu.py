"""GPU: the multi-GPU path of the sampler without a multi-GPU node — two ranks, ONE GPU, gloo.

`gcd_amd.parallel.sample_clips` shards clips over the ranks (clip i -> rank i mod world, noise seeded by the clip
index) and all-gathers the final latents; on an 8-GPU node the ranks sit on 8 GPUs and the backend is nccl (= RCCL
over xGMI, `bench.py --gpus 8`).  Here both ranks share cuda:0 and talk over gloo, which exercises everything but
the transport: the real HIP VideoUNet, the fused hipGraph EulerEDM loop, the ragged gather — and the property the
sharding promises: a clip's result does not depend on how many ranks the job ran on, bit for bit."""
import pytest
import torch

import gloo_util

pytestmark = pytest.mark.gpu


def _sampler_and_net(dev):
    from gcd_amd.denoiser import Denoiser
    from gcd_amd.sampling import EulerEDMSampler, FusedDenoiser
    from gcd_amd.video_model import VideoUNet
    from gcd_amd.wrappers import OpenAIWrapper
    from oracle import svd_unet_ref as O, weights
    cfg = O.TINY
    with torch.device("meta"):
        net = VideoUNet(**cfg.as_reference_kwargs())
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net = net.to_empty(device=dev)
    net.load_state_dict(sd)
    net.eval()
    T = 14
    sampler = EulerEDMSampler(
        discretization_config={"target": "gcd_amd.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
        num_steps=6,
        guider_config={"target": "gcd_amd.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": 1.5, "min_scale": 1.0}},
        device="cuda")
    fd = FusedDenoiser(Denoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"}), OpenAIWrapper(net),
                       num_video_frames=T, image_only_indicator=torch.zeros(2, T, device=dev))

    def one(i, g):
        noise, c, uc = weights.synth_inputs(1, T, 8, 8, cfg.context_dim, cfg.adm_in_channels + cfg.aux_emb_dim, 300 + i)
        x = torch.randn(noise.shape, generator=g, device=dev)      # the clip's own generator: seed = base + clip index
        out = sampler(fd, x, cond={k: v.to(dev) for k, v in c.items()}, uc={k: v.to(dev) for k, v in uc.items()})
        assert sampler.last_path == "fused"
        return out.cpu()
    return one


def _worker(rank, world, store, num_clips, q):
    import torch.distributed as dist
    from gcd_amd import parallel
    dev = gloo_util.rank_device(rank)
    torch.cuda.set_device(dev)
    gloo_util.init(rank, world, store)
    try:
        one = _sampler_and_net(dev)
        got = parallel.sample_clips(one, num_clips, dist, base_seed=900, device=dev)
        ok = len(got) == num_clips and all(t is not None and t.shape == (14, 4, 8, 8) for t in got)
        if rank == 0:
            want = parallel.sample_clips(one, num_clips, None, base_seed=900, device=dev)
            ok = ok and all(torch.equal(a, b) for a, b in zip(got, want))
            ok = ok and not torch.equal(want[0], want[1])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_clips", [2, 3])
def test_two_ranks_one_gpu_fused_loop_equals_single_process(gpu, num_clips):
    assert gloo_util.run_world(_worker, 2, num_clips, timeout=900) == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------------------------
# cfg4 in miniature: data-parallel fine-tune step, two ranks on one GPU over gloo — each rank runs forward + backward
# of the TINY VideoUNet on ITS batch on the HIP training path, gradients are exchanged by GradBucketer (all-reduce
# launched from gradient hooks while the backward pass is still running), AdamHIP steps; the averaged gradients equal
# the mean of the two per-batch gradients computed in one process, and both ranks end with identical weights.
# ---------------------------------------------------------------------------------------------------------------
def _train_setup(dev, seed):
    from gcd_amd import training as TR
    from gcd_amd.video_model import VideoUNet
    from oracle import svd_unet_ref as O, weights
    cfg = O.TINY
    with torch.device("meta"):
        net = VideoUNet(**cfg.as_reference_kwargs())
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 6)
    net = net.to_empty(device=dev)
    net.load_state_dict(sd)
    net.train()
    T, H, W = 4, 16, 16
    g = torch.Generator().manual_seed(seed)
    batch = dict(
        x0=torch.randn(T, 4, H, W, generator=g).to(dev), noise=torch.randn(T, 4, H, W, generator=g).to(dev),
        sig=torch.full((T,), float(torch.randn(1, generator=g).mul(1.6).add(1.0).exp())).to(dev),
        cond={"crossattn": torch.randn(T, 1, cfg.context_dim, generator=g).to(dev),
              "concat": (torch.randn(T, 4, H, W, generator=g) * 0.8).to(dev),
              "vector": torch.randn(T, cfg.adm_in_channels + cfg.aux_emb_dim, generator=g).clamp(-1, 1).to(dev)})
    den = TR.TrainDenoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"})

    def loss_of(b):
        noised = b["x0"] + b["noise"] * b["sig"][:, None, None, None]
        out = den(net, noised, b["sig"], b["cond"], num_video_frames=T, image_only_indicator=torch.zeros(1, T, device=dev))
        return ((out - b["x0"]) ** 2).mean()
    return net, batch, loss_of


def _ddp_worker(rank, world, store, q):
    import torch.distributed as dist
    from gcd_amd import training as TR
    dev = gloo_util.rank_device(rank)
    torch.cuda.set_device(dev)
    gloo_util.init(rank, world, store)
    try:
        net, mine, loss_of = _train_setup(dev, 500 + rank)
        bucketer = TR.GradBucketer(net.parameters(), dist, bucket_bytes=1 << 20)
        (loss_of(mine) * 256.0).backward()
        nb = bucketer.finish()
        bucketer.close()          # the single-process reference passes below must not fire the hooks again
        ok = nb >= 3 and bucketer.launched_during_backward >= 1
        got = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        if rank == 0:       # the same two batches in ONE process: mean of the two gradients
            ref = {}
            for r in range(world):
                _, other, _ = _train_setup(dev, 500 + r)
                for p in net.parameters():
                    p.grad = None
                (loss_of(other) * 256.0).backward()
                for n, p in net.named_parameters():
                    if p.grad is not None:
                        ref[n] = ref.get(n, 0) + p.grad / world
            num = sum(float((got[n].double() - ref[n].double()).pow(2).sum()) for n in ref)
            den_ = sum(float(ref[n].double().pow(2).sum()) for n in ref)
            err = (num / den_) ** 0.5
            print(f"data-parallel gradients vs single-process mean: global rel-L2 {err:.2e}")
            ok = ok and err < 1e-3 and set(ref) <= set(got)
            for n, p in net.named_parameters():
                p.grad = got.get(n)
        opt = TR.AdamHIP(net.parameters(), lr=1e-4)
        opt.step(grad_scale=1.0 / 256.0)
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        if gloo_util.backend_name() != "nccl":
            flat = flat.cpu()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        ok = ok and all(torch.equal(gathered[0], t) for t in gathered[1:])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_data_parallel_finetune_step_two_ranks_one_gpu(gpu):
    assert gloo_util.run_world(_ddp_worker, 2, timeout=240) == {0: True, 1: True}

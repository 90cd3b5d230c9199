"""CPU: host side of the fine-tune step (BASELINE.json cfg4): the training loss
against known answers from the unmodified reference classes (oracle/make_golden_loss.py), its oracle
restatement, the plugin socket, and the data-parallel gradient exchange over gloo (world size 2)."""
from pathlib import Path

import pytest
import torch
import torch.distributed as dist

import gloo_util
from conftest import rel_l2
from oracle import loss_ref as R

GOLD = Path(__file__).resolve().parent / "golden" / "loss_kat.pt"
CFG = dict(
    harmonize_sigmas=True, focus_top=0.1, focus_steps=5000,
    batch2model_keys=["image_only_indicator", "num_video_frames"],
    loss_weighting_config={"target": "gcd_amd.training.EDMWeighting", "params": {"sigma_data": 1.0}},
    sigma_sampler_config={"target": "gcd_amd.training.EDMSampling", "params": {"p_mean": 1.0, "p_std": 1.6}})


@pytest.fixture(scope="module")
def g():
    return torch.load(GOLD)


def test_sigma_sampling_and_weighting_vs_reference(g):
    from gcd_amd.training import EDMSampling, EDMWeighting
    sig = EDMSampling(p_mean=1.0, p_std=1.6)(8, rand=g["rand"])
    assert torch.equal(sig, g["sigmas"]) and torch.equal(R.edm_sigmas(g["rand"], 1.0, 1.6), g["sigmas"])
    assert torch.equal(EDMWeighting(sigma_data=1.0)(sig), g["weights"])
    assert torch.equal(R.edm_weighting(sig, 1.0), g["weights"])
    assert EDMSampling()(5).shape == (5,)


def test_get_loss_vs_reference_known_answers(g):
    """L2 / L1, the annealed top-fraction focal loss at steps before, inside and after the annealing
    window (keep fraction 1 -> 0.1 over 5000 steps), EDM weighting: product and oracle vs reference."""
    from gcd_amd.training import StandardDiffusionLoss
    w = g["weights"][:, None, None, None]
    for case in g["cases"]:
        loss = StandardDiffusionLoss(**dict(CFG, loss_type=case["loss_type"]))
        got = loss.get_loss(g["out"], g["tgt"], w, {"global_step": case["step"]})
        ora = R.get_loss(g["out"], g["tgt"], w, case["step"], case["loss_type"], 0.1, 5000)
        assert torch.allclose(got, case["loss"], rtol=1e-6, atol=0), case
        assert torch.allclose(ora, case["loss"], rtol=1e-6, atol=0), case
    with pytest.raises(AssertionError):
        StandardDiffusionLoss(**dict(CFG, loss_type="huber"))


def test_get_loss_parallel_domain_class_weights_vs_reference_known_answers():
    """loss.py:196-230 (configs/train_pardom_semantic.yaml:145-146: person x7, vehicle x3): semantic-map colours ->
    area-averaged latent masks -> half of the weighted loss before the focal top-fraction, half after.  Known answers
    from the unmodified reference get_loss (oracle/make_golden_loss.py pd) for person+vehicle / person only / vehicle
    only, L2 / L1, before / inside / after the focal annealing window; product and oracle."""
    from oracle.make_golden_loss import pd_semantic_frames
    from gcd_amd.training import StandardDiffusionLoss
    g = torch.load(GOLD.parent / "loss_pd_kat.pt")
    jpg = pd_semantic_frames(g["out"].shape[0], *g["jpg_hw"], seed=g["jpg_seed"])
    w = g["weights"][:, None, None, None]
    assert len(g["cases"]) == 18
    for case in g["cases"]:
        loss = StandardDiffusionLoss(**dict(CFG, loss_type=case["loss_type"], pd_person_weight=case["person"],
                                            pd_vehicle_weight=case["vehicle"]))
        got = loss.get_loss(g["out"], g["tgt"], w, {"global_step": case["step"], "jpg": jpg})
        ora = R.get_loss_pd(g["out"], g["tgt"], w, case["step"], jpg, case["person"], case["vehicle"],
                            case["loss_type"], 0.1, 5000)
        assert torch.allclose(got, case["loss"], rtol=2e-6, atol=0), case
        assert torch.allclose(ora, case["loss"], rtol=2e-6, atol=0), case
    # the weighting changes the loss (the fixture paints pedestrians / cars into every frame), and it is differentiable
    plain = StandardDiffusionLoss(**CFG).get_loss(g["out"], g["tgt"], w, {"global_step": 1})
    assert float((g["cases"][0]["loss"] / plain).min()) > 1.01
    out = g["out"].clone().requires_grad_(True)
    StandardDiffusionLoss(**dict(CFG, pd_person_weight=7.0, pd_vehicle_weight=3.0)).get_loss(
        out, g["tgt"], w, {"global_step": 2501, "jpg": jpg}).sum().backward()
    assert bool(torch.isfinite(out.grad).all()) and float(out.grad.abs().sum()) > 0


def test_forward_plumbing_vs_reference(g):
    """_forward: per-clip harmonised sigmas, noised input, batch2model_keys, weighting — same global RNG
    seed as the reference run, so the draws are identical."""
    from gcd_amd.training import StandardDiffusionLoss
    f = g["forward"]
    seen = {}

    def denoiser(network, noised, sigmas, cond, **kw):
        seen.update(noised=noised.clone(), sigmas=sigmas.clone(), kw=dict(kw))
        return noised * 0.5

    T = g["T"]
    torch.manual_seed(f["seed"])
    batch = {"global_step": 2500, "num_video_frames": T, "image_only_indicator": torch.zeros(2, T),
             "unrelated": 1}
    val = StandardDiffusionLoss(**CFG)._forward(None, denoiser, {}, g["tgt"], batch)
    assert torch.equal(seen["sigmas"], f["sigmas"]) and torch.equal(seen["noised"], f["noised"])
    assert sorted(seen["kw"]) == f["kw_keys"]
    assert torch.allclose(val, f["loss"], rtol=1e-6, atol=0)
    s = seen["sigmas"].reshape(-1, T)
    assert torch.equal(s, s[:, :1].expand_as(s))                     # one noise level per clip
    assert torch.equal(R.harmonize(torch.arange(8.0), 4), torch.tensor([0., 0, 0, 0, 4, 4, 4, 4]))


def test_loss_socket_instantiates_from_config():
    from gcd_amd.util import instantiate_from_config
    loss = instantiate_from_config({"target": "gcd_amd.training.StandardDiffusionLoss", "params": CFG})
    assert loss.batch2model_keys == {"image_only_indicator", "num_video_frames"} and loss.focus_steps == 5000


# ------------------------------------------------------------------------------- DDP gradient exchange
def _worker(rank, world, port, q):
    from gcd_amd.training import allreduce_gradients
    gloo_util.init(rank, world, port)
    try:
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(s)) for s in [(300, 7), (5,), (64, 64), (1,)]]
        frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
        for i, p in enumerate(params):
            p.grad = torch.full(p.shape, float(rank + 1) * (i + 1))
        if rank == 0:
            params[1].grad = None           # ranks may disagree on which parameters got a gradient: zeros enter
        nb = allreduce_gradients(params + [frozen], dist, bucket_bytes=4096)
        ok = nb >= 2
        for i, p in enumerate(params):
            if i == 1:
                ok = ok and torch.allclose(p.grad, torch.full(p.shape, 2.0))             # (0 + 2 * 2) / 2
            else:
                ok = ok and torch.allclose(p.grad, torch.full(p.shape, 1.5 * (i + 1)))   # mean of ranks 1, 2
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_allreduce_gradients_gloo_world2():
    assert gloo_util.run_world(_worker, 2) == {0: True, 1: True}


def _bucketer_worker(rank, world, port, q):
    """GradBucketer (all-reduce launched from gradient hooks during backward) == allreduce_gradients == the mean
    over ranks of a small model's gradients, including a parameter the graph never reaches, and the weights after
    an Adam step are bit-identical on both ranks."""
    from gcd_amd.training import GradBucketer, allreduce_gradients
    gloo_util.init(rank, world, port)
    try:
        torch.manual_seed(0)                        # same initial weights on every rank
        net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                                  torch.nn.Linear(64, 4))
        unused = torch.nn.Linear(8, 8)              # trainable, never in the graph
        params = list(net.parameters()) + list(unused.parameters())
        g = torch.Generator().manual_seed(100 + rank)          # a different batch per rank
        x, y = torch.randn(32, 16, generator=g), torch.randn(32, 4, generator=g)
        # reference: plain backward + the post-hoc exchange
        ((net(x) - y) ** 2).mean().backward()
        allreduce_gradients(params, dist, bucket_bytes=4096)
        want = [p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        b = GradBucketer(params, dist, bucket_bytes=4096)
        assert len(b.buckets) >= 3
        ((net(x) - y) ** 2).mean().backward()
        nb = b.finish()
        ok = nb == len(b.buckets) and b.launched_during_backward >= 2
        ok = ok and all(torch.equal(p.grad, w) for p, w in zip(params, want))
        ok = ok and all(float(p.grad.abs().max()) == 0.0 for p in unused.parameters())
        # a second step through the same bucketer (state reset) and an optimizer step: identical weights on all ranks
        opt = torch.optim.Adam(params, lr=1e-2)
        opt.step()
        for p in params:
            p.grad = None
        ((net(x) - y) ** 2).mean().backward()
        b.finish()
        opt.step()
        b.close()
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        ok = ok and all(torch.equal(gathered[0], t) for t in gathered[1:])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_grad_bucketer_overlapped_allreduce_gloo_world2():
    assert gloo_util.run_world(_bucketer_worker, 2) == {0: True, 1: True}


def _bucketer_late_worker(rank, world, port, q):
    """parameters whose gradient is only final at the END of the backward pass (the planned engine's few-row
    ones) sit in nearly every bucket of the registration order and hold all of them back.  Named as `late` they get the
    last bucket(s) to themselves: here the biases stand in for them and their gradients are delivered (through the
    planned engine's listener interface) only after backward() has returned — every bucket without one must already be
    in flight by then, and the result must still be the mean over ranks."""
    from gcd_amd.training import GradBucketer, allreduce_gradients
    gloo_util.init(rank, world, port)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                                  torch.nn.Linear(64, 4))
        params = list(net.parameters())
        late = [p for n, p in net.named_parameters() if n.endswith("bias")]
        g = torch.Generator().manual_seed(100 + rank)
        x, y = torch.randn(32, 16, generator=g), torch.randn(32, 4, generator=g)
        ((net(x) - y) ** 2).mean().backward()
        allreduce_gradients(params, dist, bucket_bytes=4096)
        want = [p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        b = GradBucketer(params, dist, bucket_bytes=4096, late=late)
        n_late_buckets = sum(1 for bk in b.buckets if any(p is q_ for p in bk for q_ in late))
        ok = all(all(any(p is q_ for q_ in late) for p in bk) or not any(any(p is q_ for q_ in late) for p in bk)
                 for bk in b.buckets)                      # no bucket mixes the two kinds
        ok = ok and any(b.buckets[-1][0] is q_ for q_ in late)      # ... and the late ones come last
        # the weights' gradients arrive from autograd's hooks during backward(); the biases' are held back
        held = {}
        for p in late:
            held[id(p)] = p
        orig = b._on_grad
        b._on_grad = lambda p: None if id(p) in held else orig(p)
        for h in b._hooks:
            h.remove()
        b._hooks = [p.register_post_accumulate_grad_hook(b._on_grad) for p in b.params]
        ((net(x) - y) ** 2).mean().backward()
        ok = ok and b.launched_during_backward == len(b.buckets) - n_late_buckets      # all the others left already
        b._on_grad = orig
        for p in late:                                          # ... what train_plan._small_backward does at the end
            orig(p)
        nb = b.finish()
        ok = ok and nb == len(b.buckets)
        ok = ok and all(torch.equal(p.grad, w) for p, w in zip(params, want))
        b.close()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_grad_bucketer_late_parameters_get_their_own_last_buckets_gloo_world2():
    assert gloo_util.run_world(_bucketer_late_worker, 2) == {0: True, 1: True}


def test_late_parameters_names_the_few_row_parameters():
    """train_plan.late_parameters: every ResBlock's emb_layers, the one-key cross-attention's to_v / to_out, the embedding
    MLPs, time_pos_embed and the mix factors — and none of the token-sized contractions."""
    from gcd_amd import train_plan as TP

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb_layers = torch.nn.Sequential(torch.nn.SiLU(), torch.nn.Linear(8, 8))
            self.in_layers = torch.nn.Sequential(torch.nn.GroupNorm(2, 8), torch.nn.SiLU(), torch.nn.Conv2d(8, 8, 3))
            self.attn2 = torch.nn.Module()
            self.attn2.to_q, self.attn2.to_v = torch.nn.Linear(8, 8, bias=False), torch.nn.Linear(8, 8, bias=False)
            self.attn2.to_out = torch.nn.Sequential(torch.nn.Linear(8, 8))
            self.attn1 = torch.nn.Module()
            self.attn1.to_v = torch.nn.Linear(8, 8, bias=False)
            self.mix_factor = torch.nn.Parameter(torch.zeros(1))

    net = torch.nn.Module()
    net.time_embed = torch.nn.Sequential(torch.nn.Linear(8, 8))
    net.label_emb = torch.nn.Sequential(torch.nn.Linear(8, 8))
    net.aux_label_emb = torch.nn.Sequential(torch.nn.Linear(8, 8))
    net.blocks = torch.nn.ModuleList([Blk()])
    net.blocks[0].time_pos_embed = torch.nn.Sequential(torch.nn.Linear(8, 8))
    names = {n for n, p in net.named_parameters() if any(p is q_ for q_ in TP.late_parameters(net))}
    assert names == {"time_embed.0.weight", "time_embed.0.bias", "label_emb.0.weight", "label_emb.0.bias",
                     "aux_label_emb.0.weight", "aux_label_emb.0.bias", "blocks.0.emb_layers.1.weight",
                     "blocks.0.emb_layers.1.bias", "blocks.0.attn2.to_v.weight", "blocks.0.attn2.to_out.0.weight",
                     "blocks.0.attn2.to_out.0.bias", "blocks.0.mix_factor", "blocks.0.time_pos_embed.0.weight",
                     "blocks.0.time_pos_embed.0.bias"}


def _bucketer_unused_worker(rank, world, port, q):
    """(a) unreached parameters inside EVERY bucket (the attn2.to_q / to_k / norm2 pattern of the VideoUNet)
    must not push the launches into finish(): named up front, or learned from the first pass, every bucket launches
    during backward; a parameter recorded as unreached that later IS reached still ends up with the right mean.
    (b) gradient accumulation: micro-batches under no_sync() + a last synchronised one == the mean over ranks of the
    summed micro-batch gradients; a second synchronised backward() without finish() raises."""
    from gcd_amd.training import GradBucketer, allreduce_gradients
    gloo_util.init(rank, world, port)
    try:
        torch.manual_seed(0)
        blocks = torch.nn.ModuleList([torch.nn.Linear(24, 24) for _ in range(6)])
        dead = torch.nn.ModuleList([torch.nn.Linear(24, 24) for _ in range(6)])      # one unreached layer per block
        params = []
        for blk, dd in zip(blocks, dead):
            params += list(blk.parameters()) + list(dd.parameters())

        def fwd(x, use_dead=None):
            for i, blk in enumerate(blocks):
                x = torch.tanh(blk(x))
                if use_dead is not None and i == use_dead:
                    x = x + 0.1 * dead[i](x)
            return x

        g = torch.Generator().manual_seed(200 + rank)
        xs = [torch.randn(16, 24, generator=g) for _ in range(3)]

        def reference(batches, use_dead=None):
            for p in params:
                p.grad = None
            for x in batches:
                (fwd(x, use_dead) ** 2).mean().backward()
            allreduce_gradients(params, dist, bucket_bytes=5000)
            want = [p.grad.clone() for p in params]
            for p in params:
                p.grad = None
            return want

        # every reference first: no bucketer (and none of its hooks) exists yet
        want1, want_dead, want_acc = reference(xs[:1]), reference(xs[1:2], use_dead=2), reference(xs)
        ok = True
        # (a1) learned: step 1 leaves the buckets with dead layers to finish(); from step 2 on all launch in backward
        b = GradBucketer(params, dist, bucket_bytes=5000)
        nb = len(b.buckets)
        ok = ok and nb >= 4
        (fwd(xs[0]) ** 2).mean().backward()
        first = b.launched_during_backward
        b.finish()
        ok = ok and first < nb and all(torch.equal(p.grad, w) for p, w in zip(params, want1))
        for p in params:
            p.grad = None
        (fwd(xs[0]) ** 2).mean().backward()
        ok = ok and b.launched_during_backward - first == nb
        b.finish()
        ok = ok and all(torch.equal(p.grad, w) for p, w in zip(params, want1))
        # a layer recorded as unreached is reached now: its bucket left with zeros and is reduced again in finish()
        for p in params:
            p.grad = None
        (fwd(xs[1], use_dead=2) ** 2).mean().backward()
        b.finish()
        ok = ok and all(torch.equal(p.grad, w) for p, w in zip(params, want_dead))
        ok = ok and float(dead[2].weight.grad.abs().max()) > 0
        b.close()
        # (a2) named up front: every bucket launches during the very first backward
        for p in params:
            p.grad = None
        b = GradBucketer(params, dist, bucket_bytes=5000, unused=[q_ for dd in dead for q_ in dd.parameters()])
        (fwd(xs[0]) ** 2).mean().backward()
        ok = ok and b.launched_during_backward == nb
        b.finish()
        ok = ok and all(torch.equal(p.grad, w) for p, w in zip(params, want1))
        # (b) accumulation over three micro-batches
        for p in params:
            p.grad = None
        with b.no_sync():
            (fwd(xs[0]) ** 2).mean().backward()
            (fwd(xs[1]) ** 2).mean().backward()
        (fwd(xs[2]) ** 2).mean().backward()
        b.finish()
        ok = ok and all(torch.allclose(p.grad, w, rtol=1e-6, atol=1e-9) for p, w in zip(params, want_acc))
        for p in params:
            p.grad = None
        (fwd(xs[0]) ** 2).mean().backward()
        raised = False
        try:
            (fwd(xs[1]) ** 2).mean().backward()
        except RuntimeError as e:
            raised = "no_sync" in str(e)
        b.finish()
        b.close()
        q.put((rank, bool(ok and raised)))
    finally:
        dist.destroy_process_group()


def test_grad_bucketer_unreached_parameters_and_accumulation_gloo_world2():
    assert gloo_util.run_world(_bucketer_unused_worker, 2) == {0: True, 1: True}


def test_grad_bucketer_single_process_is_inert():
    from gcd_amd.training import GradBucketer
    p = torch.nn.Parameter(torch.zeros(3))
    b = GradBucketer([p], None)
    (p * 2.0).sum().backward()
    assert b.finish() == 0 and torch.equal(p.grad, torch.full((3,), 2.0))


def test_allreduce_single_process_is_noop():
    from gcd_amd.training import allreduce_gradients
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    assert allreduce_gradients([p], None) == 0 and torch.equal(p.grad, torch.ones(3))


# ------------------------------------------------------------------------------- packed-operand cache
def test_pack_cache_registry_semantics():
    """autograd_ops._PackCache caches packed operands ONLY for parameters of attached modules (looked up by address,
    verified against the still-alive owner), never for temporaries — whose addresses the allocator recycles."""
    from gcd_amd.autograd_ops import _PackCache
    pc = _PackCache()
    lin = torch.nn.Linear(8, 4)
    calls = []

    def pack(w):
        calls.append(1)
        return w.detach().clone()

    # un-attached: packed every time, nothing kept
    pc.get(lin.weight, "k", pack)
    pc.get(lin.weight, "k", pack)
    assert len(calls) == 2 and not pc._d
    pc.attach(lin)
    a = pc.get(lin.weight, "k", pack)
    b = pc.get(lin.weight, "k", pack)
    assert a is b and len(calls) == 3
    # a detached alias (what non-reentrant checkpointing hands the backward pass) and a reshaped view hit the entry
    assert pc.get(lin.weight.detach(), "k", pack) is a and pc.get(lin.weight.view(2, 16), "k", pack) is a
    # an in-place update bumps the version: repacked
    with torch.no_grad():
        lin.weight.add_(1.0)
    c = pc.get(lin.weight, "k", pack)
    assert c is not a and torch.equal(c, lin.weight)
    # a temporary of the same shape is never cached, and never served from the parameter's entry
    t = torch.zeros(4, 8)
    d = pc.get(t, "k", pack)
    assert torch.equal(d, t) and len(pc._d) == 1
    # several parameters packed together: cached on all of them
    lin2 = torch.nn.Linear(8, 4)
    pc.attach(lin2)
    m1 = pc.get_multi((lin.weight, lin2.weight), "cat", lambda ws: torch.cat(ws, 0))
    m2 = pc.get_multi((lin.weight, lin2.weight), "cat", lambda ws: torch.cat(ws, 0))
    assert m1 is m2 and m1.shape == (8, 8)
    assert pc.get_multi((lin.weight, t), "cat", lambda ws: torch.cat(ws, 0)) is not \
        pc.get_multi((lin.weight, t), "cat", lambda ws: torch.cat(ws, 0))
    # the owner dies: its address is no longer trusted
    ptr = lin2.weight.data_ptr()
    del lin2
    import gc
    gc.collect()
    assert pc._reg[ptr]() is None
    pc.clear()
    assert not pc._d


def test_train_engine_switch_and_planned_engine_refuse_loudly():
    """Host logic of round 5's engine switch: unknown names raise (an environment typo must not silently pick an engine);
    the planned engine refuses a network that is not on a GPU — there is no CPU path — and GradBucketer registers /
    unregisters its listener for the planned engine's in-place gradients."""
    import torch
    from gcd_amd import _lib, autograd_ops as A, train_plan as TP, training as TR
    from gcd_amd.video_model import VideoUNet
    from oracle import svd_unet_ref as O
    old = TR.TRAIN_ENGINE
    try:
        TR.set_train_engine("autograd")
        assert TR.TRAIN_ENGINE == "autograd"
        with pytest.raises(ValueError):
            TR.set_train_engine("planed")
    finally:
        TR.set_train_engine(old)
    with pytest.raises(ValueError):
        A.set_wgrad_impl("trr")
    net = VideoUNet(**O.TINY.as_reference_kwargs())
    with pytest.raises(_lib.GcdError, match="GPU"):
        TP.TrainPlan(net)
    # a bucketer that is not active (single process) registers nothing; close() is idempotent
    b = TR.GradBucketer(net.parameters(), None)
    assert not any("_gcd_grad_listeners" in p.__dict__ for p in net.parameters())
    b.close()
    b.close()
    # listeners hang on the parameters they are for (round 6: no process-wide list); removing the last one removes the slot
    ps = list(net.parameters())[:3]
    seen = []
    TP.add_grad_listener(ps, seen.append)
    TP.add_grad_listener(ps[:1], seen.append)              # registering twice does not call twice
    assert all(p.__dict__["_gcd_grad_listeners"] == [seen.append] for p in ps)
    TP.remove_grad_listener(ps, seen.append)
    assert not any("_gcd_grad_listeners" in p.__dict__ for p in net.parameters())
    with pytest.raises(ValueError):
        TR.TrainDenoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"}, engine="planed")
    # the in-place gradient sink and the fp16 pass-through are per-thread scopes, restored on exit
    assert A._sink() is None
    with A.grad_sink("plan-a"):
        with A.grad_sink("plan-b"):
            assert A._sink() == "plan-b"
        assert A._sink() == "plan-a"
    assert A._sink() is None
    d = A._f16_passthrough_on()
    with A.f16_passthrough(not d):
        assert A._f16_passthrough_on() == (not d)
    assert A._f16_passthrough_on() == d


def test_checkpoint_policy_is_a_memory_decision():
    """TrainPlan.contexts_fit: cfg4's 2 clips (43 008 L0 tokens, 320 channels: 27.5 GB with margin) and 8 clips (110 GB) fit a
    mostly free 288 GB part and are not recomputed; the same 8 clips do not fit an 80 GB part (the reference's hardware) nor
    a 288 GB part that has 60 GB left; 8 clips at 64 x 96 latents (440 GB) never do."""
    from gcd_amd.train_plan import TrainPlan
    GB = 1 << 30
    t2, t8 = 28 * 32 * 48, 112 * 32 * 48
    assert TrainPlan.contexts_fit(t2, 320, 250 * GB) and TrainPlan.contexts_fit(t8, 320, 250 * GB)
    assert TrainPlan.contexts_fit(t2, 320, 60 * GB) and not TrainPlan.contexts_fit(t8, 320, 60 * GB)
    assert not TrainPlan.contexts_fit(t8, 320, 80 * GB - 30 * GB)
    assert not TrainPlan.contexts_fit(112 * 64 * 96, 320, 280 * GB)

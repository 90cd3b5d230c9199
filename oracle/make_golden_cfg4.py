"""ORACLE support (build container only): BASELINE.json cfg4 golden — ONE fine-tune step of the full-width
1.53 B-parameter Kubric VideoUNet at cfg4's per-GPU shape (2 clips x 14 frames, 32 x 48 latents, N = 28, no
CFG; configs/train_kubric_max90.yaml:209-210,234) through the UNMODIFIED reference classes: VideoUNet with
its own activation checkpointing (use_checkpoint True, openaimodel.py:326-329, attention.py:544-546,
video_attention.py:104-105), OpenAIWrapper, Denoiser + VScalingWithEDMcNoise and
StandardDiffusionLoss.get_loss + EDMWeighting (loss.py:163-273), fp32 torch.autograd on the CPU.
Re-run with:  python -m oracle.make_golden_cfg4      (~10 min, ~35 GiB)

  train_kubric_32x48.pt   loss, 65 536 strided samples of the denoiser output, and for every one of the 1432
                          parameters the fp64 norm of its gradient + 128 strided samples of it (< 1 MB)
"""
from __future__ import annotations

import importlib
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_shim, svd_unet_ref as O, weights  # noqa: E402
from oracle.make_golden_fullres import sample            # noqa: E402
from oracle.make_golden_loss import CFG as LOSS_CFG      # noqa: E402

OUT = ROOT / "tests" / "golden"
SALT, SEED = 4, 141
B, T, H, W = 2, 14, 32, 48
STEP = 0                 # global_step 0: plain per-frame mean (the focal top-k schedule starts at 1.0)
# `python -m oracle.make_golden_cfg4 focal`: global_step 2500 instead — the annealed top-fraction ("focal") loss half-way
# through its schedule (keep the 55 % largest per-pixel losses of every frame, 0.9 top + 0.1 mean; loss.py:236-256) ->
# train_kubric_32x48_focal.pt (norms + 32 samples per gradient)


def inputs():
    """Seeded batch shared with tests/test_backward_gpu.py: x0, noise, per-clip sigma, cond."""
    cfg = O.KUBRIC
    g = torch.Generator().manual_seed(SEED)
    BT = B * T
    x0 = torch.randn(BT, 4, H, W, generator=g)
    noise = torch.randn(BT, 4, H, W, generator=g)
    cond = {"crossattn": torch.randn(BT, 1, cfg.context_dim, generator=g),
            "concat": torch.randn(BT, 4, H, W, generator=g) * 0.8,
            "vector": torch.randn(BT, cfg.adm_in_channels + cfg.aux_emb_dim, generator=g).clamp(-1, 1)}
    # EDMSampling(p_mean 1.0, p_std 1.6) harmonised per clip (loss.py:131-136): one noise level per clip
    sig = (1.0 + 1.6 * torch.randn(B, generator=g)).exp().repeat_interleave(T)
    return x0, noise, cond, sig


def main():
    focal = "focal" in sys.argv[1:]
    step = 2500 if focal else STEP
    nsamp = 32 if focal else 128
    fname = "train_kubric_32x48_focal.pt" if focal else "train_kubric_32x48.pt"
    torch.manual_seed(0)
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", str(os.cpu_count()))))
    cfg = O.KUBRIC
    VideoUNet, OpenAIWrapper, Denoiser, _ = ref_shim.reference_classes()
    ref_shim.reference_diffusion_module()           # installs the stubs loss.py needs
    L = importlib.import_module("sgm.modules.diffusionmodules.loss")
    kw = cfg.as_reference_kwargs()
    kw["use_checkpoint"] = True
    with torch.device("meta"):
        net = VideoUNet(**kw)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net = net.to_empty(device="cpu")
    net.load_state_dict(weights.synth_state_dict(shapes, SALT))
    net.train()
    x0, noise, cond, sig = inputs()
    den = Denoiser(ref_shim.DENOISER_CFG)
    model = OpenAIWrapper(net)
    loss_fn = L.StandardDiffusionLoss(**LOSS_CFG)
    noised = x0 + noise * sig[:, None, None, None]
    t0 = time.time()
    out = den(model, noised, sig, cond, num_video_frames=T, image_only_indicator=torch.zeros(B, T))
    out = out.contiguous()      # the CPU convolution returns channels_last strides; get_loss uses .view (loss.py:244)
    w = loss_fn.loss_weighting(sig)[:, None, None, None]
    loss = loss_fn.get_loss(out, x0, w, {"global_step": step}).mean()
    t1 = time.time()
    print(f"forward {t1 - t0:.0f} s, loss {float(loss):.6f}", flush=True)
    loss.backward()
    print(f"backward {time.time() - t1:.0f} s", flush=True)
    norms, samples, dead = {}, {}, []
    for name, p in net.named_parameters():
        if p.grad is None or float(p.grad.abs().max()) == 0.0:
            dead.append(name)
            continue
        norms[name] = float(p.grad.double().norm())
        samples[name] = sample(p.grad, nsamp)
    torch.save({"config": "KUBRIC", "salt": SALT, "seed": SEED, "B": B, "T": T, "H": H, "W": W, "step": step,
                "loss": float(loss), "out_samples": sample(out.detach(), 65536),
                "out_norm": float(out.detach().double().norm()),
                "grad_norms": norms, "grad_samples": samples, "dead": dead,
                "reference_cpu_seconds": time.time() - t0,
                "reference_cpu_threads": torch.get_num_threads()}, OUT / fname)
    print(f"{len(norms)} gradients, {len(dead)} exactly-zero parameters; wrote {fname}", flush=True)


if __name__ == "__main__":
    main()

"""ORACLE support (build container only): sampling-LOOP goldens at the metric's own resolution, made by
the UNMODIFIED reference plugin stack (EulerEDMSampler + LinearPredictionGuider + Denoiser +
OpenAIWrapper + VideoUNet through oracle/ref_shim.py) with the full-width 1.53 B-parameter networks on
the procedural weights of oracle/weights.py.  Hours of fp32 CPU time (≈200-300 s per 72x128 step on 8
cores, ≈16 GiB RSS), so every job checkpoints its state under /tmp and resumes.

  python -m oracle.make_golden_loop72 cfg1      BASELINE.json cfg1: Kubric net, the full 25-step loop at
                                                14 x 72 x 128                  -> loop_kubric_72x128.pt
  python -m oracle.make_golden_loop72 cfg3mid   cfg3: ParDom net, the full 50-step loop at 14 x 40 x 64
                                                (latent sides must be multiples of 8: three stride-2 levels)
                                                                               -> loop_pardom_40x64.pt
  python -m oracle.make_golden_loop72 cfg3tail  cfg3: ParDom net at 14 x 72 x 128, the LAST 15 of the 50
                                                steps (sigma_35 = 1.17 .. 0: where the network term carries
                                                the result) from a seeded mid-trajectory state
                                                x_35 = n * sqrt(1 + sigma_35^2)  (the marginal of the EDM
                                                forward process on unit-variance data)
                                                                               -> loop_pardom_72x128_tail.pt
  python -m oracle.make_golden_loop72 cfg3      cfg3 in full: ParDom net, all 50 steps at 14 x 72 x 128 (~3 h)
                                                                               -> loop_pardom_72x128.pt
  python -m oracle.make_golden_loop72 cfg1b2mid num_samples = 2 (scripts/test.py:326): TWO clips in one sampler call
                                                (2 x 14 frames, 56 under CFG), Kubric net, the full 25-step loop at
                                                40 x 64                        -> loop_kubric_b2_40x64.pt
  python -m oracle.make_golden_loop72 cfg1b2    the same at 72 x 128 (~4-5 h)  -> loop_kubric_b2_72x128.pt

Fixtures hold 65 536 strided samples + the norm of x after the KEEP steps and the full final latents
(2 MB, needed to decode frames for the PSNR check).  The sampling grid is make_golden_fullres.sample.
"""
from __future__ import annotations

import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_shim, svd_unet_ref as O, weights  # noqa: E402
from oracle.make_golden_fullres import sample            # noqa: E402

OUT = ROOT / "tests" / "golden"

JOBS = {
    #            config    salt seed  h   w   steps first keep                              file
    "cfg1":     ("KUBRIC", 2,   181, 72, 128, 25,   0,    (1, 5, 10, 15, 20, 25),           "loop_kubric_72x128.pt"),
    "cfg3mid":  ("PARDOM", 3,   193, 40, 64,  50,   0,    (1, 10, 20, 30, 40, 45, 50),      "loop_pardom_40x64.pt"),
    "cfg3tail": ("PARDOM", 3,   194, 72, 128, 50,   35,   (36, 38, 40, 42, 44, 46, 48, 50), "loop_pardom_72x128_tail.pt"),
    "cfg3":     ("PARDOM", 3,   195, 72, 128, 50,   0,    (1, 10, 20, 30, 40, 45, 50),      "loop_pardom_72x128.pt"),
    # num_samples = 2 (scripts/test.py:326): TWO clips in one sampler call, 56 frames under CFG
    "cfg1b2mid": ("KUBRIC", 2,  201, 40, 64,  25,   0,    (1, 5, 10, 15, 20, 25),           "loop_kubric_b2_40x64.pt"),
    "cfg1b2":   ("KUBRIC", 2,   202, 72, 128, 25,   0,    (1, 5, 10, 15, 20, 25),           "loop_kubric_b2_72x128.pt"),
}
CLIPS = {"cfg1b2mid": 2, "cfg1b2": 2}


def main(job: str):
    cfg_name, salt, seed, h, w, steps, first, keep, fname = JOBS[job]
    torch.manual_seed(0)
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", str(os.cpu_count()))))
    cfg = getattr(O, cfg_name)
    VideoUNet, OpenAIWrapper, Denoiser, EulerEDMSampler = ref_shim.reference_classes()
    with torch.device("meta"):
        net = VideoUNet(**cfg.as_reference_kwargs())
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net = net.to_empty(device="cpu")
    net.load_state_dict(weights.synth_state_dict(shapes, salt))
    net.eval()
    T = 14
    B = CLIPS.get(job, 1)
    noise, c, uc = weights.synth_inputs(B, T, h, w, cfg.context_dim,
                                        cfg.adm_in_channels + cfg.aux_emb_dim, seed)
    sampler = EulerEDMSampler(num_steps=steps, device="cpu", **ref_shim.SAMPLER_CFG)
    den = Denoiser(ref_shim.DENOISER_CFG)
    model = OpenAIWrapper(net)
    extra = {"num_video_frames": T, "image_only_indicator": torch.zeros(2 * B, T)}

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)

    # the reference's own loop prologue (sampling.py:46-59), then sampler_step per i (sampling.py:128-142)
    x, s_in, sigmas, num_sigmas, cond, ucond = sampler.prepare_sampling_loop(noise.clone(), c, uc, steps)
    assert num_sigmas == steps + 1
    if first:
        # seeded mid-trajectory state (see the module docstring); prepare_sampling_loop scaled by sigma_0
        x = noise * float((1.0 + sigmas[first] ** 2) ** 0.5)
    state = Path(f"/tmp/golden_loop72_{job}.pt")
    done, trace, secs = first, {}, []
    if state.exists():
        st = torch.load(state)
        x, done, trace, secs = st["x"], st["done"], st["trace"], st["secs"]
        print(f"[{job}] resuming after step {done}", flush=True)
    for i in range(done, steps):
        t0 = time.time()
        with torch.no_grad():
            # gamma = 0: s_churn = 0 in every GCD config (sampling.py:129-133)
            x = sampler.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, ucond, 0.0)
        secs.append(time.time() - t0)
        if i + 1 in keep:
            trace[i + 1] = {"samples": sample(x, 65536), "norm": float(x.double().norm())}
        torch.save({"x": x, "done": i + 1, "trace": trace, "secs": secs}, state)
        print(f"[{job}] step {i + 1}/{steps}: {secs[-1]:.0f} s, sigma {float(sigmas[i]):.4f} -> "
              f"{float(sigmas[i + 1]):.4f}, std {float(x.std()):.4f}", flush=True)
    torch.save({"config": cfg_name, "salt": salt, "T": T, "clips": B, "h": h, "w": w, "steps": steps,
                "first_step": first, "input_seed": seed, "sigmas": sigmas.clone(),
                "trace": trace, "final": x.clone(), "final_norm": float(x.double().norm()),
                "reference_cpu_seconds_per_step": secs,
                "reference_cpu_threads": torch.get_num_threads()}, OUT / fname)
    print(f"[{job}] wrote {OUT / fname}", flush=True)


if __name__ == "__main__":
    for j in sys.argv[1:]:
        main(j)

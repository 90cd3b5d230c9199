"""tools/clip_bench.py — one whole clip on one MI355X: 25-step EulerEDM + CFG sampling loop on the HIP
VideoUNet (14 x 72 x 128 latents) followed by the HIP first-stage decode to 14 frames of 576 x 1024,
i.e. what DiffusionEngine.sample_video + decode_first_stage do per clip after the conditioner.

    python tools/clip_bench.py [--steps 25] [--clips 2] [--json out.json]

Random-init weights of the real architectures (no checkpoints offline), synthetic conditioning."""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--json", type=str, default="")
    a = ap.parse_args()
    import bench
    from gcd_amd.denoiser import Denoiser
    from gcd_amd.first_stage import decode_first_stage
    from gcd_amd.sampling import EulerEDMSampler, FusedDenoiser
    from gcd_amd.temporal_ae import VideoDecoder
    from gcd_amd.wrappers import OpenAIWrapper
    dev = torch.device("cuda:0")
    T, h, w = 14, 72, 128
    net = bench.build_model(dev)
    torch.manual_seed(1)
    dec = VideoDecoder(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3,
                       out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[],
                       dropout=0.0, video_kernel_size=[3, 1, 1])
    with torch.no_grad():
        for p in dec.parameters():
            if p.dim() > 1 and float(p.abs().max()) == 0.0:
                p.normal_(0, p[0].numel() ** -0.5)
    dec = dec.to(dev).eval()
    sampler = EulerEDMSampler(
        discretization_config={"target": "gcd_amd.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
        num_steps=a.steps,
        guider_config={"target": "gcd_amd.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": 1.5, "min_scale": 1.0}},
        device="cuda")
    fd = FusedDenoiser(Denoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"}),
                       OpenAIWrapper(net), num_video_frames=T,
                       image_only_indicator=torch.zeros(2, T, device=dev))
    times = []
    for clip in range(a.clips + 1):                      # clip 0 = warm-up (packing, graph capture)
        noise, c, uc = bench.synth_inputs(dev, T, h, w, seed=200 + clip)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z = sampler(fd, noise, cond=c, uc=uc)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        frames = decode_first_stage(dec, z * 0.18215 / max(float(z.std()), 1e-6), 0.18215,
                                    en_and_decode_n_samples_a_time=T)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert frames.shape == (T, 3, 8 * h, 8 * w) and bool(torch.isfinite(frames).all())
        if clip:
            times.append((t1 - t0, t2 - t1))
    loop = sorted(t[0] for t in times)[len(times) // 2]
    decd = sorted(t[1] for t in times)[len(times) // 2]
    res = dict(workload=f"{a.steps}-step EulerEDM + CFG loop at 14x72x128 latents + decode to 14x3x576x1024",
               loop_s=round(loop, 3), decode_s=round(decd, 3), clip_s=round(loop + decd, 3),
               clips_per_s=round(1.0 / (loop + decd), 4), frames_per_s=round(T / (loop + decd), 2),
               sampler_path=sampler.last_path)
    print(json.dumps(res))
    if a.json:
        Path(a.json).parent.mkdir(parents=True, exist_ok=True)
        Path(a.json).write_text(json.dumps(res))


if __name__ == "__main__":
    main()

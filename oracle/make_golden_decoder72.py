"""ORACLE support (build container only): first-stage DECODER golden at the metric's own resolution, made by the
UNMODIFIED reference `VideoDecoder` (sgm/modules/autoencoding/temporal_ae.py:293-349 on Decoder,
sgm/modules/diffusionmodules/model.py:604-748) in fp32 on the CPU, called the way
`DiffusionEngine.decode_first_stage` calls it (sgm/models/diffusion.py:233-251: z / scale_factor, one chunk of
`en_and_decode_n_samples_a_time` = 14 frames, `timesteps` = the chunk length).

  python -m oracle.make_golden_decoder72          -> tests/golden/decoder_kubric_72x128.pt  (~1.4 MB)

Input: the FINAL latents of the reference's own 25-step cfg1 loop (tests/golden/loop_kubric_72x128.pt, 14 x 4 x 72 x 128),
so the frames in this fixture are what the unmodified reference stack produces end to end (loop + decode) on the
procedural weights: full 128-channel decoder config (oracle/vae_decoder_ref.KUBRIC), weights salt 1.
Output: 262 144 strided samples + the norm of the 14 x 3 x 576 x 1024 frames, 65 536 samples + norm of conv_in, the
three mid blocks, the last block and the Upsample of every up level (the 72x128 mid attention has S = 9216; the level-0
GroupNorms run over 8.3 M tokens).  Sampling grid: make_golden_fullres.sample (float64 index grid).
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import vae_decoder_ref as D, weights              # noqa: E402
from oracle.make_golden_decoder import reference_decoder_class  # noqa: E402
from oracle.make_golden_fullres import sample                 # noqa: E402

OUT = ROOT / "tests" / "golden"
SCALE_FACTOR = 0.18215                                        # configs/infer_kubric.yaml:6


def main():
    torch.manual_seed(0)
    loop = torch.load(OUT / "loop_kubric_72x128.pt")
    z = loop["final"].float() / SCALE_FACTOR                  # diffusion.py:235
    T = z.shape[0]
    assert tuple(z.shape) == (14, 4, 72, 128)
    VideoDecoder = reference_decoder_class()
    dec = VideoDecoder(**D.KUBRIC.as_reference_kwargs()).eval()
    shapes = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    dec.load_state_dict(weights.synth_state_dict(shapes, salt=1))
    taps = {}

    def keep(name):
        def hook(m, i, o):
            taps[name] = {"samples": sample(o.detach(), 65536), "norm": float(o.detach().double().norm()),
                          "shape": tuple(o.shape)}
            print(f"  {name}: {tuple(o.shape)} at {time.time() - t0:.0f} s", flush=True)
        return hook

    hooks = [dec.conv_in.register_forward_hook(keep("conv_in"))]
    for nm in ("block_1", "attn_1", "block_2"):
        hooks.append(getattr(dec.mid, nm).register_forward_hook(keep(f"mid.{nm}")))
    for lv, up in enumerate(dec.up):
        hooks.append(up.block[-1].register_forward_hook(keep(f"up.{lv}.block.{len(up.block) - 1}")))
        if hasattr(up, "upsample"):
            hooks.append(up.upsample.register_forward_hook(keep(f"up.{lv}.upsample")))
    t0 = time.time()
    with torch.no_grad():
        out = dec(z, timesteps=T)                             # diffusion.py:243-247
    secs = time.time() - t0
    for hk in hooks:
        hk.remove()
    assert tuple(out.shape) == (14, 3, 576, 1024)
    torch.save({
        "config": "KUBRIC", "T": T, "h": 72, "w": 128, "weight_salt": 1, "scale_factor": SCALE_FACTOR,
        "latents_from": "loop_kubric_72x128.pt['final']",
        "out_samples": sample(out, 262144), "out_norm": float(out.double().norm()),
        "out_mean": float(out.double().mean()), "out_shape": tuple(out.shape),
        "taps": taps, "reference_cpu_seconds": secs, "reference_cpu_threads": torch.get_num_threads(),
    }, OUT / "decoder_kubric_72x128.pt")
    print(f"decoder_kubric_72x128: out {tuple(out.shape)} std {float(out.std()):.4f} in {secs:.0f} s, taps {len(taps)}")


if __name__ == "__main__":
    main()

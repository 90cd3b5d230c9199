"""tools/stress_diag.py — per-block rel-L2 of the HIP forward against the heavy-tailed / GEGLU-gain stress fixtures
(oracle/make_golden_stress.py and its GOLDEN_STRESS_* diagnostic variants under tests/golden/dbg_*.pt)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from oracle import svd_unet_ref as O, weights  # noqa: E402
from gcd_amd.video_model import VideoUNet      # noqa: E402


def rel(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm())


def main():
    gpu = torch.device("cuda:0")
    for f in sys.argv[1:]:
        g = torch.load(ROOT / "tests" / "golden" / f)
        with torch.device("meta"):
            net = VideoUNet(**O.TINY.as_reference_kwargs())
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net = net.to_empty(device=gpu)
        net.load_state_dict(weights.synth_state_dict_heavy(shapes, g["salt"], g["nu"], g["geglu_gain"]))
        net.eval()
        noise, c, uc = weights.synth_inputs(1, g["T"], g["h"], g["w"], O.TINY.context_dim,
                                            O.TINY.adm_in_channels + O.TINY.aux_emb_dim, g["input_seed"])
        x = torch.cat([torch.cat([noise, uc["concat"]], 1), torch.cat([noise, c["concat"]], 1)])
        ts = torch.linspace(-1.5, 1.63, 2 * g["T"])
        ctx = torch.cat([uc["crossattn"], c["crossattn"]])
        y = torch.cat([uc["vector"], c["vector"]])
        net.engine.taps = {}
        out = net(x.to(gpu), ts.to(gpu), context=ctx.to(gpu), y=y.to(gpu), num_video_frames=g["T"],
                  image_only_indicator=torch.zeros(2, g["T"], device=gpu))
        torch.cuda.synchronize()
        taps, net.engine.taps = net.engine.taps, None
        errs = {}
        for k, v in taps.items():
            fl = v.reshape(-1).cpu()
            idx = torch.linspace(0, fl.numel() - 1, min(4096, fl.numel())).long()
            errs[k] = rel(fl[idx], g["tap_samples"][k])
        print(f"{f}: nu {g['nu']} gain {g['geglu_gain']} hidden absmax {g['geglu_hidden_absmax']:.0f} | out {rel(out, g['out']):.2e} | "
              + " ".join(f"{k.replace('input_blocks', 'i').replace('output_blocks', 'o').replace('middle_block', 'm')}:{e:.1e}"
                         for k, e in errs.items()), flush=True)
        del net


if __name__ == "__main__":
    main()

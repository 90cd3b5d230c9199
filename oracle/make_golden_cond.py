"""ORACLE support (build container only): golden for SURVEY.md §8a a22 — the conditioning producers —
made by the UNMODIFIED reference classes of sgm/modules/encoders/modules.py (SphericalEmbedder
:247-287, CameraEmbedder :231-244, ConcatTimestepEmbedderND :1000-1016, IdentityEncoder,
GeneralConditioner :84-208 incl. get_unconditional_conditioning) and sgm/data/common.py
(construct_trajectory :450-479).  The module imports kornia / open_clip at file scope; both are
stubbed with empty modules (none of the classes used here touches them).
Re-run with:  python -m oracle.make_golden_cond   ->  tests/golden/cond_tiny.pt
"""
from __future__ import annotations

import importlib
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_shim, weights  # noqa: E402

OUT = ROOT / "tests" / "golden"
T = 14


def reference_encoder_module():
    ref_shim.install()
    for name in ("kornia", "open_clip"):
        sys.modules.setdefault(name, types.ModuleType(name))
    for name, sub in [("sgm.modules.encoders", "sgm/modules/encoders"),
                      ("sgm.modules.autoencoding", "sgm/modules/autoencoding"),
                      ("sgm.modules.distributions", "sgm/modules/distributions")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [str(ref_shim.REF / sub)]
            m.__package__ = name
            sys.modules[name] = m
    return importlib.import_module("sgm.modules.encoders.modules")


def reference_construct_trajectory():
    """sgm/data/common.py imports cv2 & friends at file scope: load the one function by exec'ing its
    source lines from the read-only tree (nothing is copied into the repo)."""
    src = (ref_shim.REF / "sgm" / "data" / "common.py").read_text().splitlines()
    a = next(i for i, l in enumerate(src) if l.startswith("def construct_trajectory"))
    b = next((i for i in range(a + 1, len(src)) if src[i].startswith(("def ", "class "))), len(src))
    ns = {"np": np}
    exec("\n".join(src[a:b]), ns)
    return ns["construct_trajectory"]


def inputs():
    g = torch.Generator().manual_seed(17)
    return {
        "fps_id": torch.full((T,), 12.0),
        "motion_bucket_id": torch.full((T,), 127.0),
        "cond_aug": torch.full((T,), 0.02),
        "cond_frames_without_noise": torch.randn(T, 1, 64, generator=g),      # stands in for the CLIP token
        "cond_frames": torch.randn(T, 4, 8, 8, generator=g),                  # stands in for the VAE latents
        "scaled_relative_pose": torch.randn(T, 3, 4, generator=g),
    }


EMB = "sgm.modules.encoders.modules."
EMBEDDERS = [
    dict(input_key="fps_id", is_trainable=False, target=EMB + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
    dict(input_key="motion_bucket_id", is_trainable=True, target=EMB + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
    dict(input_key="cond_frames_without_noise", is_trainable=False, target=EMB + "IdentityEncoder"),
    dict(input_key="cond_frames", is_trainable=False, target=EMB + "IdentityEncoder"),
    dict(input_key="cond_aug", is_trainable=False, target=EMB + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
    dict(input_key="scaled_relative_angles", is_trainable=True, target=EMB + "SphericalEmbedder",
         params=dict(embed_dim=128, zero_init=False)),
]


def main():
    mod = reference_encoder_module()
    traj = reference_construct_trajectory()
    batch = inputs()
    out = {"T": T, "trajectories": {}}
    for name, (end, kind, mt) in {"gradual_linear": ((30.0, 15.0, 1.0), "interpol_linear", 13),
                                  "gradual_sine": ((-90.0, 40.0, -2.5), "interpol_sine", 13),
                                  "direct": ((60.0, -10.0, 0.5), "interpol_linear", 0)}.items():
        src, dst = traj(np.zeros(3, np.float32), np.array(end, np.float32), kind, T, mt)
        rel = dst - src
        rel[:, 0] *= np.pi / 180.0
        rel[:, 1] *= np.pi / 180.0
        out["trajectories"][name] = dict(end=end, kind=kind, move_time=mt, src=torch.tensor(src),
                                         dst=torch.tensor(dst), rel=torch.tensor(rel, dtype=torch.float32))
    batch["scaled_relative_angles"] = out["trajectories"]["gradual_linear"]["rel"]

    cond = mod.GeneralConditioner(EMBEDDERS)
    sph = cond.embedders[5]
    sph.proj.weight.data.copy_(weights.synth_tensor("conditioner.embedders.5.proj.weight", (128, 13)))
    sph.proj.bias.data.copy_(weights.synth_tensor("conditioner.embedders.5.proj.bias", (128,)))
    cam = mod.CameraEmbedder(embed_dim=128)
    cam.proj.weight.data.copy_(weights.synth_tensor("camera.proj.weight", (128, 12)))
    cam.proj.bias.data.copy_(weights.synth_tensor("camera.proj.bias", (128,)))
    with torch.no_grad():
        c, uc = cond.get_unconditional_conditioning(
            batch, batch_uc=batch, force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"])
        out["spherical"] = {k: sph(v["rel"]) for k, v in out["trajectories"].items()}
        out["camera"] = cam(batch["scaled_relative_pose"])
        out["timestep_nd"] = mod.ConcatTimestepEmbedderND(256)(
            torch.stack([batch["fps_id"], batch["motion_bucket_id"], batch["cond_aug"]], 1))
    out["c"], out["uc"] = dict(c), dict(uc)
    out["state_dict_keys"] = {k: tuple(v.shape) for k, v in cond.state_dict().items()}
    OUT.mkdir(parents=True, exist_ok=True)
    torch.save(out, OUT / "cond_tiny.pt")
    print({k: tuple(v.shape) for k, v in c.items()}, list(out["state_dict_keys"]))


if __name__ == "__main__":
    main()

"""ORACLE support (build container only): import the *unmodified* reference hot-path modules from
/root/reference without executing sgm/__init__.py (which pulls in Lightning, kornia, open_clip...).

Recipe from SURVEY.md Appendix C.  /root/reference does not exist on the GPU box, so nothing that
runs there (tests -m gpu, smoke(), bench.py) may import this file.
"""
from __future__ import annotations

import sys
import types
from pathlib import Path

REF = Path("/root/reference/gcd-model")


def available() -> bool:
    return (REF / "sgm" / "modules" / "diffusionmodules" / "video_model.py").exists()


def install() -> None:
    if "sgm" in sys.modules and getattr(sys.modules["sgm"], "_gcd_shim", False):
        return
    if not available():
        raise RuntimeError("/root/reference is not mounted: the reference oracle is unavailable here")
    for name, sub in [("sgm", "sgm"), ("sgm.modules", "sgm/modules"),
                      ("sgm.modules.diffusionmodules", "sgm/modules/diffusionmodules")]:
        m = types.ModuleType(name)
        m.__path__ = [str(REF / sub)]
        m.__package__ = name
        m._gcd_shim = True
        sys.modules[name] = m
    if "omegaconf" not in sys.modules:     # type hints only (sampling.py:9)
        oc = types.ModuleType("omegaconf")
        oc.ListConfig = list
        oc.OmegaConf = dict
        sys.modules["omegaconf"] = oc


def reference_classes():
    install()
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    return VideoUNet, OpenAIWrapper, Denoiser, EulerEDMSampler


SAMPLER_CFG = dict(
    discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization",
                           "params": {"sigma_max": 700.0}},
    guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                   "params": {"num_frames": 14, "max_scale": 1.5, "min_scale": 1.0}},
)
DENOISER_CFG = {"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}


def _bare(name: str, sub: str):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = [str(REF / sub)]
        m.__package__ = name
        m._gcd_shim = True
        sys.modules[name] = m
    return sys.modules[name]


def _stub(name: str, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m._gcd_stub = True
        sys.modules[name] = m
    for k, v in attrs.items():
        if not hasattr(m, k):
            setattr(m, k, v)
    return m


def reference_diffusion_module():
    """Import the UNMODIFIED sgm/models/diffusion.py (the caller of the hot path: DiffusionEngine
    .sample_video, diffusion.py:504-577) with its unavailable third-party imports stubbed by empty
    modules: pytorch_lightning (LightningModule := nn.Module), peft, lovely_tensors, lovely_numpy,
    skimage, kornia, open_clip, and the LPIPS module (which needs torchvision)."""
    install()
    import torch.nn as nn
    for name, sub in [("sgm.models", "sgm/models"), ("sgm.modules.encoders", "sgm/modules/encoders"),
                      ("sgm.modules.autoencoding", "sgm/modules/autoencoding"),
                      ("sgm.modules.distributions", "sgm/modules/distributions"),
                      ("sgm.modules.autoencoding.lpips", "sgm/modules/autoencoding/lpips"),
                      ("sgm.modules.autoencoding.lpips.loss", "sgm/modules/autoencoding/lpips/loss")]:
        _bare(name, sub)
    _stub("kornia")
    _stub("open_clip")
    _stub("pytorch_lightning", LightningModule=nn.Module)
    _stub("peft")
    _stub("peft.tuners")
    _stub("peft.tuners.lora", layer=types.SimpleNamespace())
    _stub("lovely_tensors", monkey_patch=lambda: None)
    _stub("lovely_numpy", lo=lambda x: x)
    _stub("skimage", metrics=types.SimpleNamespace())
    _stub("skimage.metrics")
    _stub("sgm.modules.autoencoding.lpips.loss.lpips", LPIPS=nn.Identity)
    import importlib
    enc = importlib.import_module("sgm.modules.encoders.modules")
    sys.modules["sgm.modules"].GeneralConditioner = enc.GeneralConditioner
    sys.modules["sgm.modules"].UNCONDITIONAL_CONFIG = {
        "target": "sgm.modules.GeneralConditioner", "params": {"emb_models": []}}
    return importlib.import_module("sgm.models.diffusion")

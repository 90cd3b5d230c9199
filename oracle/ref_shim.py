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

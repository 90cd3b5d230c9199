"""ORACLE support (test infrastructure): procedural, machine-independent weights and inputs.

A random-init VideoUNet outputs exactly 0 (61 zero_module tensors), and no
checkpoint is available, so parity needs synthetic non-degenerate weights.  Every tensor is drawn
from its own generator seeded by crc32(name), so the reference modules (oracle/make_golden.py), the
oracle restatement and the HIP product all see bit-identical fp32 weights on any machine without
shipping a state_dict.
"""
from __future__ import annotations

import zlib
from typing import Dict

import torch


def _gen(name: str, salt: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * salt) % (2 ** 31))


def synth_tensor(name: str, shape, salt: int = 0) -> torch.Tensor:
    g = _gen(name, salt)
    shape = tuple(shape)
    if name.endswith("mix_factor"):
        # logits in [-1, 1.5] -> alpha = sigmoid in (0.27, 0.82), different per blender
        return torch.rand(shape, generator=g) * 2.5 - 1.0
    if name.endswith(".weight") and len(shape) == 1:      # norm scales
        return 1.0 + 0.2 * torch.randn(shape, generator=g)
    if name.endswith(".bias"):
        return 0.1 * torch.randn(shape, generator=g)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) / (fan_in ** 0.5)


def synth_state_dict(shapes: Dict[str, tuple], salt: int = 0) -> Dict[str, torch.Tensor]:
    """shapes: name -> shape (e.g. from a module's state_dict()).  fp32 CPU tensors."""
    return {k: synth_tensor(k, v, salt) for k, v in shapes.items()}


def synth_inputs(batch_clips: int, T: int, h: int, w: int, context_dim: int, vector_dim: int,
                 seed: int = 2):
    """Synthetic sampler inputs in the format DiffusionEngine.sample_video hands the sampler
    (diffusion.py:522-543): returns (noise, c, uc)."""
    g = torch.Generator().manual_seed(seed)
    n = batch_clips * T
    noise = torch.randn(n, 4, h, w, generator=g)
    c = {
        "crossattn": torch.randn(n, 1, context_dim, generator=g),
        "concat": torch.randn(n, 4, h, w, generator=g) * 0.8,
        "vector": torch.randn(n, vector_dim, generator=g).clamp(-1, 1),
    }
    uc = {"crossattn": torch.zeros_like(c["crossattn"]), "concat": torch.zeros_like(c["concat"]),
          "vector": c["vector"].clone()}
    return noise, c, uc


def synth_tensor_heavy(name: str, shape, salt: int = 0, nu: float = 3.0, geglu_gain: float = 1.0) -> torch.Tensor:
    """Stress variant of synth_tensor: matrix / conv weights drawn from a Student-t with `nu` degrees of freedom (heavy
    tails: single weights tens of sigma out), scaled to the variance synth_tensor uses; `geglu_gain` multiplies the
    GEGLU projections (`ff.net.0.proj.*`) so that the fp16 hidden tensor value * gelu(gate) is pushed towards the top
    of the fp16 range.  Norm scales, biases and blend logits as in synth_tensor."""
    shape = tuple(shape)
    if name.endswith("mix_factor") or (name.endswith(".weight") and len(shape) == 1):
        return synth_tensor(name, shape, salt)
    g = _gen(name, salt)
    if name.endswith(".bias"):
        b = 0.1 * torch.randn(shape, generator=g)
        return b * geglu_gain if ".ff.net.0.proj." in name else b
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    z = torch.randn(shape, generator=g)
    if nu >= 1e6:                                   # (diagnostic: Gaussian weights, only the GEGLU gain)
        t = z / (fan_in ** 0.5)
    else:
        chi = sum(torch.randn(shape, generator=g) ** 2 for _ in range(int(nu)))
        t = z / (chi / nu).sqrt()                   # Student-t(nu): variance nu / (nu - 2)
        t = t / (nu / (nu - 2.0)) ** 0.5 / (fan_in ** 0.5)
    return t * geglu_gain if ".ff.net.0.proj." in name else t


def synth_state_dict_heavy(shapes: Dict[str, tuple], salt: int = 0, nu: float = 3.0,
                           geglu_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor_heavy(k, v, salt, nu, geglu_gain) for k, v in shapes.items()}

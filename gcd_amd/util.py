"""Plugin plumbing shared by the drop-in classes: mirrors the helpers of sgm/util.py that the hot
path uses (reference gcd-model/sgm/util.py:168-199) so that YAML `target:` strings can point into
`gcd_amd.*` exactly the way they point into `sgm.*`.
"""
from __future__ import annotations

import importlib

import torch


def get_obj_from_str(string: str, reload: bool = False):
    """'pkg.module.Class' -> the object (sgm/util.py:178-185)."""
    module, cls = string.rsplit(".", 1)
    mod = importlib.import_module(module)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config):
    """{'target': dotted path, 'params': {...}} -> instance (sgm/util.py:168-175)."""
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def append_dims(x: torch.Tensor, target_dims: int) -> torch.Tensor:
    """Append singleton dims until x has target_dims dims (sgm/util.py:192-199)."""
    extra = target_dims - x.ndim
    if extra < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * extra]


def append_zero(x: torch.Tensor) -> torch.Tensor:
    """sgm/util.py:188-189."""
    return torch.cat([x, x.new_zeros([1])])


def default(val, d):
    return val if val is not None else d

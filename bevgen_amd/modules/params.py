"""Parameter containers that reproduce the reference modules' ``state_dict`` key names without re-implementing their Python
forward passes: the arithmetic lives in libbevgen_hip, these modules only hold (and upload) the tensors."""
from __future__ import annotations

from typing import Iterable, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn as nn

BUFFER_LEAVES = ("beta", "bev_grid", "master_layout")


class ParamNode(nn.Module):
    """A bare module whose children/parameters are attached by dotted name."""


def attach(root: nn.Module, name: str, shape: Sequence[int], *, dtype=torch.float32, buffer: Optional[bool] = None, init: Optional[torch.Tensor] = None) -> None:
    parts = name.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, ParamNode())
        node = node._modules[p]
    leaf = parts[-1]
    t = init if init is not None else torch.zeros(tuple(shape), dtype=dtype)
    is_buffer = (leaf in BUFFER_LEAVES) if buffer is None else buffer
    if is_buffer:
        node.register_buffer(leaf, t)
    else:
        node.register_parameter(leaf, nn.Parameter(t, requires_grad=False))


def build_tree(root: nn.Module, shapes: Mapping[str, Tuple[int, ...]], *, int_leaves: Iterable[str] = ("master_layout",)) -> nn.Module:
    for name, shape in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        attach(root, name, shape, dtype=torch.int64 if leaf in int_leaves else torch.float32)
    return root


def module_device(m: nn.Module) -> torch.device:
    for p in m.parameters():
        return p.device
    for b in m.buffers():
        return b.device
    return torch.device("cpu")

"""Checkpoint loading with the reference's semantics (multi_view_generation/utils/general.py:119-160, ``init_from_ckpt``):
``torch.load`` a file, unwrap ``['state_dict']``, strip DeepSpeed's ``_forward_module.`` prefix, drop ``ignore_keys`` (substring match),
report missing / unexpected keys, ``load_state_dict(strict=False)``.  DeepSpeed ZeRO checkpoint *directories* need deepspeed's
converter (general.py:81-116) and are rejected with a clear message."""
from __future__ import annotations

import logging
import os
from typing import Sequence

import torch

from .weights import strip_prefixes

log = logging.getLogger(__name__)


def init_from_ckpt(module, path, ignore_keys: Sequence[str] = (), unfrozen_keys: Sequence[str] = (), strict: bool = False):
    if os.path.isdir(path):
        raise NotImplementedError(f"{path} is a DeepSpeed ZeRO checkpoint directory; convert it to a single file with deepspeed's zero_to_fp32 first")
    # Lightning / DeepSpeed checkpoints pickle hyper-parameters and callback state next to the tensors: weights_only=True (the default since torch 2.6)
    # rejects them.  The released checkpoints are trusted input, exactly as in the reference (general.py:124).
    sd = torch.load(path, map_location="cpu", weights_only=False)
    sd = strip_prefixes(sd, ignore_keys)
    own = module.state_dict()
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    for k in missing:
        print(f"Missing {k}")
    for k in unexpected:
        print(f"Unexpected {k}")
    module.load_state_dict(sd, strict=strict)
    if unfrozen_keys:   # general.py:150-158 re-enables gradients for these parameters: meaningless on an inference-only path, reported instead of silently dropped
        log.info("unfrozen_keys %s ignored: this package is inference-only", list(unfrozen_keys))
    log.info("Restored from %s", path)
    return missing, unexpected

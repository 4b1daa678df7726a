"""Loss stand-ins needed to construct the stage-1 models for sampling (the sampling path never evaluates a loss)."""
import torch.nn as nn


class DummyLoss(nn.Module):
    """Drop-in for modules/losses/vqperceptual.py `DummyLoss` (the `lossconfig` target of every shipped stage-2 config)."""

    def __init__(self, *args, **kwargs):
        super().__init__()

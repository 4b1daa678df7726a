"""Import shim for the *Python reference* (build-container only).

TEST INFRASTRUCTURE - never imported by the product package `bevgen_amd`.

The reference (alexanderswerdlow/BEVGen, mounted read-only at /root/reference) needs
14 third-party packages that are absent from this image.  None of
them contributes arithmetic on the stage-2 sampling path, so we install empty
`sys.modules` stand-ins that expose only the attribute surface the reference touches
at import time, and then import the reference modules from where they lie.

Nothing from /root/reference is copied: this file only makes `import
multi_view_generation...` succeed so that `make_golden.py` can run the reference and
dump input/output vectors under tests/golden/.  /root/reference does not exist on the
GPU box, and nothing under tests -m gpu / smoke() / bench.py imports this module.
"""
from __future__ import annotations

import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("BEVGEN_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "multi_view_generation"))


def _mod(name: str, **attrs) -> types.ModuleType:
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []  # behave like a package so sub-imports resolve
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(_mod(parent), child, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


class _Anything:
    """Callable/attribute sink for names that are only referenced, never used for math."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, item):
        return _Anything()


def _identity_decorator(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]
    return lambda fn: fn


class _LightningModule(nn.Module):
    global_rank = 0
    trainer = None
    logger = None

    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass

    def log_dict(self, *a, **k):
        pass

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")


class _SparsityConfig:
    """Stand-in for deepspeed.ops.sparse_attention.SparsityConfig (stores 3 attrs)."""

    def __init__(self, num_heads, block=16, different_layout_per_head=False):
        self.num_heads = num_heads
        self.block = block
        self.different_layout_per_head = different_layout_per_head

    def make_layout(self, seq_len):
        n = seq_len // self.block
        return torch.ones((self.num_heads, n, n), dtype=torch.int64)


def install() -> None:
    """Install the stand-ins and put the reference on sys.path (idempotent)."""
    if not reference_available():
        raise RuntimeError(
            f"reference tree not found at {REFERENCE_ROOT}; the import oracle only runs in the build container"
        )
    if "multi_view_generation" in sys.modules:
        return

    pl = _mod(
        "pytorch_lightning",
        LightningModule=_LightningModule,
        LightningDataModule=object,
        Callback=object,
        Trainer=_Anything,
        seed_everything=lambda seed, **k: torch.manual_seed(seed),
    )
    _mod("pytorch_lightning.trainer", Trainer=_Anything)
    _mod("pytorch_lightning.callbacks", Callback=object, ModelCheckpoint=_Anything, LearningRateMonitor=_Anything)
    _mod("pytorch_lightning.loggers", WandbLogger=_Anything, Logger=object)
    _mod("pytorch_lightning.loggers.logger", Logger=object)
    _mod("pytorch_lightning.utilities", rank_zero_only=_identity_decorator)
    _mod("pytorch_lightning.utilities.rank_zero", rank_zero_only=_identity_decorator)
    assert pl is sys.modules["pytorch_lightning"]

    _mod("deepspeed")
    _mod("deepspeed.ops")
    _mod("deepspeed.ops.sparse_attention", SparsityConfig=_SparsityConfig)
    _mod("deepspeed.utils")
    _mod("deepspeed.utils.zero_to_fp32", get_fp32_state_dict_from_zero_checkpoint=_Anything())

    _mod("pyrootutils", setup_root=lambda *a, **k: None)
    _mod("image_utils", Im=_Anything, library_ops=_Anything())
    _mod("wandb", Image=_Anything, Histogram=_Anything)
    _mod("cv2", LINE_8=8, LINE_AA=16, FILLED=-1)
    tv = _mod("torchvision")
    _mod("torchvision.transforms", Compose=_Anything, Normalize=_Anything)
    _mod("torchvision.transforms.functional")
    _mod("torchvision.utils", make_grid=_Anything())
    tv.transforms = sys.modules["torchvision.transforms"]
    tv.utils = sys.modules["torchvision.utils"]

    _mod("hydra", main=lambda *a, **k: (lambda fn: fn))
    _mod("hydra.utils", instantiate=_Anything())
    _mod("hydra.core")
    _mod("hydra.core.hydra_config", HydraConfig=_Anything)
    _mod("omegaconf", DictConfig=dict, OmegaConf=_Anything, open_dict=_Anything)
    _mod("beartype", beartype=_identity_decorator)
    _mod("muse_maskgit_pytorch")
    _mod("muse_maskgit_pytorch.vqgan_vae", VQGanVAE=_Anything)
    _mod("muse_maskgit_pytorch.t5", t5_encode_text=_Anything(), get_encoded_dim=_Anything(), DEFAULT_T5_NAME="t5")
    _mod("nuscenes")
    _mod("nuscenes.map_expansion")
    _mod("nuscenes.map_expansion.map_api", NuScenesMap=_Anything)
    _mod("nuscenes.nuscenes", NuScenes=_Anything)
    _mod("nuscenes.utils")
    _mod("nuscenes.utils.data_classes", Box=_Anything, LidarPointCloud=_Anything)
    _mod("nuscenes.utils.geometry_utils", view_points=_Anything(), box_in_image=_Anything(), BoxVisibility=_Anything)
    _mod("nuscenes.utils.splits", create_splits_scenes=_Anything())
    _mod("pyquaternion", Quaternion=_Anything)
    _mod("shapely")
    _mod("shapely.geometry", MultiPolygon=_Anything, Polygon=_Anything, box=_Anything(), LineString=_Anything)
    _mod("shapely.ops")

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    # circular import vqgan <-> utils.callback: utils must come first
    import multi_view_generation.utils  # noqa: F401


def import_reference():
    """Return a namespace with the hot-path reference modules."""
    install()
    import importlib

    names = {
        "gpt": "multi_view_generation.modules.transformer.mingpt_sparse",
        "ssa": "multi_view_generation.modules.transformer.sparse_self_attention",
        "maskgen": "multi_view_generation.modules.transformer.mask_generator",
        "perm": "multi_view_generation.modules.transformer.permuter",
        "muse_net": "multi_view_generation.modules.stage2.muse_maskgit_pytorch",
        "muse_lm": "multi_view_generation.modules.stage2.cond_transformer_multi_view_muse",
        "ar_lm": "multi_view_generation.modules.stage2.cond_transformer_multi_view",
        "s1model": "multi_view_generation.modules.stage1.model",
        "vqgan": "multi_view_generation.modules.stage1.vqgan",
        "quant": "multi_view_generation.modules.stage1.quantize",
        "util": "multi_view_generation.bev_utils.util",
    }
    ns = types.SimpleNamespace()
    for short, full in names.items():
        setattr(ns, short, importlib.import_module(full))
    return ns

"""Host-side configuration mirror of the reference's ``GPTConfig``.

Reference: multi_view_generation/modules/transformer/mingpt_sparse.py:26-113 (``GPTConfig``),
multi_view_generation/bev_utils/util.py:20-39 (``Cameras`` / ``Dataset`` enums).

The constructor keyword names are identical to the reference's dataclass so that the
Hydra ``_target_`` in ``configs/model/stage_2.yaml`` / ``configs/experiment/
muse_stage_two_multi_view.yaml`` can point at this class unchanged.  All derived tables
(decode order, allowed mask, block layouts, camera-bias prior) are built on the host once
(setup time, exactly as in the reference) by :mod:`bevgen_amd.tables`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np
import torch


class Cameras(Enum):
    """Camera rigs (bev_utils/util.py:20-33).  Values are the dataset's channel names."""

    NUSCENES_FRONT = ("CAM_FRONT",)
    NUSCENES_CAMERAS = ("CAM_FRONT", "CAM_BACK", "CAM_FRONT_RIGHT", "CAM_FRONT_LEFT", "CAM_BACK_RIGHT", "CAM_BACK_LEFT")
    NUSCENES_ABLATION_CAMERAS = ("CAM_FRONT", "CAM_FRONT_RIGHT", "CAM_FRONT_LEFT")
    ARGOVERSE_CAMERAS = ("ring_side_left", "ring_front_left", "ring_front_right", "ring_side_right")
    ARGOVERSE_FRONT_CAMERAS = ("ring_front_left", "ring_front_center", "ring_front_right")
    ARGOVERSE_ALL_CAMERAS = ("ring_side_left", "ring_front_left", "ring_front_center", "ring_front_right", "ring_side_right")

    def __getitem__(self, index):
        return self._value_[index]

    def __len__(self):
        return len(self._value_)

    def __iter__(self):
        return iter(self._value_)

    def index(self, name):
        return self._value_.index(name)


class Dataset(Enum):
    NUSCENES = 0
    ARGOVERSE = 1


def _as_enum(enum_cls, v):
    if isinstance(v, enum_cls):
        return v
    if isinstance(v, int):
        return enum_cls(v)
    return enum_cls[str(v)]


@dataclass(eq=False)
class GPTConfig:
    """Sizes + static tables of the stage-2 transformer (gpt:26-113).

    Derived attributes (same names as the reference): ``num_cond_tokens`` K, ``num_cam_tokens`` T,
    ``num_img_tokens`` N, ``gpt_block_size`` L, ``num_pad_tokens``, ``forward_shuffle_idx``,
    ``backward_shuffle_idx``, ``attention_mask`` [L,L] f32 (0/1), ``layout`` [H,L/blk,L/blk] i64,
    ``prob_matrix`` [L,L] (camera-bias prior, only if ``camera_bias``).
    """

    embd_pdrop: float = 0.0
    resid_pdrop: float = 0.0
    attn_pdrop: float = 0.0
    num_layers: int = 24
    num_heads: int = 16
    num_embed: int = 1024
    hidden_size: int = 1024
    vocab_size: int = 1024
    cond_vocab_size: int = 1024
    num_cams: int = 6
    window_len: int = 32
    density: float = 1.0
    sparse_block_size: int = 16
    n_unmasked: int = 0
    backend: str = "hip"
    plot: bool = False
    cam_res: Tuple[int, int] = (256, 256)
    cam_latent_res: Tuple[int, int] = (16, 16)
    bev_latent_res: Tuple[int, int] = (16, 16)
    camera_bias: bool = False
    bev_embed: bool = False
    image_embed: bool = True
    cam_names: Any = "NUSCENES_CAMERAS"
    causal_order: bool = False
    output_dir: str = "output"
    legacy_prob_matrix: bool = True
    cam_intrinsics: Optional[Any] = None  # [Cm,3,3] for the non-legacy prior (else pretrained/cam_data_<ds>.pt)
    cam_extrinsics: Optional[Any] = None  # [Cm,4,4]
    dataset: Any = "NUSCENES"
    # explicit per-head block layouts (needed when density < 1: the reference draws them from
    # torch.multinomial under pl.seed_everything, perm:125-143); None -> built by tables.py
    layouts: Optional[Any] = None

    # derived ------------------------------------------------------------------------------
    cam_name_to_idx: Dict[str, int] = field(init=False, repr=False)
    num_cond_tokens: int = field(init=False)
    num_cam_tokens: int = field(init=False)
    num_img_tokens: int = field(init=False)
    num_pad_tokens: int = field(init=False)
    gpt_block_size: int = field(init=False)
    cam_latent_h: int = field(init=False)
    cam_latent_w: int = field(init=False)
    dataset_name: str = field(init=False)

    def __post_init__(self):
        from . import tables

        self.dataset = _as_enum(Dataset, self.dataset)
        self.dataset_name = self.dataset.name.lower()
        self.cam_names = _as_enum(Cameras, self.cam_names)
        if len(self.cam_names) != self.num_cams:
            raise ValueError(f"cam_names {self.cam_names.name} has {len(self.cam_names)} cameras, num_cams={self.num_cams}")
        self.cam_res = tuple(int(v) for v in self.cam_res)
        self.cam_latent_res = tuple(int(v) for v in self.cam_latent_res)
        self.bev_latent_res = tuple(int(v) for v in self.bev_latent_res)
        self.cam_name_to_idx = {k: v for v, k in enumerate(self.cam_names)}
        self.cam_latent_h, self.cam_latent_w = self.cam_latent_res
        self.num_cond_tokens = self.bev_latent_res[0] * self.bev_latent_res[1]
        self.num_cam_tokens = self.cam_latent_h * self.cam_latent_w
        self.num_img_tokens = self.num_cam_tokens * self.num_cams
        blk = self.sparse_block_size
        self.gpt_block_size = blk * int(np.ceil((self.num_img_tokens + self.num_cond_tokens) / blk))
        self.num_pad_tokens = self.gpt_block_size - (self.num_img_tokens + self.num_cond_tokens)

        fwd = tables.decode_order(self)
        self.forward_shuffle_idx = fwd
        self.backward_shuffle_idx = torch.argsort(fwd)

        pat = tables.attention_patterns(self)
        self.attention_mask = pat.allowed  # [L, L] f32 0/1 (== reference allowed_pattern[0])
        if self.layouts is not None:
            self.layout = torch.as_tensor(self.layouts).to(torch.int64)
        else:
            self.layout = tables.head_layouts(self, pat)
        self._patterns = pat
        self.prob_matrix = tables.camera_bias_prior(self, pat) if self.camera_bias else None

    # reference API (gpt:104-113)
    def get_mask(self):
        allowed = self.attention_mask[None].expand(self.num_heads, -1, -1).clone()
        return self.layout, allowed

    def forward_permuter(self, x):
        return x[:, self.forward_shuffle_idx]

    def backward_permuter(self, x):
        return x[:, self.backward_shuffle_idx]

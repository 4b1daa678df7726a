"""Static tables of the stage-2 transformer, built once on the host (setup time).

This is the host-side mirror of the reference's table builders:

* decode order            <- ``CustomPermuter.__init__``          permuter.py:33-88
* pixel grids             <- ``generate_grid`` / ``get_bev_grid``  mingpt_sparse.py:256-264, 116-141
* column yaw angles       <- ``get_col_angles``                    permuter.py:151-162,
                             ``compute_pixel_ray_directions``      bev_utils/nuscenes_helper.py:222-262
* allowed / window / prior <- ``outward_pattern``                  mask_generator.py:130-214
* per-head block layouts  <- ``multi_outward_pattern``             mask_generator.py:217-251
* camera-bias prior       <- ``outward_pattern(return_camera_bias_matrix=True)``  mask_generator.py:175-190,
                             ``get_bev_weights`` :73-86, ``get_image_direction_vectors`` :89-110, ``get_bev_sim`` :120-128

In the reference these run on the CPU too (they are not part of the GPU hot path); what the
GPU consumes is their *output*, uploaded through ``bevgen_set_tables`` (include/bevgen_hip.h).
The formulas are written directly (closed form in sequence coordinates) rather than through the
reference's scatter/gather construction; equality with the imported reference is pinned by
tests/test_tables.py against tests/golden/tables_*.npz.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F
from scipy.spatial import distance


# nuScenes per-camera (fx, fy, yaw) used by the legacy prior (permuter.py:151); data, in NUSCENES_CAMERAS order.
_NUSC_CAM_FX_YAW = (
    (1266.417203046554, 0.005684811144346602),  # CAM_FRONT
    (809.2209905677063, 3.1391709219861887),  # CAM_BACK
    (1260.8474446004698, 5.298742851167251),  # CAM_FRONT_RIGHT
    (1272.5979470598488, 0.9627404474321728),  # CAM_FRONT_LEFT
    (1259.5137405846733, 4.349372983905386),  # CAM_BACK_RIGHT
    (1256.7414812095406, 1.895431863668132),  # CAM_BACK_LEFT
)


# --------------------------------------------------------------------------------------------
# S2: decode order
# --------------------------------------------------------------------------------------------
def decode_order(cfg) -> torch.Tensor:
    """forward_shuffle_idx[s] = camera-major token index decoded at step s (perm:33-82).

    * ``causal_order=False`` -> identity (perm:79-80).
    * nuScenes rigs: per latent row, each camera triple (left, centre, right) is walked from the
      centre column outwards, alternating left/right; the front and back triples are interleaved.
    * any other rig: row-major across cameras (all cameras' row 0, then row 1, ...).
    """
    from .config import Cameras, Dataset

    C, h, w = cfg.num_cams, cfg.cam_latent_h, cfg.cam_latent_w
    N = C * h * w
    if not cfg.causal_order:
        return torch.arange(N)

    def tok(cam, row, col):
        return (cam * h + row) * w + col

    order = []
    if cfg.dataset == Dataset.NUSCENES:
        if C == 3:
            rig = Cameras.NUSCENES_ABLATION_CAMERAS
            triples = [("CAM_FRONT_LEFT", "CAM_FRONT", "CAM_FRONT_RIGHT")]
        else:
            rig = Cameras.NUSCENES_CAMERAS
            triples = [("CAM_FRONT_LEFT", "CAM_FRONT", "CAM_FRONT_RIGHT"), ("CAM_BACK_RIGHT", "CAM_BACK", "CAM_BACK_LEFT")]
        mid = w // 2
        for row in range(h):
            walks = []
            for left, centre, right in triples:
                li, ci, ri = rig.index(left), rig.index(centre), rig.index(right)
                head = []
                if w % 2 == 1:  # odd width: the centre column goes first, alone
                    head.append(tok(ci, row, mid))
                    right_centre = [tok(ci, row, c) for c in range(mid + 1, w)]
                else:
                    right_centre = [tok(ci, row, c) for c in range(mid, w)]
                leftwards = [tok(ci, row, c) for c in range(mid - 1, -1, -1)] + [tok(li, row, c) for c in range(w - 1, -1, -1)]
                rightwards = right_centre + [tok(ri, row, c) for c in range(w)]
                walk = list(head)
                for a, b in zip(leftwards, rightwards):  # zip truncates like the reference
                    walk += [a, b]
                walks.append(walk)
            for group in zip(*walks):
                order.extend(group)
    else:
        for row in range(h):
            for cam in range(C):
                order.extend(tok(cam, row, c) for c in range(w))
    return torch.tensor(order, dtype=torch.int64)


def seq_to_pixel(cfg) -> torch.Tensor:
    """[N,3] (cam,row,col) of each camera-major token (perm:26-30)."""
    C, h, w = cfg.num_cams, cfg.cam_latent_h, cfg.cam_latent_w
    cam, row, col = torch.meshgrid(torch.arange(C), torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([cam, row, col], -1).reshape(-1, 3)


# --------------------------------------------------------------------------------------------
# S6: grids
# --------------------------------------------------------------------------------------------
def generate_grid(height: int, width: int) -> torch.Tensor:
    """[1,3,h,w]: (x in [0,1] along width, y in [0,1] along height, 1)  (gpt:256-264)."""
    xs = torch.linspace(0, 1, width)
    ys = torch.linspace(0, 1, height)
    gx = xs[None, :].expand(height, width)
    gy = ys[:, None].expand(height, width)
    return torch.stack([gx, gy, torch.ones(height, width)], 0)[None].contiguous()


def image_plane(cfg) -> torch.Tensor:
    """[1,1,3,h,w] pixel plane; x scaled by cam_res[0] (=height, reference quirk gpt:291), y by cam_res[1]."""
    plane = generate_grid(cfg.cam_latent_h, cfg.cam_latent_w)[None].clone()
    plane[:, :, 0] *= cfg.cam_res[0]
    plane[:, :, 1] *= cfg.cam_res[1]
    return plane


def get_bev_grid(cfg, offset: int = 0) -> torch.Tensor:
    """[3,h,w] ego-frame metres of each BEV latent cell, 80 m extent (gpt:116-141)."""
    h, w = cfg.bev_latent_res
    grid = generate_grid(h, w)[0].clone()
    grid[0] = w * grid[0]
    grid[1] = h * grid[1]
    sh, sw = h / 80, w / 80
    view = torch.tensor([[0.0, -sw, w / 2.0], [-sh, 0.0, h * offset + h / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    out = view.inverse() @ grid.reshape(3, h * w)
    return out.reshape(3, h, w)


# --------------------------------------------------------------------------------------------
# S5: per-column yaw (legacy prior)
# --------------------------------------------------------------------------------------------
def col_angles(cfg) -> np.ndarray:
    """[6,w] float32 yaw of every latent column of the six nuScenes cameras (perm:153-162).

    The reference calls ``compute_pixel_ray_directions(uv, fx, fy, img_h, img_w)`` with height and
    width swapped relative to the callee's signature, so the principal point used is
    (img_h/2, img_w/2) = (450, 800); reproduced here because the prior depends on it.
    """
    img_w, img_h = 1600.0, 900.0
    w = cfg.cam_latent_w
    out = np.zeros((len(_NUSC_CAM_FX_YAW), w), dtype=np.float32)
    for ci, (fx, yaw) in enumerate(_NUSC_CAM_FX_YAW):
        for i in range(w):
            u, v = img_w * ((i + 0.5) / w), img_h / 2
            ray = np.array([u - img_h / 2, v - img_w / 2, fx], dtype=np.float64)
            x_dir = ray[0] / np.linalg.norm(ray)
            out[ci, i] = np.mod(yaw + (-x_dir), 2 * np.pi).astype(np.float32)
    return out


# --------------------------------------------------------------------------------------------
# S3: allowed / window / prior
# --------------------------------------------------------------------------------------------
@dataclass
class Patterns:
    allowed: torch.Tensor  # [L,L] f32 0/1
    static_layout: torch.Tensor  # [L/blk, L/blk] i64
    prob_layout: torch.Tensor  # [L/blk, L/blk] f32
    prob_img: torch.Tensor  # [N,N] image<->image prior, zeroed where not allowed (sequence order)
    angles_seq: Optional[np.ndarray]  # per camera-major token yaw (legacy only)


def _load_cam_data(cfg):
    if cfg.cam_intrinsics is not None and cfg.cam_extrinsics is not None:
        return torch.as_tensor(cfg.cam_intrinsics, dtype=torch.float32), torch.as_tensor(cfg.cam_extrinsics, dtype=torch.float32)
    path = os.path.join("pretrained", f"cam_data_{cfg.dataset_name}.pt")
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"legacy_prob_matrix=False needs camera calibration: pass cam_intrinsics/cam_extrinsics or provide {path} "
            "({'intrinsics': [1,C,3,3], 'extrinsics': [1,C,4,4]}, as written by the reference's Argoverse.save_cam_data)"
        )
    data = torch.load(path)
    return data["intrinsics"][0].float(), data["extrinsics"][0].float()


def image_direction_vectors(cfg) -> torch.Tensor:
    """[N,3] unit ray direction (ego frame) of every camera-major token (maskgen:89-110)."""
    from .config import Cameras, Dataset

    intr, extr = _load_cam_data(cfg)
    rig = Cameras.NUSCENES_CAMERAS if cfg.dataset == Dataset.NUSCENES else cfg.cam_names
    h, w = cfg.cam_latent_h, cfg.cam_latent_w
    plane = generate_grid(h, w).clone()  # [1,3,h,w]
    plane[:, 0] *= 1600
    plane[:, 1] *= 900
    E_inv = extr.inverse()  # [Cm,4,4]
    I_inv = intr.inverse()  # [Cm,3,3]
    pix = plane.reshape(1, 3, h * w)
    cam = I_inv @ pix  # [Cm,3,hw]
    cam = F.pad(cam, (0, 0, 0, 1), value=1)  # [Cm,4,hw]
    d = E_inv @ cam  # [Cm,4,hw]
    origin = E_inv[:, :, 3:]  # [Cm,4,1]
    rays = (d - origin).permute(0, 2, 1)[..., :3]  # [Cm,hw,3]
    rays = torch.stack([rays[rig.index(name)] for name in cfg.cam_names], 0).reshape(-1, 3)
    return F.normalize(rays, dim=1)


def _cos_sim01(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return (F.normalize(a) @ F.normalize(b).t() + 1) / 2


def attention_patterns(cfg) -> Patterns:
    N, K, P, L = cfg.num_img_tokens, cfg.num_cond_tokens, cfg.num_pad_tokens, cfg.gpt_block_size
    blk = cfg.sparse_block_size
    fwd = cfg.forward_shuffle_idx
    r = torch.arange(N)[:, None]
    c = torch.arange(N)[None, :]
    # both are expressed in *sequence* (decode-order) coordinates (maskgen:132-148)
    causal = c <= r
    window = causal & (c >= r - cfg.window_len)

    angles = None
    if cfg.legacy_prob_matrix:
        s2p = seq_to_pixel(cfg)
        rows = s2p[:, 1]
        angles = col_angles(cfg)[s2p[:, 0].numpy(), s2p[:, 2].numpy()]
        unit = np.stack([np.cos(angles), np.sin(angles)], 1)
        ang_d = torch.from_numpy(np.rad2deg(distance.cdist(unit, unit, metric="cosine")))  # f64, degrees of a cosine *distance*
        pts = torch.stack([rows, torch.zeros_like(rows)], 1).float()
        row_d = torch.cdist(pts, pts, p=2.0)
        sigma = 4.0
        prob = torch.exp(-0.5 * sigma ** (-2.0) * (ang_d + row_d))
    else:
        rays = image_direction_vectors(cfg)
        prob = _cos_sim01(rays, rays)
    if cfg.causal_order:
        prob = prob[:, fwd][fwd, :]
    prob = prob.clone()
    prob[~causal] = 0

    # prior used for the *layout* sampling: 0.5 on every cond column (maskgen:192-193)
    pm = F.pad(prob, (0, P, 0, P), value=0).clamp(0, 1)
    pm_layout = F.pad(F.pad(pm, (0, 0, K, 0), value=0), (K, 0, 0, 0), value=0.5)
    prob_layout = F.avg_pool2d(pm_layout[None].float(), kernel_size=blk, stride=blk)[0]
    if prob_layout.ndim == 0:
        prob_layout = prob_layout.reshape(1, 1)

    static = torch.zeros((L, L), dtype=torch.bool)
    static[K : K + N, K : K + N] = window
    if P:
        static[L - P :, 0] = True
        static[L - P :, 1:] = False
    static_layout = F.max_pool2d(static[None].float(), kernel_size=blk, stride=blk)[0].to(torch.int64)

    allowed = torch.zeros((L, L), dtype=torch.bool)
    allowed[:, :K] = True
    allowed[K : K + N, K : K + N] = causal
    if P:
        allowed[L - P :, 1:] = False
    return Patterns(allowed.float(), static_layout, prob_layout, prob, angles)


# --------------------------------------------------------------------------------------------
# S4: per-head block layouts
# --------------------------------------------------------------------------------------------
def head_layouts(cfg, pat: Patterns) -> torch.Tensor:
    """[H, L/blk, L/blk] i64 (maskgen:217-228).

    ``static ∪ multinomial(prob_layout, n)`` with ``n = nblocks·density − static.sum()`` drawn
    without replacement, zero-probability blocks dropped.  When n covers every positive block
    (density = 1.0, the shipped configs) the result is the deterministic ``static ∪ (prob>0)``;
    otherwise the draw uses ``torch.multinomial`` on the caller's CPU RNG state, the same primitive
    and call order as the reference, so an identically seeded process reproduces its layouts.
    """
    static = pat.static_layout.bool()
    prob = pat.prob_layout
    n = int((prob.shape[0] * prob.shape[1]) * cfg.density - static.sum())
    positive = prob > 0
    layouts = []
    for _ in range(cfg.num_heads):
        if n >= int(positive.sum()):
            sampled = positive.clone()
        else:
            idx = torch.multinomial(prob.flatten(), n, replacement=False)
            sampled = torch.zeros_like(positive).flatten()
            sampled[idx] = True
            sampled = sampled.reshape(positive.shape) & positive
        layouts.append((static | sampled).to(torch.int64))
    return torch.stack(layouts)


# --------------------------------------------------------------------------------------------
# camera-bias prior
# --------------------------------------------------------------------------------------------
def bev_weights(cfg, angles: np.ndarray) -> torch.Tensor:
    """[N,K] f64 (cos-sim+1)/2 between token yaw and BEV-cell bearing (maskgen:73-86)."""
    bh, bw = cfg.bev_latent_res
    rr, cc = torch.meshgrid(torch.arange(bh), torch.arange(bw), indexing="ij")
    # kept as strided column views of one [K,2] tensor: torch's CPU atan2 takes its scalar (non-SLEEF) path
    # for strided operands, which is 1 ulp away from the vectorised path on some cells; the reference
    # evaluates it that way, so the prior is only bit-identical if we do too.
    yx = torch.stack([rr.reshape(-1), cc.reshape(-1)], -1).float()
    yx[:, 0] = -yx[:, 0] + ((bh // 2) - 0.5)
    yx[:, 1] = yx[:, 1] - ((bw // 2) - 0.5)
    bearing = torch.remainder(torch.atan2(yx[:, 0], yx[:, 1]) - torch.pi / 2, 2 * torch.pi)
    a = np.stack([np.cos(angles), np.sin(angles)], 1)
    b = np.stack([np.cos(bearing), np.sin(bearing)], 1)
    sim = torch.from_numpy(1 - distance.cdist(a, b, metric="cosine"))
    return (sim + 1) / 2


def bev_sim(cfg) -> torch.Tensor:
    """[N,K] (cos-sim+1)/2 between token rays and flattened BEV cell positions (maskgen:120-128)."""
    grid = get_bev_grid(cfg).reshape(3, -1).t().clone()
    grid[:, 2] = 0
    grid = F.normalize(grid, dim=1)
    return _cos_sim01(image_direction_vectors(cfg), grid)


def camera_bias_prior(cfg, pat: Patterns) -> torch.Tensor:
    """[L,L] prior added (with the learned tril ``camera_bias_emb``) to the attention scores (maskgen:172-190)."""
    N, K, P = cfg.num_img_tokens, cfg.num_cond_tokens, cfg.num_pad_tokens
    fwd = cfg.forward_shuffle_idx
    pm = F.pad(pat.prob_img, (0, P, 0, P), value=0).clamp(0, 1)
    pm = F.pad(F.pad(pm, (0, 0, K, 0), value=0), (K, 0, 0, 0), value=1.0)
    if cfg.legacy_prob_matrix:
        sim = bev_weights(cfg, pat.angles_seq[fwd.numpy()])
    else:
        sim = bev_sim(cfg)[fwd, :]
    pm[K : K + N, :K] = sim.to(pm.dtype)
    return pm

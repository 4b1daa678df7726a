"""Output writer of the sampling path: the on-disk layout of the reference's ``GenerateImages.save_raw_data`` (utils/callback.py:72-132),
which is the wire format its evaluation tooling reads (one directory per sample token, one JPEG per camera, the BEV condition as ``bev.npz``).

    <save_dir>/sample/<token>[_XXXXX]/<cam_name>.jpg        generated views            (callback.py:95-96, 112-113)
    <save_dir>/sample_gt/<token>[_XXXXX]/<cam_name>.jpg     ground-truth views         (callback.py:98-99, 115-116)
    <save_dir>/sample/<token>[_XXXXX]/bev.npz               np.savez_compressed(seg)   (callback.py:101-105, 118-120): key 'arr_0', float32
    <save_dir>/sample_gt/<token>[_XXXXX]/bev.npz            same array                 (callback.py:107-109, 124-125)
    <save_dir>/{gt,rec,gen}/<image_paths[cam][b]>           nuScenes-format copies + <path>.npz with the intrinsics (callback.py:127-137), only when the
                                                            batch carries 'image_paths' and the dataset is nuScenes

``_XXXXX`` is the 5-character [A-Z0-9] suffix of ``rand_str=True`` (callback.py:93: several samples of one layout do not overwrite each other).
    <save_dir>/viz/<token>.png                              contact sheet (six-view scenes)      (callback.py:76-86)
    <save_dir>/sample/<token>/bev.png                       class-coloured BEV (rand_str runs: callback.py:105; nuScenes otherwise: callback.py:117-118)
The two PNGs are visualisations: same file names and content (generated row over ground-truth row, the BEV rendering beside them; class colours of
the nuScenes devkit, highest class wins, blended with light grey by confidence - bev_utils/visualize.py:67-107), own layout (the reference composes
its sheet with the third-party ``image_utils`` package).

MI355X side: the float pixels are converted to uint8 ON THE GPU (``parallel.to_uint8``: round(x*255), clamp) and leave the device in ONE copy per
batch (1.18 MB per six-view 256x256 scene instead of 4.7 MB of fp32); JPEG encoding runs on a host thread pool off the sampling path - ``flush()``
joins it.  The reference converts through the third-party ``image_utils.Im`` (not in the tree): float->uint8 rounding and JPEG quality are therefore
"parity unpinned"; this writer uses round-to-nearest and PIL's default quality (75), files decode to the same size and channel order.
"""
from __future__ import annotations

import os
import random
import string
from concurrent.futures import Future, ThreadPoolExecutor
from datetime import datetime
from pathlib import Path
from typing import Dict, List, Mapping, Optional, Sequence

import numpy as np
import torch

from .config import Dataset
from .parallel import to_uint8


def _save_jpeg(chw_u8: np.ndarray, path: Path) -> None:
    from PIL import Image  # deferred: the sampling path itself never needs PIL

    os.makedirs(path.parent, exist_ok=True)
    Image.fromarray(np.ascontiguousarray(chw_u8.transpose(1, 2, 0))).save(path)


# class colours (nuScenes devkit colour map; Argoverse ordering of the 7-channel segmentation after the reference's channel shuffle, visualize.py:86-88)
_BEV_COLORS = np.array([(110, 110, 110), (130, 130, 130), (255, 200, 0), (255, 127, 80), (0, 0, 230), (255, 158, 0), (255, 99, 71)], dtype=np.float32)
_BEV_EMPTY = np.array((200, 200, 200), dtype=np.float32)
_BEV_CHANNEL_ORDER = [4, 5, 6, 3, 1, 0, 2]


def render_bev(seg: np.ndarray, channel_axis: Optional[int] = None) -> np.ndarray:
    """Class probabilities in [0, 1] (or uint8 0..255) -> [h, w, 3] uint8: per cell the highest class label wins (ties go to the higher index), its
    colour blended with light grey by its confidence (visualize.py:67-107).  Other channel counts get the palette cyclically.
    ``channel_axis``: 0 for [c, h, w], -1 for [h, w, c] (the batch layout of the reference: get_input moves the LAST axis, muse_lm:166-179);
    None = the smallest axis (class counts are far below the grid size)."""
    a = np.asarray(seg, dtype=np.float32)
    if a.ndim != 3:
        raise ValueError(f"render_bev: expected a 3-D segmentation, got shape {a.shape}")
    if a.max() > 1:
        a = a / 255.0
    if channel_axis is None:
        channel_axis = 0 if a.shape[0] < a.shape[2] else -1
    if channel_axis % 3 != 2:
        a = np.moveaxis(a, channel_axis, -1)
    c = a.shape[-1]
    if c == len(_BEV_CHANNEL_ORDER):
        a = a[..., _BEV_CHANNEL_ORDER]
    colors = _BEV_COLORS[np.arange(c) % len(_BEV_COLORS)]
    idx = (a + 1e-5 * np.arange(c)[None, None]).argmax(-1)
    val = np.take_along_axis(a, idx[..., None], -1).clip(0, 1)
    return np.uint8(val * colors[idx] + (1 - val) * _BEV_EMPTY[None, None])


def _resize_nearest(img: np.ndarray, h: int, w: int) -> np.ndarray:
    ys = (np.arange(h) * img.shape[0] // h).clip(0, img.shape[0] - 1)
    xs = (np.arange(w) * img.shape[1] // w).clip(0, img.shape[1] - 1)
    return img[ys][:, xs]


def contact_sheet(gen: np.ndarray, gt: Optional[np.ndarray], bev_rgb: np.ndarray) -> np.ndarray:
    """gen / gt [C, 3, H, W] uint8 -> [rows*H, C*W + H, 3] uint8: the generated views in a row, the ground truth below, the BEV rendering at the right."""
    rows = [gen] + ([gt] if gt is not None else [])
    H = gen.shape[-2]
    bev = _resize_nearest(bev_rgb, H, H)
    strips = [np.concatenate([np.concatenate([v.transpose(1, 2, 0) for v in r], 1), bev], 1) for r in rows]
    return np.concatenate(strips, 0)


def _save_png(hwc_u8: np.ndarray, path: Path) -> None:
    from PIL import Image

    os.makedirs(path.parent, exist_ok=True)
    Image.fromarray(np.ascontiguousarray(hwc_u8)).save(path)


def _cam_name(batch: Mapping, cam: int, b: int) -> str:
    """batch['cam_name'] is the default-collated list [cam][batch] of strings (callback.py:95)."""
    names = batch["cam_name"]
    entry = names[cam]
    return str(entry[b] if isinstance(entry, (list, tuple)) else entry)


class SceneWriter:
    """Writes batches of generated scenes; one instance per process (per GPU rank - every rank writes its own files, as in the reference)."""

    def __init__(self, save_dir: Optional[str], rand_str: bool = False, workers: int = 8, seed: Optional[int] = None):
        self.save_dir = None if save_dir is None else Path(save_dir)
        self.rand_str = rand_str
        self._pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="bevgen-jpeg")
        self._pending: List[Future] = []
        self._rng = random.Random(seed)

    # ------------------------------------------------------------------------------------------------------------------ public
    def write(self, outputs: Mapping[str, torch.Tensor], batch: Mapping, dataset: Dataset = Dataset.NUSCENES, save_nuscenes_fmt: bool = True) -> List[str]:
        """outputs: {'gen','gt'[,'rec']} float [B,C,3,H,W] in [0,1] (device or host); batch: 'sample_token' [B], 'cam_name' [C][B], 'segmentation' [B,...]
        and optionally 'image_paths' [C][B] + 'intrinsics' [B,C,3,3].  Returns the token directory names used.  Encoding is asynchronous: call flush()."""
        if self.save_dir is None:
            raise ValueError("SceneWriter needs a save_dir")
        host: Dict[str, np.ndarray] = {}
        for k in ("gen", "gt", "rec"):
            if k in outputs and outputs[k] is not None:
                t = outputs[k].detach()
                # uint8 is the storage / wire format (vq_decode(uint8=True), parallel.gather_scenes): passed through; floats are converted on the device
                host[k] = (t if t.dtype == torch.uint8 else to_uint8(t)).cpu().numpy()   # one D2H copy per tensor
        gen = host["gen"]
        B, C = gen.shape[:2]
        seg = batch["segmentation"]
        tokens: List[str] = []
        for b in range(B):
            tok = str(batch["sample_token"][b])
            if self.rand_str:
                tok = tok + "_" + "".join(self._rng.choices(string.ascii_uppercase + string.digits, k=5))
            tokens.append(tok)
            seg_b = seg[b].detach().to(dtype=torch.float).cpu().numpy() if isinstance(seg, torch.Tensor) else np.asarray(seg[b], dtype=np.float32)
            for split in ("sample", "sample_gt"):
                d = self.save_dir / split / tok
                os.makedirs(d, exist_ok=True)
                np.savez_compressed(d / "bev.npz", seg_b)
            if seg_b.ndim == 3:
                bev_rgb = render_bev(seg_b)
                if self.rand_str or dataset == Dataset.NUSCENES:                         # callback.py:105 (rand_str runs), :117-118 (nuScenes otherwise)
                    self._pending.append(self._pool.submit(_save_png, bev_rgb, self.save_dir / "sample" / tok / "bev.png"))
                if C == 6:                                                               # callback.py:76-86 (six-view scenes only)
                    sheet = contact_sheet(gen[b], host.get("gt", None)[b] if "gt" in host else None, bev_rgb)
                    self._pending.append(self._pool.submit(_save_png, sheet, self.save_dir / "viz" / f"{batch['sample_token'][b]}.png"))
            for cam in range(C):
                name = _cam_name(batch, cam, b)
                self._submit(gen[b, cam], self.save_dir / "sample" / tok / f"{name}.jpg")
                if "gt" in host:
                    self._submit(host["gt"][b, cam], self.save_dir / "sample_gt" / tok / f"{name}.jpg")
                if save_nuscenes_fmt and dataset == Dataset.NUSCENES and "image_paths" in batch:
                    rel = Path(str(batch["image_paths"][cam][b]))
                    for split in ("gt", "rec", "gen"):
                        if split in host:
                            self._submit(host[split][b, cam], self.save_dir / split / rel)
                    if "intrinsics" in batch:
                        p = self.save_dir / "gen" / rel
                        os.makedirs(p.parent, exist_ok=True)
                        np.savez(p.with_suffix(".npz"), batch["intrinsics"][b, cam].detach().to(dtype=torch.float).cpu().numpy())
        return tokens

    def flush(self) -> None:
        """Wait for the JPEG encoders; re-raises the first failure."""
        pending, self._pending = self._pending, []
        for f in pending:
            f.result()

    def close(self) -> None:
        self.flush()
        self._pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------------------------------------------------------ private
    def _submit(self, chw_u8: np.ndarray, path: Path) -> None:
        self._pending.append(self._pool.submit(_save_jpeg, chw_u8, path))


class GenerateImages:
    """Drop-in for the reference callback of the same name (utils/callback.py:33-37, 139-160; Hydra target `…utils.callback.GenerateImages`): same
    constructor arguments, `save_raw_data(trainer, pl_module, outputs, batch)` and the Lightning `on_test_batch_end` hook that generate.py relies on.
    Derives from pytorch_lightning.Callback when Lightning is installed."""

    def __init__(self, save_dir=None, figure_format=False, rand_str=False, **kwargs):
        self.save_dir = save_dir
        self.figure_format = figure_format
        self.rand_str = rand_str
        self._writer: Optional[SceneWriter] = None

    def _get_writer(self, trainer) -> SceneWriter:
        if self._writer is None:
            save_dir = self.save_dir
            if save_dir is None:   # callback.py:74: <log_dir>/results/<timestamp>
                save_dir = os.path.join(getattr(trainer, "log_dir", None) or ".", "results", datetime.now().strftime("%Y_%m_%d-%H_%M"))
            self._writer = SceneWriter(save_dir, rand_str=self.rand_str)
        return self._writer

    def save_raw_data(self, trainer, pl_module, outputs, batch, save_nuscenes_fmt: bool = True):
        w = self._get_writer(trainer)
        return w.write(outputs, batch, dataset=pl_module.cfg.dataset, save_nuscenes_fmt=save_nuscenes_fmt)

    def on_test_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0):
        self.save_raw_data(trainer, pl_module, outputs, batch)

    def on_test_end(self, trainer=None, pl_module=None):
        if self._writer is not None:
            self._writer.flush()

    def flush(self):
        if self._writer is not None:
            self._writer.flush()


try:  # make it a real Lightning callback when Lightning exists (it does not in the build image)
    import pytorch_lightning as _pl

    GenerateImages = type("GenerateImages", (GenerateImages, _pl.Callback), {"__doc__": GenerateImages.__doc__})
except Exception:  # pragma: no cover
    pass

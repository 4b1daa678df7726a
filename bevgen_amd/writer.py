"""Output writer of the sampling path: the on-disk layout of the reference's ``GenerateImages.save_raw_data`` (utils/callback.py:72-132),
which is the wire format its evaluation tooling reads (one directory per sample token, one JPEG per camera, the BEV condition as ``bev.npz``).

    <save_dir>/sample/<token>[_XXXXX]/<cam_name>.jpg        generated views            (callback.py:95-96, 112-113)
    <save_dir>/sample_gt/<token>[_XXXXX]/<cam_name>.jpg     ground-truth views         (callback.py:98-99, 115-116)
    <save_dir>/sample/<token>[_XXXXX]/bev.npz               np.savez_compressed(seg)   (callback.py:101-105, 118-120): key 'arr_0', float32
    <save_dir>/sample_gt/<token>[_XXXXX]/bev.npz            same array                 (callback.py:107-109, 124-125)
    <save_dir>/{gt,rec,gen}/<image_paths[cam][b]>           nuScenes-format copies + <path>.npz with the intrinsics (callback.py:127-137), only when the
                                                            batch carries 'image_paths' and the dataset is nuScenes

``_XXXXX`` is the 5-character [A-Z0-9] suffix of ``rand_str=True`` (callback.py:93: several samples of one layout do not overwrite each other).
Not written: the ``viz/<token>.png`` contact sheet and ``bev.png`` (visualisation helpers of the reference, out of scope - DESIGN.md section 8).

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
                host[k] = to_uint8(outputs[k].detach()).cpu().numpy()   # uint8 on the device, one D2H copy per tensor
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

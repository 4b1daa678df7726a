"""Output writer: on-disk layout of GenerateImages.save_raw_data (utils/callback.py:72-132)."""
import os

import numpy as np
import torch

from bevgen_amd.config import Dataset
from bevgen_amd.writer import GenerateImages, SceneWriter


def _batch(B, cams, with_paths=False):
    g = torch.Generator().manual_seed(0)
    batch = {
        "sample_token": [f"tok{b:03d}" for b in range(B)],
        "cam_name": [[c] * B for c in cams],                       # default-collated [cam][batch]
        "segmentation": torch.randint(0, 2, (B, 7, 32, 32), generator=g).to(torch.uint8),
        "intrinsics": torch.rand(B, len(cams), 3, 3, generator=g),
    }
    if with_paths:
        batch["image_paths"] = [[f"samples/{c}/img_{b}.jpg" for b in range(B)] for c in cams]
    return batch


def _outputs(B, C, H=24, W=40):
    g = torch.Generator().manual_seed(1)
    def smooth():  # low-frequency pictures: JPEG reproduces them closely (noise would not survive chroma subsampling)
        coarse = torch.rand(B * C, 3, 3, 5, generator=g)
        return torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=True).reshape(B, C, 3, H, W)
    return {k: smooth() for k in ("gen", "gt", "rec")}


def test_layout_and_contents(tmp_path):
    from PIL import Image
    cams = ["CAM_FRONT_LEFT", "CAM_FRONT", "CAM_FRONT_RIGHT"]
    B = 2
    batch, out = _batch(B, cams, with_paths=True), _outputs(B, len(cams))
    with SceneWriter(str(tmp_path)) as w:
        toks = w.write(out, batch, dataset=Dataset.NUSCENES)
    assert toks == batch["sample_token"]
    for b, tok in enumerate(toks):
        for split in ("sample", "sample_gt"):
            d = tmp_path / split / tok
            assert sorted(os.listdir(d)) == sorted([f"{c}.jpg" for c in cams] + ["bev.npz"])
            seg = np.load(d / "bev.npz")["arr_0"]                  # np.savez_compressed(path, array) -> key arr_0
            assert seg.dtype == np.float32 and np.array_equal(seg, batch["segmentation"][b].float().numpy())
        img = np.asarray(Image.open(tmp_path / "sample" / tok / "CAM_FRONT.jpg"))
        assert img.shape == (24, 40, 3) and img.dtype == np.uint8
        ref = (out["gen"][b, 1] * 255).round().permute(1, 2, 0).numpy()
        assert np.abs(img.astype(np.float32) - ref).mean() < 4.0   # lossy, but the same picture (a channel swap or transposition would be far off)
    # nuScenes-format copies + intrinsics
    for split in ("gt", "rec", "gen"):
        assert (tmp_path / split / "samples" / "CAM_FRONT" / "img_1.jpg").exists()
    k = np.load(tmp_path / "gen" / "samples" / "CAM_FRONT" / "img_1.npz")["arr_0"]
    assert np.allclose(k, batch["intrinsics"][1, 1].numpy())


def test_rand_str_and_non_nuscenes(tmp_path):
    cams = ["ring_front_center"]
    batch, out = _batch(3, cams, with_paths=True), _outputs(3, 1)
    w = SceneWriter(str(tmp_path), rand_str=True, seed=7)
    toks = w.write(out, batch, dataset=Dataset.ARGOVERSE)
    w.close()
    assert all(t.startswith(s + "_") and len(t) == len(s) + 6 and t[-5:].isalnum() and t[-5:].upper() == t[-5:] for t, s in zip(toks, batch["sample_token"]))
    assert len(set(toks)) == 3
    assert not (tmp_path / "gen").exists()                          # the nuScenes-format copies are nuScenes only (callback.py:127)
    assert sorted(os.listdir(tmp_path / "sample")) == sorted(toks)


def test_callback_dropin(tmp_path):
    class M:  # stand-in for the LightningModule: the callback only reads .cfg.dataset
        class cfg:
            dataset = Dataset.NUSCENES
    cb = GenerateImages(save_dir=str(tmp_path), rand_str=False)
    batch, out = _batch(1, ["CAM_BACK"]), _outputs(1, 1)
    cb.on_test_batch_end(None, M(), out, batch, 0)
    cb.on_test_end()
    assert (tmp_path / "sample" / "tok000" / "CAM_BACK.jpg").exists() and (tmp_path / "sample_gt" / "tok000" / "bev.npz").exists()

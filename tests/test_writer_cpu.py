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
            # nuScenes, rand_str off: the class-coloured bev.png goes beside the generated views only (callback.py:117-118)
            assert sorted(os.listdir(d)) == sorted([f"{c}.jpg" for c in cams] + ["bev.npz"] + (["bev.png"] if split == "sample" else []))
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
    # Argoverse without rand_str: no bev.png (callback.py:117 is nuScenes only)
    w = SceneWriter(str(tmp_path / "plain"))
    w.write(out, batch, dataset=Dataset.ARGOVERSE)
    w.close()
    assert not (tmp_path / "plain" / "sample" / "tok000" / "bev.png").exists() and (tmp_path / "plain" / "sample" / "tok000" / "bev.npz").exists()


def test_uint8_outputs_are_the_wire_format_and_pass_through(tmp_path):
    """vq_decode(uint8=True) / parallel.gather_scenes hand over uint8 pixels: the writer must not rescale them (x255 would saturate every nonzero pixel)."""
    from PIL import Image
    cams = ["CAM_FRONT"]
    batch, out = _batch(1, cams), _outputs(1, 1)
    u8 = {k: (v * 255).round().clamp(0, 255).to(torch.uint8) for k, v in out.items()}
    with SceneWriter(str(tmp_path / "u8")) as w:
        w.write(u8, batch, dataset=Dataset.ARGOVERSE)
    with SceneWriter(str(tmp_path / "f32")) as w:
        w.write(out, batch, dataset=Dataset.ARGOVERSE)
    a = np.asarray(Image.open(tmp_path / "u8" / "sample" / "tok000" / "CAM_FRONT.jpg"))
    b = np.asarray(Image.open(tmp_path / "f32" / "sample" / "tok000" / "CAM_FRONT.jpg"))
    assert np.array_equal(a, b) and a.max() < 255


def test_callback_dropin(tmp_path):
    class M:  # stand-in for the LightningModule: the callback only reads .cfg.dataset
        class cfg:
            dataset = Dataset.NUSCENES
    cb = GenerateImages(save_dir=str(tmp_path), rand_str=False)
    batch, out = _batch(1, ["CAM_BACK"]), _outputs(1, 1)
    cb.on_test_batch_end(None, M(), out, batch, 0)
    cb.on_test_end()
    assert (tmp_path / "sample" / "tok000" / "CAM_BACK.jpg").exists() and (tmp_path / "sample_gt" / "tok000" / "bev.npz").exists()


def test_bev_rendering_and_contact_sheet(tmp_path):
    """viz/<token>.png (six-view scenes, callback.py:76-86) and sample/<token>/bev.png (rand_str runs, callback.py:105): the class-colour rule of
    visualize.py:67-107 on a hand-made map, and the sheet's geometry."""
    from PIL import Image
    from bevgen_amd.writer import contact_sheet, render_bev

    seg = np.zeros((7, 8, 8), dtype=np.float32)   # (channel-first is recognised by c < h == w, as in the reference)
    seg[5, 0, 0] = 1.0            # channel 5 -> slot 1 after the channel shuffle [4,5,6,3,1,0,2]: lane_divider grey
    seg[0, 1, 1] = 1.0            # channel 0 -> slot 5: vehicle orange
    seg[0, 2, 2] = seg[1, 2, 2] = 1.0   # a tie: the higher slot wins; channel 0 -> slot 5, channel 1 -> slot 4
    seg[2, 3, 3] = 0.5            # channel 2 -> slot 6 (large vehicle) at half confidence: blended with the empty colour
    rgb = render_bev(seg)
    assert rgb.shape == (8, 8, 3) and rgb.dtype == np.uint8
    assert tuple(rgb[0, 0]) == (130, 130, 130) and tuple(rgb[1, 1]) == (255, 158, 0) and tuple(rgb[2, 2]) == (255, 158, 0)
    assert tuple(rgb[3, 3]) == (227, 149, 135)                     # 0.5 (255, 99, 71) + 0.5 (200, 200, 200)
    assert tuple(rgb[0, 3]) == (200, 200, 200)                     # nothing there
    assert np.array_equal(render_bev((seg * 255).astype(np.uint8)), rgb)
    # non-square grids and channel-last input: the channel axis is explicit or the smallest one, never guessed from squareness
    rect = np.zeros((7, 6, 10), dtype=np.float32)
    rect[0, 1, 9] = 1.0
    assert render_bev(rect).shape == (6, 10, 3) and tuple(render_bev(rect)[1, 9]) == (255, 158, 0)
    assert np.array_equal(render_bev(rect.transpose(1, 2, 0), channel_axis=-1), render_bev(rect, channel_axis=0))

    cams = ["CAM_FRONT_LEFT", "CAM_FRONT", "CAM_FRONT_RIGHT", "CAM_BACK_LEFT", "CAM_BACK", "CAM_BACK_RIGHT"]
    batch, out = _batch(2, cams), _outputs(2, 6)
    with SceneWriter(str(tmp_path), rand_str=True, seed=1) as w:
        toks = w.write(out, batch, dataset=Dataset.NUSCENES)
    for b, tok in enumerate(toks):
        sheet = np.asarray(Image.open(tmp_path / "viz" / f"{batch['sample_token'][b]}.png"))
        assert sheet.shape == (2 * 24, 6 * 40 + 24, 3)             # generated row over ground-truth row, BEV tile at the right
        gen_u8 = (out["gen"][b] * 255).round().clamp(0, 255).to(torch.uint8).numpy()
        assert np.abs(sheet[:24, 40:80].astype(int) - gen_u8[1].transpose(1, 2, 0).astype(int)).max() <= 1
        bev = np.asarray(Image.open(tmp_path / "sample" / tok / "bev.png"))
        assert np.array_equal(bev, render_bev(batch["segmentation"][b].float().numpy()))
    # three-camera scenes get no sheet (the reference builds it for six views only)
    w2 = SceneWriter(str(tmp_path / "c3"))
    w2.write(_outputs(1, 3), _batch(1, cams[:3]), dataset=Dataset.NUSCENES)
    w2.close()
    assert not (tmp_path / "c3" / "viz").exists() and (tmp_path / "c3" / "sample" / "tok000" / "bev.png").exists()   # bev.png: nuScenes (callback.py:117-118)
    assert contact_sheet(gen_u8, None, render_bev(seg)).shape == (24, 6 * 40 + 24, 3)

"""Hydra-free ``generate.py`` (reference entry point generate.py:26-77): compose the configuration tree, build the model through the drop-in classes,
run ``forward(batch) -> {'gen','rec','gt'}`` over the test batches and hand the results to the ``GenerateImages`` writer.

    python -m bevgen_amd.generate --config-dir /path/to/reference/configs \\
        experiment=muse_stage_two_multi_view datamodule=stage_2_argoverse_generate 'modes=[argoverse,generate]' \\
        datamodule.batch_size=16 'datamodule.test.eval_generate="/out"'

is the README's inference command with ``python generate.py`` replaced.  Differences from the reference entry point, all forced by the deployment image:
* no Lightning Trainer: one process per GPU (``torch.distributed.run``), rank r takes the batches r, r+world, … and writes its own files
  (what the reference's DDP + DistributedSampler does, configs/modes/generate.yaml:17-18);
* the dataset classes (nuScenes / Argoverse 2 devkits + data) are out of scope: batches come from ``--batches file.pt`` (a list of collated batch
  dicts saved with torch.save - exactly what the reference DataLoader yields) or ``--synthetic N`` (seeded BEV token ids + ring cameras);
* ``--random-weights`` nulls every ``ckpt_path`` (deterministic generated weights) for smoke runs without the released checkpoints.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import Any, Dict, Iterable, List

import torch

from . import hydra_lite, synthetic


def _null_ckpts(node: Any) -> None:
    if isinstance(node, dict):
        for k in list(node):
            if k == "ckpt_path":
                node[k] = None
            else:
                _null_ckpts(node[k])
    elif isinstance(node, list):
        for v in node:
            _null_ckpts(v)


def _cfg_nodes(node: Any, target: Any) -> List[dict]:
    """every GPTConfig node of the model tree (the config is interpolated into several places: model.cfg, model.maskgit.transformer.cfg, ...)"""
    out: List[dict] = []
    if isinstance(node, dict):
        if node.get("_target_") == target and "num_cams" in node:
            out.append(node)
        for v in node.values():
            out.extend(_cfg_nodes(v, target))
    elif isinstance(node, list):
        for v in node:
            out.extend(_cfg_nodes(v, target))
    return out


def _synthetic_batches(cfg, n_scenes: int, batch_size: int, seed: int) -> Iterable[Dict[str, Any]]:
    done = 0
    while done < n_scenes:
        b = min(batch_size, n_scenes - done)
        batch = synthetic.make_batch(cfg, b, seed=seed + done)
        batch["sample_token"] = [f"synthetic_{done + i:06d}" for i in range(b)]
        batch["cam_name"] = [[name] * b for name in cfg.cam_names.value]
        H, W = cfg.bev_latent_res[0] * 16, cfg.bev_latent_res[1] * 16
        batch["segmentation"] = torch.zeros(b, H, W, 1, dtype=torch.uint8)   # placeholder condition image for bev.npz (the ids are given directly)
        yield batch
        done += b


def _to_device(batch: Dict[str, Any], device) -> Dict[str, Any]:
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def _config_backend():
    """Which composer / instantiator runs the configuration tree (reference generate.py:75-77 uses @hydra.main): REAL Hydra when it is importable - the drop-in classes are
    ordinary `_target_`s, nothing here needs the built-in loader - else `bevgen_amd.hydra_lite` (the subset of Hydra the reference's tree uses; this image ships no Hydra).
    $BEVGEN_HYDRA = lite | real pins the choice."""
    want = os.environ.get("BEVGEN_HYDRA", "").lower()
    if want != "lite":
        try:
            import hydra  # noqa: F401
            import omegaconf  # noqa: F401
            return "real"
        except ImportError:
            if want == "real":
                raise
    return "lite"


def _compose(config_dir: str, config_name: str, overrides: List[str]):
    """-> (plain nested dict, instantiate function)."""
    if _config_backend() == "real":
        import hydra
        from hydra import compose, initialize_config_dir
        from omegaconf import OmegaConf

        with initialize_config_dir(config_dir=os.path.abspath(config_dir), version_base="1.2"):
            cfg = compose(config_name=config_name, overrides=list(overrides), return_hydra_config=False)
        return OmegaConf.to_container(cfg, resolve=True), (lambda node: hydra.utils.instantiate(node, _convert_="all"))
    return hydra_lite.compose(config_dir, config_name, overrides), hydra_lite.instantiate


def main(argv: List[str] = None) -> int:
    ap = argparse.ArgumentParser(prog="python -m bevgen_amd.generate", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config-dir", required=True, help="the configuration tree (e.g. <reference>/configs)")
    ap.add_argument("--config-name", default="train.yaml")
    ap.add_argument("--synthetic", type=int, default=0, metavar="N", help="generate N synthetic scenes instead of reading --batches")
    ap.add_argument("--batches", default=None, help="torch.save'd list of collated batch dicts")
    ap.add_argument("--random-weights", action="store_true", help="ignore every ckpt_path (deterministic generated weights)")
    ap.add_argument("--synthetic-calibration", action="store_true",
                    help="legacy_prob_matrix=false needs pretrained/cam_data_<dataset>.pt (the rig calibration the reference's dataset class writes); "
                         "use a synthetic ring rig instead when that file is absent (smoke runs)")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--print-config", action="store_true", help="print the composed configuration and exit")
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args(argv)

    cfg, instantiate = _compose(args.config_dir, args.config_name, args.overrides)
    if args.random_weights:
        _null_ckpts(cfg.get("model"))
    if args.synthetic_calibration:
        gcfg = (cfg.get("model") or {}).get("cfg") or ((cfg.get("model") or {}).get("transformer") or {}).get("cfg")
        if isinstance(gcfg, dict) and not gcfg.get("legacy_prob_matrix", True):
            name = str(gcfg.get("dataset", "NUSCENES")).lower()
            if not os.path.exists(os.path.join("pretrained", f"cam_data_{name}.pt")):
                intr, extr = synthetic.rig_calibration(int(gcfg["num_cams"]))
                for node in _cfg_nodes(cfg.get("model"), gcfg.get("_target_")):
                    node["cam_intrinsics"], node["cam_extrinsics"] = intr, extr
    if args.print_config:
        import yaml

        def plain(v):
            if isinstance(v, torch.Tensor):
                return v.tolist()
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, list):
                return [plain(x) for x in v]
            return v
        yaml.safe_dump(plain({k: v for k, v in cfg.items() if k != "hydra"}), sys.stdout, sort_keys=False)
        return 0

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise RuntimeError("bevgen_amd.generate needs a ROCm GPU (MI355X / gfx950); there is no CPU path in the product")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    seed = cfg.get("seed", 0) if args.seed is None else args.seed
    torch.manual_seed(int(seed or 0))

    model = instantiate(cfg["model"]).to(device).eval()
    callbacks = [instantiate(c) for c in (cfg.get("callbacks") or {}).values() if isinstance(c, dict) and "GenerateImages" in str(c.get("_target_", ""))]
    if not callbacks:
        raise hydra_lite.ConfigError("no GenerateImages callback in the configuration (callbacks.image_logger)")

    batch_size = int((cfg.get("datamodule") or {}).get("batch_size", 1))
    if args.batches:
        batches: Iterable[Dict[str, Any]] = torch.load(args.batches)
    elif args.synthetic > 0:
        batches = _synthetic_batches(model.cfg, args.synthetic, batch_size, int(seed or 0))
    else:
        raise hydra_lite.ConfigError("pass --batches FILE or --synthetic N: the dataset classes of the reference are not part of this package")

    class _Trainer:   # what the callback reads from the Lightning trainer
        log_dir = cfg.get("paths", {}).get("output_dir", ".") if isinstance(cfg.get("paths"), dict) else "."
        global_rank = rank

    n = 0
    with torch.no_grad():
        for i, batch in enumerate(batches):
            if i % world != rank:
                continue
            outputs = model(_to_device(batch, device))
            for cb in callbacks:
                cb.on_test_batch_end(_Trainer, model, outputs, batch, i)
            n += outputs["gen"].shape[0]
    for cb in callbacks:
        cb.on_test_end(_Trainer, model)
    print(f"[bevgen_amd.generate] rank {rank}/{world}: wrote {n} scenes to {callbacks[0].save_dir or _Trainer.log_dir}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

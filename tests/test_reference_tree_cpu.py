"""The entry point against the REAL configuration tree of the reference (build container only: /root/reference is absent on the GPU box, so these tests skip there).
hydra_lite's own tests use a synthetic tree; this pins that the README's inference command (README.md:65-72) and the Route-A model (configs/model/stage_2.yaml)
still compose, that every `_target_` on the path is rewritten to a drop-in class that exists, and that the library's mode keys ride on the same mechanism."""
import inspect
import os

import pytest

from bevgen_amd import hydra_lite as H

REF_CONFIGS = "/root/reference/configs"
pytestmark = [pytest.mark.reference, pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="needs /root/reference (build container only)")]

README_COMMAND = ["experiment=muse_stage_two_multi_view", "datamodule=stage_2_argoverse_generate", "modes=[argoverse,generate]", "trainer.devices=1",
                  "extras.mini_dataset=False", "datamodule.batch_size=16", 'datamodule.test.eval_generate="/tmp/bevgen_out"']


def _located(target):
    return H.locate(H.rewrite_target(target))


def test_readme_inference_command_composes_against_the_reference_tree():
    cfg = H.compose(REF_CONFIGS, "train.yaml", README_COMMAND)
    m = cfg["model"]
    from bevgen_amd.modules.stage1.vqgan import VQModel, VQSegmentationModel
    from bevgen_amd.modules.stage2.cond_transformer_multi_view_muse import Net2NetTransformer
    from bevgen_amd.modules.stage2.muse_maskgit_pytorch import MaskGit, MaskGitTransformerMultiView
    from bevgen_amd.config import GPTConfig

    assert m["_target_"] == "multi_view_generation.modules.stage2.cond_transformer_multi_view_muse.Net2NetTransformer"
    assert _located(m["_target_"]) is Net2NetTransformer
    assert _located(m["maskgit"]["_target_"]) is MaskGit
    assert _located(m["maskgit"]["transformer"]["_target_"]) is MaskGitTransformerMultiView
    assert _located(m["first_stage"]["_target_"]) is VQModel
    assert _located(m["cond_stage"]["_target_"]) is VQSegmentationModel
    assert _located(m["cfg"]["_target_"]) is GPTConfig
    # the released Argoverse shape: 3 cameras, 256 x 256 -> 16 x 16 latents, 14 layers x 16 heads x 1024
    assert cfg["num_cams"] == 3 and m["cfg"]["num_cams"] == 3
    assert list(cfg["cam_res"]) == [256, 256] and list(cfg["cam_latent_res"]) == [16, 16]
    tr = m["maskgit"]["transformer"]
    assert (tr["depth"], tr["heads"], tr["dim"]) == (m["cfg"]["num_layers"], m["cfg"]["num_heads"], m["cfg"]["num_embed"]) == (14, 16, 1024)
    assert m["cfg"]["sparse_block_size"] == 1 and m["cfg"]["legacy_prob_matrix"] is False and m["cfg"]["camera_bias"] is True
    # sample_iterations is not set by any shipped YAML: the constructor default (muse_lm:55) decides, and the drop-in's default is the reference's 18
    assert "sample_iterations" not in m
    assert inspect.signature(Net2NetTransformer.__init__).parameters["sample_iterations"].default == 18
    # datamodule + the GenerateImages callback of modes/generate.yaml, with the output directory of the command line
    dm = cfg["datamodule"]
    assert dm["_target_"] == "multi_view_generation.dataloader.DataModuleFromConfig" and dm["batch_size"] == 16
    assert dm["test"]["_target_"] == "multi_view_generation.bev_utils.argoverse.Argoverse" and dm["test"]["eval_generate"] == "/tmp/bevgen_out"
    cb = cfg["callbacks"]["image_logger"]
    assert cb["_target_"] == "multi_view_generation.utils.GenerateImages" and cb["save_dir"] == "/tmp/bevgen_out"
    from bevgen_amd.writer import GenerateImages
    assert _located(cb["_target_"]) is GenerateImages
    # modes/generate.yaml: test split only, DDP (one process per GPU, no result exchange: generate.yaml:17-18); generate.py:61-62 always runs trainer.test
    assert cfg["task_name"] == "generate" and dm["train"] is None and dm["validation"] is None and cfg["trainer"]["strategy"] == "ddp" and cfg["trainer"]["devices"] == 1


def test_route_a_model_composes_and_takes_the_mode_keys():
    # train.yaml's default datamodule (stage_1_nuscenes) is not in the released tree: name one that is
    cfg = H.compose(REF_CONFIGS, "train.yaml", ["model=stage_2", "datamodule=default", "+model.transformer.kv_cache=f16", "+model.transformer.decode_weights=f16",
                                                "+model.precision=f16x3"])
    m = cfg["model"]
    from bevgen_amd.modules.stage1.vqgan import VQModel, VQSegmentationModel
    from bevgen_amd.modules.stage2.cond_transformer_multi_view import Net2NetTransformer
    from bevgen_amd.modules.transformer.mingpt_sparse import GPT, GPTConfig

    assert _located(m["_target_"]) is Net2NetTransformer
    assert _located(m["transformer"]["_target_"]) is GPT
    assert _located(m["transformer"]["cfg"]["_target_"]) is GPTConfig
    assert _located(m["first_stage"]["_target_"]) is VQModel and _located(m["cond_stage"]["_target_"]) is VQSegmentationModel
    c = m["transformer"]["cfg"]
    assert (c["num_layers"], c["num_heads"], c["hidden_size"], c["sparse_block_size"], c["window_len"], c["density"]) == (24, 16, 1024, 16, 32, 1.0)
    assert m["transformer"]["kv_cache"] == "f16" and m["transformer"]["decode_weights"] == "f16" and m["precision"] == "f16x3"
    # every key of the model node is accepted by the drop-in constructor (named or through **kwargs, like the reference's)
    sig = inspect.signature(Net2NetTransformer.__init__)
    assert any(p.kind is inspect.Parameter.VAR_KEYWORD for p in sig.parameters.values())
    assert any(p.kind is inspect.Parameter.VAR_KEYWORD for p in inspect.signature(GPT.__init__).parameters.values())


def test_print_config_entry_point_on_the_reference_tree(capsys):
    from bevgen_amd import generate

    rc = generate.main(["--config-dir", REF_CONFIGS, "--print-config"] + README_COMMAND)
    out = capsys.readouterr().out
    assert rc == 0 and "multi_view_generation.modules.stage2.cond_transformer_multi_view_muse.Net2NetTransformer" in out and "batch_size: 16" in out and "save_dir: /tmp/bevgen_out" in out

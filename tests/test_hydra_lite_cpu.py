"""Hydra-free composition (bevgen_amd/hydra_lite.py) on a synthetic config tree that uses every rule the reference's tree uses
(configs/train.yaml defaults list, `# @package _global_`, nested defaults, `override /group`, list-valued group selection, interpolation)."""
import os
import textwrap

import pytest

from bevgen_amd import hydra_lite as H


def _tree(tmp_path):
    files = {
        "main.yaml": """
            # @package _global_
            defaults:
              - _self_
              - data: small
              - model: base.yaml
              - paths: default
              - hydra: default
              - experiment: null
              - modes: null
            task_name: default
            res: [8, 8]
            num_cams: 6
            config_name: ${hydra:job.name}
        """,
        "data/small.yaml": "batch_size: 2\nroot: ${paths.data_dir}/small\n",
        "data/big.yaml": "batch_size: 64\nroot: ${paths.data_dir}/big\n",
        "model/base.yaml": """
            _target_: collections.OrderedDict
            depth: 4
            res: ${res}
            head:
              _target_: fractions.Fraction
              numerator: 3
              denominator: ${model.depth}
        """,
        "model/wide.yaml": """
            defaults:
              - base
              - _self_
            depth: 8
            width: 1024
        """,
        "paths/default.yaml": "data_dir: ${oc.env:BEVGEN_TEST_DATA,/data}\nout: ${hydra:runtime.output_dir}\nsecret: ${oc.env:BEVGEN_UNSET_VARIABLE}\n",
        "hydra/default.yaml": "defaults:\n  - override hydra_logging: colorlog\nrun:\n  dir: /runs/${task_name}\n",
        "experiment/exp1.yaml": """
            # @package _global_
            defaults:
              - override /data: missing_option
              - override /model: wide
            config_name: exp_one
            res: [14, 25]
            model:
              depth: 14
        """,
        "modes/a.yaml": "# @package _global_\ndefaults:\n  - override /data: big\nnum_cams: 3\ntags: [a]\n",
        "modes/b.yaml": "# @package _global_\ntask_name: generate\ndata:\n  extra: ${data.batch_size}\n",
    }
    for rel, text in files.items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(textwrap.dedent(text).lstrip("\n"))
    return str(tmp_path)


def test_defaults_only(tmp_path):
    cfg = H.compose(_tree(tmp_path), "main.yaml")
    assert cfg["data"] == {"batch_size": 2, "root": "/data/small"}
    assert cfg["model"]["depth"] == 4 and cfg["model"]["res"] == [8, 8] and cfg["model"]["head"]["denominator"] == 4
    assert cfg["config_name"] == "main" and cfg["paths"]["out"] == "/runs/default"
    assert cfg["paths"]["secret"] == "${oc.env:BEVGEN_UNSET_VARIABLE}"      # unset env vars only fail when used (Hydra resolves lazily)


def test_readme_style_command(tmp_path):
    cfg = H.compose(_tree(tmp_path), "main.yaml", ["experiment=exp1", "modes=[a,b]", "data.batch_size=16", 'paths.data_dir="/mnt/x"', "+model.extra=true"])
    # experiment's `override /data: missing_option` is superseded by the later mode `a` (last override wins) -> the missing file is never opened
    assert cfg["hydra"]["runtime"]["choices"]["data"] == "big" and cfg["hydra"]["runtime"]["choices"]["model"] == "wide"
    assert cfg["data"] == {"batch_size": 16, "root": "/mnt/x/big", "extra": 16}
    # model/wide.yaml: defaults [base, _self_] then the experiment's own keys (experiment comes later in the primary defaults list)
    assert cfg["model"]["depth"] == 14 and cfg["model"]["width"] == 1024 and cfg["model"]["res"] == [14, 25] and cfg["model"]["extra"] is True
    assert cfg["num_cams"] == 3 and cfg["task_name"] == "generate" and cfg["config_name"] == "exp_one" and cfg["tags"] == ["a"]
    assert cfg["paths"]["out"] == "/runs/generate"


def test_cli_group_choice_beats_config_override(tmp_path):
    cfg = H.compose(_tree(tmp_path), "main.yaml", ["experiment=exp1", "data=small", "modes=a"])
    assert cfg["data"]["batch_size"] == 2


def test_missing_final_choice_is_an_error(tmp_path):
    with pytest.raises(H.ConfigError, match="missing_option"):
        H.compose(_tree(tmp_path), "main.yaml", ["experiment=exp1"])


def test_env_and_delete(tmp_path, monkeypatch):
    monkeypatch.setenv("BEVGEN_TEST_DATA", "/env/data")
    cfg = H.compose(_tree(tmp_path), "main.yaml", ["~model.head"])
    assert cfg["data"]["root"] == "/env/data/small" and "head" not in cfg["model"]


def test_instantiate_recursive_and_rewrite(tmp_path):
    cfg = H.compose(_tree(tmp_path), "main.yaml")
    m = H.instantiate(cfg["model"])
    from collections import OrderedDict
    from fractions import Fraction
    assert isinstance(m, OrderedDict) and m["head"] == Fraction(3, 4) and m["depth"] == 4
    assert H.rewrite_target("multi_view_generation.modules.stage2.muse_maskgit_pytorch.MaskGit") == "bevgen_amd.modules.stage2.muse_maskgit_pytorch.MaskGit"
    assert H.rewrite_target("multi_view_generation.utils.GenerateImages") == "bevgen_amd.writer.GenerateImages"
    assert H.rewrite_target("torch.nn.Linear") == "torch.nn.Linear"
    w = H.instantiate({"_target_": "multi_view_generation.utils.GenerateImages", "save_dir": str(tmp_path), "rand_str": True})
    assert type(w).__name__ == "GenerateImages" and w.rand_str is True
    p = H.instantiate({"_target_": "fractions.Fraction", "_partial_": True, "numerator": 1})
    assert p(denominator=2) == Fraction(1, 2)


def test_generate_uses_real_hydra_when_installed_and_hydra_lite_otherwise(tmp_path, monkeypatch):
    """reference generate.py:75-77 composes through @hydra.main.  bevgen_amd.generate hands the tree to REAL Hydra when it is importable (the drop-in classes are ordinary
    `_target_`s) and to hydra_lite otherwise; this image has no Hydra, so the real branch is exercised against a stand-in module that records the calls."""
    import sys
    import types

    from bevgen_amd import generate as G

    monkeypatch.delenv("BEVGEN_HYDRA", raising=False)
    for m in ("hydra", "omegaconf"):
        monkeypatch.delitem(sys.modules, m, raising=False)
    assert G._config_backend() == "lite"
    monkeypatch.setenv("BEVGEN_HYDRA", "real")
    with pytest.raises(ImportError):
        G._config_backend()
    calls = {}
    hydra = types.ModuleType("hydra")

    class _Init:
        def __init__(self, config_dir, version_base):
            calls["config_dir"], calls["version_base"] = config_dir, version_base

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    hydra.initialize_config_dir = _Init
    hydra.compose = lambda config_name, overrides, return_hydra_config=False: calls.setdefault("compose", (config_name, list(overrides))) and {"model": {"_target_": "builtins.dict", "a": 1}}
    hydra.utils = types.SimpleNamespace(instantiate=lambda node, _convert_=None: ("instantiated", node, _convert_))
    omegaconf = types.ModuleType("omegaconf")
    omegaconf.OmegaConf = types.SimpleNamespace(to_container=lambda cfg, resolve=True: dict(cfg, resolved=resolve))
    monkeypatch.setitem(sys.modules, "hydra", hydra)
    monkeypatch.setitem(sys.modules, "omegaconf", omegaconf)
    assert G._config_backend() == "real"
    cfg, inst = G._compose(str(tmp_path), "train.yaml", ["experiment=muse_stage_two_multi_view"])
    assert calls["config_dir"] == str(tmp_path) and calls["version_base"] == "1.2" and calls["compose"] == ("train.yaml", ["experiment=muse_stage_two_multi_view"])
    assert cfg["resolved"] is True and inst(cfg["model"]) == ("instantiated", cfg["model"], "all")
    monkeypatch.setenv("BEVGEN_HYDRA", "lite")
    assert G._config_backend() == "lite"

"""Hydra-free loader for the reference's configuration tree.

The reference is driven by ``python generate.py experiment=… datamodule=… 'modes=[argoverse,generate]' key=value …`` over a Hydra 1.2 config
directory (generate.py:75-77, configs/train.yaml:5-22).  Hydra / OmegaConf are not available in the deployment image, so this module re-implements
the subset of the composition rules that tree uses; when pointed at the reference's ``configs/`` it yields the same resolved dictionary:

* primary config + ``defaults`` list; group entries ``group: option`` load ``<dir>/<group>/<option>.yaml`` into package ``group``;
  ``# @package _global_`` headers merge at the root; ``null`` options are skipped; ``_self_`` fixes where a file's own keys merge (last by default);
* nested defaults inside group files (``- stage_2`` = sibling file, ``- sub: opt`` = sub-group);
* ``- override /group: option`` entries of later files (experiment, modes) re-select an earlier group; a command-line ``group=option`` beats them;
  ``group=[a,b]`` selects several options (merged in order); ``+group=option`` appends a group that is not in the defaults list;
* value overrides ``a.b.c=value`` (YAML-typed), ``+a.b=value`` (add), ``~a.b`` (delete), applied after composition, "last wins";
* interpolation ``${a.b}`` (absolute, node or string-embedded), ``${oc.env:VAR[,default]}``, ``${hydra:job.name}``, ``${hydra:runtime.output_dir}``,
  ``${hydra:runtime.cwd}``;
* ``instantiate``: recursive ``_target_`` construction (``_partial_`` and ``_args_`` supported), with a prefix table that redirects the reference's
  class paths to this package's drop-ins.

If the real ``hydra`` is importable, ``compose(..., prefer_hydra=True)`` delegates to it.
"""
from __future__ import annotations

import copy
import functools
import importlib
import os
import re
from datetime import datetime
from pathlib import Path
from typing import Any, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import yaml

# reference class path prefix -> drop-in (longest prefix wins)
TARGET_REWRITES: Dict[str, str] = {
    "multi_view_generation.modules.stage2.": "bevgen_amd.modules.stage2.",
    "multi_view_generation.modules.stage1.vqgan.": "bevgen_amd.modules.stage1.vqgan.",
    "multi_view_generation.modules.transformer.mingpt_sparse.": "bevgen_amd.modules.transformer.mingpt_sparse.",
    "multi_view_generation.modules.transformer.sparse_self_attention.": "bevgen_amd.modules.transformer.sparse_self_attention.",
    "multi_view_generation.modules.losses.vqperceptual.DummyLoss": "bevgen_amd.modules.losses.DummyLoss",
    "multi_view_generation.utils.GenerateImages": "bevgen_amd.writer.GenerateImages",
    "multi_view_generation.utils.callback.GenerateImages": "bevgen_amd.writer.GenerateImages",
}


class ConfigError(ValueError):
    pass


# ---------------------------------------------------------------------------------------------------------------------- files
_PACKAGE_RE = re.compile(r"^\s*#\s*@package\s+(\S+)\s*$")


def _load_file(path: Path) -> Tuple[Optional[str], List[Any], Dict[str, Any]]:
    """-> (package header or None, defaults list, body without `defaults`)"""
    if not path.exists():
        raise ConfigError(f"config file not found: {path}")
    text = path.read_text()
    package = None
    for line in text.splitlines():
        if not line.strip():
            continue
        m = _PACKAGE_RE.match(line)
        if m:
            package = m.group(1)
        if not line.lstrip().startswith("#"):
            break
    body = yaml.safe_load(text) or {}
    if not isinstance(body, dict):
        raise ConfigError(f"{path}: top level must be a mapping")
    defaults = body.pop("defaults", None) or []
    return package, list(defaults), body


def _cfg_path(root: Path, group: str, option: str) -> Path:
    name = option[:-5] if option.endswith(".yaml") else option
    return root / group / f"{name}.yaml" if group else root / f"{name}.yaml"


def _as_options(v) -> List[str]:
    if v is None:
        return []
    if isinstance(v, (list, tuple)):
        return [str(x) for x in v if x is not None]
    return [str(v)]


def deep_merge(dst: Dict[str, Any], src: Mapping[str, Any]) -> Dict[str, Any]:
    """OmegaConf.merge semantics for plain containers: mappings merge recursively, everything else (incl. lists) is replaced."""
    for k, v in src.items():
        if isinstance(v, Mapping) and isinstance(dst.get(k), dict):
            deep_merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _merge_at(root: Dict[str, Any], package: str, body: Mapping[str, Any]) -> None:
    node = root
    if package and package != "_global_":
        for part in package.replace("/", ".").split("."):
            nxt = node.get(part)
            if not isinstance(nxt, dict):
                nxt = {}
                node[part] = nxt
            node = nxt
    deep_merge(node, body)


# ---------------------------------------------------------------------------------------------------------------------- composition
def _parse_default_entry(entry) -> Tuple[str, Optional[str], Any]:
    """-> (kind, group, option); kind in {'self', 'file', 'group', 'override'}"""
    if isinstance(entry, str):
        return ("self", None, None) if entry == "_self_" else ("file", None, entry)
    if isinstance(entry, dict) and len(entry) == 1:
        (k, v), = entry.items()
        k = str(k).strip()
        if k.startswith("override "):
            return "override", k[len("override "):].strip().lstrip("/"), v
        if k.startswith("optional "):
            k = k[len("optional "):].strip()
        return "group", k.lstrip("/"), v
    raise ConfigError(f"unsupported defaults entry: {entry!r}")


def _split_overrides(overrides: Sequence[str], groups: Iterable[str]):
    """Command line -> (group selections {group: option | [options]}, appended groups, value overrides).  A key is a GROUP selection when it has
    no dot and names a sub-directory of the config dir (Hydra's rule); everything else is a value override."""
    group_sel: Dict[str, Any] = {}
    group_add: List[Tuple[str, Any]] = []
    values: List[Tuple[str, str, Any]] = []   # (op, key, value)
    groups = set(groups)
    for ov in overrides:
        ov = ov.strip()
        if not ov:
            continue
        if ov.startswith("~"):
            values.append(("del", ov[1:].split("=")[0], None))
            continue
        if "=" not in ov:
            raise ConfigError(f"override without '=': {ov!r}")
        key, raw = ov.split("=", 1)
        add = key.startswith("+")
        key = key.lstrip("+")
        val = yaml.safe_load(raw) if raw != "" else ""
        if "." not in key and key in groups:
            (group_add.append((key, val)) if add else group_sel.__setitem__(key, val))
        else:
            values.append(("add" if add else "set", key, val))
    return group_sel, group_add, values


def compose(config_dir: str, config_name: str = "train.yaml", overrides: Sequence[str] = (), *, resolve: bool = True, job_name: Optional[str] = None,
            output_dir: Optional[str] = None, prefer_hydra: bool = False) -> Dict[str, Any]:
    """Compose the configuration exactly like ``@hydra.main(config_path=config_dir, config_name=config_name)`` + command-line `overrides`."""
    if prefer_hydra:
        try:
            from hydra import compose as _hc, initialize_config_dir
            from omegaconf import OmegaConf
            with initialize_config_dir(config_dir=str(Path(config_dir).resolve()), version_base="1.2"):
                return OmegaConf.to_container(_hc(config_name=config_name, overrides=list(overrides)), resolve=resolve)
        except ImportError:
            pass
    root = Path(config_dir)
    _, primary_defaults, primary_body = _load_file(_cfg_path(root, "", config_name))
    parsed = [_parse_default_entry(e) for e in primary_defaults]
    group_order = [g for kind, g, _ in parsed if kind == "group"]
    group_sel, group_add, values = _split_overrides(overrides, [d.name for d in root.iterdir() if d.is_dir()])

    choices: Dict[str, Any] = {g: opt for kind, g, opt in parsed if kind == "group"}
    cli_fixed = set()
    for g, v in group_sel.items():
        if g not in choices:
            group_add.append((g, v))
        else:
            choices[g] = v
        cli_fixed.add(g)
    for g, v in group_add:
        if g not in choices:
            group_order.append(g)
            parsed.append(("group", g, v))
        choices[g] = v
        cli_fixed.add(g)

    # pass 1: collect `override /group: option` entries of the selected files (later files win, the command line beats them all)
    for _ in range(8):
        changed = False
        for g in group_order:
            for opt in _as_options(choices.get(g)):
                path = _cfg_path(root, g, opt)
                if not path.exists():
                    continue   # may be re-selected by a later override; pass 2 reports a missing final choice
                for kind, og, oopt in map(_parse_default_entry, _load_file(path)[1]):
                    if kind == "override" and og not in cli_fixed and choices.get(og) != oopt:
                        if og not in choices:
                            if g == "hydra":
                                continue   # Hydra's own groups (hydra_logging, job_logging): logging set-up of the real Hydra, nothing to compose
                            raise ConfigError(f"{path}: override of group '{og}' which is not in the defaults list")
                        choices[og] = oopt
                        changed = True
        if not changed:
            break

    # pass 2: merge in defaults-list order
    cfg: Dict[str, Any] = {}

    def merge_config(group: str, option: str, package: Optional[str]) -> None:
        header, defaults, body = _load_file(_cfg_path(root, group, option))
        pkg = header if header is not None else (package if package is not None else group)
        entries = [_parse_default_entry(e) for e in defaults]
        if not any(k == "self" for k, _, _ in entries):
            entries.append(("self", None, None))      # version_base >= 1.1: a config overrides its defaults unless it says otherwise
        for kind, g, opt in entries:
            if kind == "self":
                _merge_at(cfg, pkg, body)
            elif kind == "file":
                merge_config(group, str(opt), pkg)
            elif kind == "group":
                for o in _as_options(opt):
                    sub = f"{group}/{g}" if group else g
                    merge_config(sub, o, None if header is None and package is None else f"{pkg}.{g}" if pkg != "_global_" else g)
            # 'override': handled in pass 1

    if not any(k == "self" for k, _, _ in parsed):
        parsed.append(("self", None, None))
    for kind, g, opt in parsed:
        if kind == "self":
            _merge_at(cfg, "_global_", primary_body)
        elif kind == "group":
            for o in _as_options(choices.get(g)):
                merge_config(g, o, None)
        elif kind == "file":
            merge_config("", str(opt), "_global_")

    for op, key, val in values:
        _apply_value(cfg, op, key, val)

    cfg.setdefault("hydra", {})
    hy = cfg["hydra"] if isinstance(cfg["hydra"], dict) else {}
    job = job_name or Path(config_name).stem
    run_dir = hy.get("run", {}).get("dir") if isinstance(hy.get("run"), dict) else None   # configs/hydra/default.yaml: the run directory pattern
    out = output_dir or run_dir or os.path.join(os.getcwd(), "outputs", datetime.now().strftime("%Y-%m-%d_%H-%M-%S"))
    deep_merge(hy, {"job": {"name": job, "num": 0}, "runtime": {"output_dir": out, "cwd": os.getcwd(), "choices": {g: choices.get(g) for g in group_order}}})
    cfg["hydra"] = hy
    if resolve:
        cfg = resolve_interpolations(cfg)
    return cfg


def _apply_value(cfg: Dict[str, Any], op: str, key: str, val: Any) -> None:
    parts = key.split(".")
    node = cfg
    for p in parts[:-1]:
        if not isinstance(node.get(p), dict):
            if op == "del":
                return
            node[p] = {}
        node = node[p]
    if op == "del":
        node.pop(parts[-1], None)
    else:
        if op == "set" and parts[-1] not in node and len(parts) > 1 and not isinstance(node, dict):
            raise ConfigError(f"cannot set {key}")
        if isinstance(val, Mapping) and isinstance(node.get(parts[-1]), dict):
            deep_merge(node[parts[-1]], val)
        else:
            node[parts[-1]] = val


# ---------------------------------------------------------------------------------------------------------------------- interpolation
_INTERP = re.compile(r"\$\{([^${}]+)\}")


def select(cfg: Mapping[str, Any], dotted: str) -> Any:
    node: Any = cfg
    for p in dotted.split("."):
        if isinstance(node, Mapping):
            if p not in node:
                raise ConfigError(f"interpolation key '{dotted}' not found (missing '{p}')")
            node = node[p]
        elif isinstance(node, list):
            node = node[int(p)]
        else:
            raise ConfigError(f"interpolation key '{dotted}': '{p}' is not a container")
    return node


def resolve_interpolations(cfg: Dict[str, Any]) -> Dict[str, Any]:
    """Returns a copy with every ``${…}`` replaced (node interpolations keep the referenced node's type)."""
    cfg = copy.deepcopy(cfg)
    resolving: set = set()
    _now = datetime.now()

    def resolver(expr: str) -> Any:
        expr = expr.strip()
        if expr.startswith("oc.env:"):
            name, _, default = expr[len("oc.env:"):].partition(",")
            if name.strip() in os.environ:
                return os.environ[name.strip()]
            if default != "":
                return yaml.safe_load(default.strip())
            raise ConfigError(f"environment variable '{name.strip()}' is not set (needed by ${{{expr}}})")
        if expr.startswith("now:"):
            return _now.strftime(expr[len("now:"):])
        if expr.startswith("hydra:"):
            return res(select(cfg, "hydra." + expr[len("hydra:"):]), ("hydra", expr))
        return res(select(cfg, expr), ("key", expr))

    def res(v: Any, tag=None) -> Any:
        if tag is not None:
            if tag in resolving:
                raise ConfigError(f"interpolation cycle at {tag[1]}")
            resolving.add(tag)
        try:
            if isinstance(v, str):
                m = _INTERP.fullmatch(v.strip())
                if m and v.strip().count("${") == 1:
                    return copy.deepcopy(resolver(m.group(1)))
                prev = None
                while prev != v and "${" in v:
                    prev = v
                    v = _INTERP.sub(lambda mm: str(resolver(mm.group(1))), v)
                return v
            if isinstance(v, dict):
                return {k: res(x) for k, x in v.items()}
            if isinstance(v, list):
                return [res(x) for x in v]
            return v
        finally:
            if tag is not None:
                resolving.discard(tag)

    out: Dict[str, Any] = {}
    for k, v in cfg.items():
        out[k] = _resolve_tolerant(v, res)
    return out


def _resolve_tolerant(v: Any, res) -> Any:
    """Resolve a subtree; a leaf whose interpolation needs an unset environment variable keeps its raw text (Hydra resolves lazily: such leaves only
    fail when read, and the sampling path never reads e.g. ${oc.env:NUSCENES_DATA_DIR} of an unused datamodule)."""
    if isinstance(v, dict):
        return {k: _resolve_tolerant(x, res) for k, x in v.items()}
    if isinstance(v, list):
        return [_resolve_tolerant(x, res) for x in v]
    try:
        return res(v)
    except ConfigError as e:
        if "environment variable" in str(e):
            return v
        raise


# ---------------------------------------------------------------------------------------------------------------------- instantiate
def rewrite_target(target: str, table: Mapping[str, str] = TARGET_REWRITES) -> str:
    best = ""
    for prefix in table:
        if target.startswith(prefix) and len(prefix) > len(best):
            best = prefix
    return table[best] + target[len(best):] if best else target


def locate(path: str) -> Any:
    mod, _, name = path.rpartition(".")
    if not mod:
        raise ConfigError(f"cannot locate '{path}'")
    try:
        return getattr(importlib.import_module(mod), name)
    except (ImportError, AttributeError):
        # class attribute of a module-level class (a.b.Class.method)
        mod2, _, cls = mod.rpartition(".")
        if mod2:
            return getattr(getattr(importlib.import_module(mod2), cls), name)
        raise


def instantiate(node: Any, *args, _rewrites_: Optional[Mapping[str, str]] = None, **overrides) -> Any:
    """``hydra.utils.instantiate`` for plain containers: builds ``_target_`` nodes depth-first (``_recursive_`` default), honours ``_partial_`` and
    ``_args_``; keyword `overrides` are merged into the top node first.  Class paths go through TARGET_REWRITES."""
    table = TARGET_REWRITES if _rewrites_ is None else _rewrites_
    if node is None:
        return None
    if isinstance(node, list):
        return [instantiate(x, _rewrites_=table) for x in node]
    if not isinstance(node, Mapping):
        return node
    node = dict(node)
    node.update(overrides)
    if "_target_" not in node:
        return {k: instantiate(v, _rewrites_=table) for k, v in node.items()}
    target = rewrite_target(str(node.pop("_target_")), table)
    partial = bool(node.pop("_partial_", False))
    recursive = bool(node.pop("_recursive_", True))
    node.pop("_convert_", None)
    pos = list(node.pop("_args_", [])) + list(args)
    kwargs = {k: (instantiate(v, _rewrites_=table) if recursive else v) for k, v in node.items()}
    pos = [instantiate(v, _rewrites_=table) if recursive else v for v in pos]
    fn = locate(target)
    return functools.partial(fn, *pos, **kwargs) if partial else fn(*pos, **kwargs)

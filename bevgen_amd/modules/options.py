"""Arithmetic / storage modes of libbevgen_hip as seen from the drop-in modules.

The reference's plugin boundary is Hydra ``_target_`` instantiation (configs/model/stage_2.yaml:1-34, configs/experiment/muse_stage_two_multi_view.yaml:17-41);
every reference constructor on the path ends in ``**kwargs`` (gpt:270, ar_lm:55, muse_lm:56, muse_net:223), so a mode is selected the same way any other model
option is: a key next to the ``_target_`` (``+model.transformer.kv_cache=f16`` on the command line), or process-wide by an environment variable.

    key             env var                  values                 drop-in default
    precision       $BEVGEN_PRECISION        fp32 | f16x3           f16x3   (3 f16 MFMAs per product on hi/lo splits, fp32 accumulate; token-exact on every fixture)
    weights         $BEVGEN_WEIGHTS          f32 | f16              f32     (f16: GEMM / conv matrices rounded once at load, two MFMAs per product)
    kv_cache        $BEVGEN_KV_CACHE         f32 | f16              f32     (Route A; f16 = fp16 storage / fp32 accumulate: BASELINE config 4)
    decode_weights  $BEVGEN_DECODE_WEIGHTS   f32 | f16              f32     (Route A decode step streams 2-byte q/k/v, MLP and head weights)
    decode_path     $BEVGEN_DECODE_PATH      auto | fused | split | per_op   auto

Precedence: explicit constructor key > environment variable > default.  A key given to an outer module (``Net2NetTransformer``) is handed down to the
modules it owns unless they were given their own.
"""
from __future__ import annotations

import os
from typing import Dict, Mapping, MutableMapping, Optional

CHOICES = {
    "precision": ("fp32", "f16x3"),
    "weights": ("f32", "f16"),
    "kv_cache": ("f32", "f16"),
    "decode_weights": ("f32", "f16"),
    "decode_path": ("auto", "fused", "split", "per_op"),
}
ENV = {"precision": "BEVGEN_PRECISION", "weights": "BEVGEN_WEIGHTS", "kv_cache": "BEVGEN_KV_CACHE", "decode_weights": "BEVGEN_DECODE_WEIGHTS",
       "decode_path": "BEVGEN_DECODE_PATH"}
DROPIN_DEFAULTS = {"precision": "f16x3", "weights": "f32", "kv_cache": "f32", "decode_weights": "f32", "decode_path": "auto"}
ROUTE_KEYS = {"maskgit": ("precision", "weights"), "vq": ("precision", "weights"), "ar": tuple(CHOICES)}


def _check(key: str, value) -> str:
    value = str(value)
    if value not in CHOICES[key]:
        raise ValueError(f"{key} must be one of {CHOICES[key]}, got {value!r}")
    return value


def pop_runtime_options(kwargs: MutableMapping) -> Dict[str, str]:
    """Remove the mode keys from a constructor's ``**kwargs`` and return the explicit ones (validated)."""
    out = {}
    for key in CHOICES:
        if key in kwargs:
            v = kwargs.pop(key)
            if v is not None:
                out[key] = _check(key, v)
    return out


def resolve(explicit: Optional[Mapping[str, str]], route: str) -> Dict[str, str]:
    """explicit > environment > drop-in default, restricted to the keys the route's Context takes."""
    out = {}
    for key in ROUTE_KEYS[route]:
        if explicit and key in explicit:
            out[key] = _check(key, explicit[key])
        elif os.environ.get(ENV[key]):
            out[key] = _check(key, os.environ[ENV[key]])
        else:
            out[key] = DROPIN_DEFAULTS[key]
    return out


class RuntimeOptionsMixin:
    """Modules that own a Context: explicit options + inheritance from the owner."""

    _runtime_explicit: Dict[str, str]

    def _init_runtime_options(self, kwargs: MutableMapping) -> None:
        self._runtime_explicit = pop_runtime_options(kwargs)

    def set_runtime_options(self, inherit: bool = False, **opts) -> None:
        """Change modes after construction (drops the device context).  ``inherit=True``: keys this module was given itself win."""
        new = {k: _check(k, v) for k, v in opts.items() if v is not None}
        merged = dict(new, **self._runtime_explicit) if inherit else dict(self._runtime_explicit, **new)
        if merged != self._runtime_explicit:
            self._runtime_explicit = merged
            self.invalidate()

    def runtime_options(self, route: str) -> Dict[str, str]:
        return resolve(self._runtime_explicit, route)

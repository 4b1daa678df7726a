"""The C-ABI library builds, loads (no GPU needed) and exports every symbol include/bevgen_hip.h declares."""
import ctypes
import os
import re

import pytest

from bevgen_amd import _lib


def _declared_symbols():
    text = open(_lib.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bevgen_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    assert sorted(_lib.SIGNATURES) == declared


def test_library_loads_and_exports_all_symbols():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.bevgen_abi_version() == _lib.ABI_VERSION


def test_cfg_struct_layout_matches_header():
    # 3 + 5 + 3 + 3 + 3 + 2 + 8 + 8 + 1 + 16 int32 fields
    assert ctypes.sizeof(_lib.bevgen_cfg) == 4 * (3 + 5 + 3 + 3 + 3 + 2 + 8 + 8 + 1 + 16)


def test_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bevgen_amd.runtime import Context

    with pytest.raises(RuntimeError):
        Context(None)

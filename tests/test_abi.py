"""The C-ABI library: loads without a GPU and exports exactly what include/transvae_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "transvae_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tv_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from transvae.hip import _lib
    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_library_loads_and_exports_every_symbol():
    from transvae.hip import _lib
    _lib.build()
    lib = _lib.load()
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.tv_abi_version() == 1
    assert lib.tv_last_error() is not None


def test_ops_refuse_cpu_tensors():
    import torch
    from transvae.hip import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(32, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.group_norm_silu(torch.zeros(1, 4, 4, 32, dtype=torch.bfloat16), torch.ones(32), torch.zeros(32))

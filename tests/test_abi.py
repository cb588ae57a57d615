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


def test_activation_ids_agree():
    """TV_ACT_* in the public header, the kernels' common.h and the ctypes binding are the same numbers."""
    from transvae.hip import _lib

    def defines(path):
        return {k: int(v) for k, v in re.findall(r"#define\s+(TV_ACT_[A-Z_]+)\s+(\d+)", open(path).read())}
    pub = defines(os.path.join(ROOT, "include", "transvae_hip.h"))
    dev = defines(os.path.join(ROOT, "deepl-project_amd", "csrc", "common.h"))
    assert pub == dev and len(pub) == 6, (pub, dev)
    assert (pub["TV_ACT_NONE"], pub["TV_ACT_GELU"], pub["TV_ACT_SILU"]) == (_lib.ACT_NONE, _lib.ACT_GELU, _lib.ACT_SILU)
    assert (pub["TV_ACT_DERIV"], pub["TV_ACT_SAVE_DERIV"], pub["TV_ACT_ADD"]) == (_lib.ACT_DERIV, _lib.ACT_SAVE_DERIV, _lib.ACT_ADD)
    assert pub["TV_ACT_SAVE_DERIV"] & (pub["TV_ACT_GELU"] | pub["TV_ACT_SILU"] | pub["TV_ACT_DERIV"]) == 0


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

"""CPU: the C-ABI shared library loads and exports every symbol that include/cut3r_hip.h declares (no compute)."""
import os
import re

from cut3r_slam_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "cut3r_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long long)\s+(cut3r_\w+)\s*\(", src)))


def test_header_and_ctypes_table_agree():
    assert _declared() == sorted(_lib.SIGNATURES)


def test_library_loads_and_exports_every_symbol():
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.cut3r_abi_version() >= 1
    assert lib.cut3r_lc_workspace_floats(3, 5000) == 3 * 3 * 28


def test_bad_arguments_are_rejected_without_launching():
    import ctypes as C
    lib = _lib.load()
    d = _lib.GemmDesc()
    assert lib.cut3r_gemm_f16(C.byref(d), None) == 1           # null pointers
    assert lib.cut3r_rope2d(None, 0, None, 1, 1, 1, 4, 0, 0, 0, 100.0, 1.0, None) == 1
    assert lib.cut3r_attention_f16(None, None, None, None, 1, 1, 1, 1, 64, 0, 0, 0, 0, 0, 0, 0, 0, 1.0, None) == 1


def test_product_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from cut3r_slam_amd import ops
    with pytest.raises((RuntimeError, ValueError)):
        ops.rope_2d(torch.zeros(1, 2, 1, 4), torch.zeros(1, 2, 2, dtype=torch.int64), 100.0, 1.0)
    with pytest.raises(ValueError):
        ops.linear(torch.zeros(4, 8, dtype=torch.float16), torch.zeros(4, 8, dtype=torch.float16), torch.zeros(4, 4))

"""The C-ABI library loads (no GPU needed) and exports every symbol include/mm_hip.h declares."""
import ctypes
import os
import re

from multimeditron_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_parses_and_lib_exports_everything():
    protos = _lib.parse_header()
    assert len(protos) >= 35
    src = open(os.path.join(ROOT, "include", "mm_hip.h")).read()
    declared = set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", src))
    assert declared == set(protos), declared ^ set(protos)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(L, name), f"libmmhip.so does not export {name}"


def test_no_compute_without_gpu_raises():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from multimeditron_amd import kernels
    x = torch.zeros(4, 8)
    with pytest.raises(_lib.MMHipError):
        kernels.rmsnorm_fwd(x, torch.ones(8), 1e-5)


def test_argument_validation_without_launch():
    L = _lib.lib()
    assert L.mm_version() >= 100
    # invalid arguments are rejected before any launch (safe on a CPU-only machine)
    assert L.mm_gemm(0, 7, 1, 1, 1, None, 8, None, 8, None, 8, None, None, 0, 0, None) == -1
    assert L.mm_gemm(0, 0, 16, 16, 16, 16, 3, 16, 8, 16, 8, None, None, 0, 0, None) == -2   # lda not a multiple of 8
    assert L.mm_attn_fwd(0, 16, 16, 16, 1, 8, 8, 2, 1, 32, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, 0, 1.0, 16, 16, None) == -3
    assert L.mm_error_string(-2).decode().startswith("alignment")

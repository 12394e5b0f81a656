from typing import Union

import torch


def get_torch_dtype(dtype: Union[torch.dtype, str]) -> torch.dtype:
    """reference utils/__init__.py:7-11"""
    if isinstance(dtype, torch.dtype):
        return dtype
    d = getattr(torch, dtype)
    if not isinstance(d, torch.dtype):
        raise TypeError(f"{dtype!r} is not a torch dtype")
    return d


# ---- tracing: named ranges for rocprofv3 --marker-trace (the reference's NVTX callback, train/profiling.py:5-75) -----------
import contextlib
import os

_TRACE = bool(os.environ.get("MM_TRACE"))


@contextlib.contextmanager
def trace_range(name: str):
    """roctx range (torch.cuda.nvtx maps to roctx on ROCm) around a phase of the path when MM_TRACE=1; free otherwise."""
    if not _TRACE:
        yield
        return
    torch.cuda.nvtx.range_push(name)
    try:
        yield
    finally:
        torch.cuda.nvtx.range_pop()

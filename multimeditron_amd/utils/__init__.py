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

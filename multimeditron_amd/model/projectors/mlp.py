"""3-layer GELU MLP projector (reference projectors/mlp.py:4-59): Linear(d,d) -> GELU -> Linear(d,H) -> GELU ->
Linear(H,H), all biased, exact (erf) GELU.  State-dict names `projection.{0,2,4}.{weight,bias}` as in the reference.
Each Linear is one MFMA GEMM (bias in the epilogue); the GELU keeps its pre-activation for backward."""
import torch
import torch.nn as nn

from ..._lib import EPI_GELU_ERF
from ...nn import Linear


class _GELU(nn.Module):   # placeholder so that the Sequential indices match the reference (1 and 3)
    def forward(self, x):
        return x


class MLPProjector(nn.Module):
    def __init__(self, modality_size: int, projected_size: int, dtype: torch.dtype = torch.bfloat16, device=None):
        super().__init__()
        self.projection = nn.Sequential(
            Linear(modality_size, modality_size, dtype=dtype, device=device),
            _GELU(),
            Linear(modality_size, projected_size, dtype=dtype, device=device),
            _GELU(),
            Linear(projected_size, projected_size, dtype=dtype, device=device),
        )

    def forward(self, hidden_state: torch.Tensor) -> torch.Tensor:
        shape = hidden_state.shape
        x = hidden_state.reshape(-1, shape[-1])
        if not x.is_contiguous():
            x = x.contiguous()
        x = self.projection[0](x, act=EPI_GELU_ERF)
        x = self.projection[2](x, act=EPI_GELU_ERF)
        x = self.projection[4](x)
        return x.view(*shape[:-1], x.shape[-1])

"""Parameter holders with the reference's state-dict names, and parameter packing.

Modules here own Parameters only; the math is in functional.py (libmmhip kernels).  `pack_parameters` lays all
parameters of a model out in one flat buffer per dtype so that (a) q/k/v and gate/up weights are adjacent and act
as single fused GEMM operands, (b) gradients live in one flat buffer that doubles as the RCCL all-reduce buckets,
(c) AdamW is one launch per contiguous trainable range."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import functional as Fm


class Linear(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, dtype=None, device=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, dtype=dtype, device=device))
        self.bias = nn.Parameter(torch.empty(out_features, dtype=dtype, device=device)) if bias else None

    def forward(self, x2d, residual=None, act=0, ldc_pad=False):
        return Fm.linear(x2d, self.weight, self.bias, residual=residual, act=act, ldc_pad=ldc_pad,
                         dummy=grad_dummy(self.weight))

    def extra_repr(self):
        return f"in={self.in_features}, out={self.out_features}, bias={self.bias is not None}"


class Embedding(nn.Module):
    def __init__(self, num, dim, dtype=None, device=None):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num, dim
        self.weight = nn.Parameter(torch.empty(num, dim, dtype=dtype, device=device))

    def forward(self, ids, proj=None, batch_idx=None, token_range=None):
        """Lookup -> [*ids.shape, dim]; with `proj` [n, dim] the rows (batch_idx[i], token_range[i]) take proj[i] instead
        (the embed-splice of reference model.py:433-444 as one gather pass).  The splice goes through the module's
        __call__ on purpose: forward pre-hooks (the trainer's wait for this table's in-flight AdamW update) fire for it."""
        B = ids.shape[0] if ids.dim() > 1 else 1
        S = ids.numel() // B
        out = Fm.embed_splice(self.weight, ids, proj, batch_idx, token_range, B, S, dummy=grad_dummy(self.weight))
        return out.view(*ids.shape, self.embedding_dim)


class Norm(nn.Module):
    """weight (+bias) holder for RMSNorm / LayerNorm."""

    def __init__(self, dim, eps, bias: bool, dtype=None, device=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim, dtype=dtype, device=device))
        self.bias = nn.Parameter(torch.zeros(dim, dtype=dtype, device=device)) if bias else None

    def forward(self, x2d):
        """-> (normed, x_res): use x_res (an alias of x2d) for the residual add so that its gradient is fused into the
        norm-backward kernel."""
        d = grad_dummy(self.weight)
        if self.bias is None:
            return Fm.rmsnorm(x2d, self.weight, self.eps, dummy=d)
        return Fm.layernorm(x2d, self.weight, self.bias, self.eps, dummy=d)


_HF_NO_DECAY = None


def hf_decays(name: str, owner: nn.Module) -> bool:
    """Does HF Trainer apply weight decay to this parameter?  `Trainer.get_decay_parameter_names` (transformers 5.15): every
    parameter EXCEPT those of nn.LayerNorm modules and those whose lower-cased name matches bias | layernorm | rmsnorm |
    (^|.)norm($|.) | _norm($|.).  `Norm` stands where HF has nn.LayerNorm / LlamaRMSNorm.  Note what this DOES decay:
    CLIP's 1-D `class_embedding`, the position embeddings, the patch convolution (tests/test_schedule_cpu.py)."""
    import re
    global _HF_NO_DECAY
    if _HF_NO_DECAY is None:
        _HF_NO_DECAY = [re.compile(p) for p in (r"bias", r"layernorm", r"rmsnorm", r"(?:^|\.)norm(?:$|\.)", r"_norm(?:$|\.)")]
    if isinstance(owner, Norm):
        return False
    low = name.lower()
    return not any(p.search(low) for p in _HF_NO_DECAY)


_dummies: Dict[str, torch.Tensor] = {}


def grad_dummy(param: torch.Tensor) -> Optional[torch.Tensor]:
    """A requires-grad scalar that keeps a custom Function in the autograd graph when only its (non-input)
    parameters need gradients.  None when nothing here is trainable, so frozen towers record no graph."""
    if not (torch.is_grad_enabled() and param.requires_grad):
        return None
    key = str(param.device)
    d = _dummies.get(key)
    if d is None:
        d = torch.zeros(1, device=param.device, requires_grad=True)
        _dummies[key] = d
    return d


# ------------------------------------------------------------------------------------------------------ packing
class FlatSegment:
    __slots__ = ("name", "start", "end", "decay", "component", "param")

    def __init__(self, name, start, end, decay, component, param):
        self.name, self.start, self.end, self.decay, self.component, self.param = name, start, end, decay, component, param


class FlatParams:
    """Flat parameter + gradient storage for one model (single dtype)."""

    def __init__(self, named_params, device, dtype):
        """named_params: (name, parameter, component[, weight_decay]) tuples.  Without the 4th element a parameter is decayed
        iff it has >= 2 dimensions; MultiModalModelForCausalLM passes HF Trainer's rule (`hf_decays`)."""
        # order: per component, the weight-decayed parameters first, then the others; fused groups (q/k/v, gate/up weights,
        # their biases) stay adjacent because they are adjacent in module order and share their decay flag.
        items = [(t[0], t[1], t[2], (t[3] if len(t) > 3 else t[1].dim() >= 2)) for t in named_params]
        ordered = []
        comps = []
        for _, _, c, _ in items:
            if c not in comps:
                comps.append(c)
        for c in comps:
            ordered += [(n, p, c, True) for n, p, cc, d in items if cc == c and d]
            ordered += [(n, p, c, False) for n, p, cc, d in items if cc == c and not d]
        off = 0
        self.segments: List[FlatSegment] = []
        for n, p, c, decay in ordered:
            size = p.numel()
            self.segments.append(FlatSegment(n, off, off + size, decay, c, p))
            off += (size + 7) // 8 * 8     # 16-byte alignment of every tensor
        self.numel = off
        self.dtype, self.device = dtype, device
        self.data = torch.zeros(off, dtype=dtype, device=device)
        self.grad: Optional[torch.Tensor] = None
        for seg in self.segments:
            p = seg.param
            view = self.data[seg.start:seg.end].view(p.shape)
            view.copy_(p.data.to(device=device, dtype=dtype))
            p.data = view
            p._mm_flat = self
            p._mm_grad_view = None

    def ensure_grad(self):
        if self.grad is None:
            self.grad = torch.zeros(self.numel, dtype=self.dtype, device=self.device)
            for seg in self.segments:
                seg.param._mm_grad_view = self.grad[seg.start:seg.end].view(seg.param.shape)
        return self.grad

    def attach_grads(self, fresh=True):
        """Point every trainable param's .grad at its slice of the flat buffer; `fresh` = next write overwrites."""
        self.ensure_grad()
        for seg in self.segments:
            p = seg.param
            if p.requires_grad:
                p.grad = p._mm_grad_view
                p._mm_fresh = fresh
            else:
                p.grad = None

    def trainable_ranges(self) -> List[Tuple[int, int, bool]]:
        """Maximal contiguous [start, end) ranges of trainable params with equal decay flag."""
        out: List[List] = []
        for seg in self.segments:
            if not seg.param.requires_grad:
                continue
            end = (seg.end + 7) // 8 * 8
            if out and out[-1][1] == seg.start and out[-1][2] == seg.decay:
                out[-1][1] = end
            else:
                out.append([seg.start, end, seg.decay])
        return [(a, min(b, self.numel), d) for a, b, d in out]

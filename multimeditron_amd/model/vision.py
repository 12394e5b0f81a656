"""CLIP vision tower on libmmhip kernels: the MI355X replacement for the HF `CLIPModel.vision_model` the reference
calls at image_modality.py:133.  Semantics follow HF transformers 5.15.0 models/clip/modeling_clip.py: embeddings
:138-218 (conv k=s=patch without bias == patchify + GEMM, CLS token, learned positions), pre_layrnorm :608,
24x {LayerNorm, biased q/k/v/out attention with scale d^-1/2 (non-causal), LayerNorm, fc1 -> quick_gelu -> fc2} :280-384,
and `last_hidden_state` is returned WITHOUT post_layernorm (:640-657).  Parameter names equal HF's.  Only the vision
tower is instantiated (the reference keeps the unused text tower in memory)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict

import torch
import torch.nn as nn

from .. import functional as Fm
from .._lib import EPI_GELU_ERF, EPI_GELU_TANH, EPI_QUICK_GELU

_ACT = {"quick_gelu": EPI_QUICK_GELU, "gelu": EPI_GELU_ERF, "gelu_pytorch_tanh": EPI_GELU_TANH}
from ..nn import Linear, Norm, grad_dummy


@dataclass
class VisionConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    num_channels: int = 3
    kind: str = "clip"          # "clip": CLS token, bias-free conv, pre_layrnorm, tokens returned before post-LN
                                # "siglip": no CLS, conv bias, no pre-LN, post_layernorm on the returned tokens

    @classmethod
    def from_dict(cls, d: Dict[str, Any]):
        d = d.get("vision_config", d)
        return cls(**{k: v for k, v in d.items() if k in cls.__dataclass_fields__})

    def to_dict(self):
        return {k: getattr(self, k) for k in self.__dataclass_fields__}

    @property
    def num_patches(self):
        return (self.image_size // self.patch_size) ** 2


class _PatchConv(nn.Module):
    _mm_param_holder = True      # never __call__ed: VisionEmbeddings.forward reads .weight/.bias (hooks belong on the parent)

    def __init__(self, cfg, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cfg.hidden_size, cfg.num_channels, cfg.patch_size, cfg.patch_size, dtype=dtype, device=device))
        if cfg.kind == "siglip":
            self.bias = nn.Parameter(torch.empty(cfg.hidden_size, dtype=dtype, device=device))
        else:
            self.bias = None


class _PosEmb(nn.Module):
    _mm_param_holder = True

    def __init__(self, n, d, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, d, dtype=dtype, device=device))


class VisionEmbeddings(nn.Module):
    def __init__(self, cfg: VisionConfig, dtype, device):
        super().__init__()
        self.cfg = cfg
        has_cls = cfg.kind != "siglip"
        self.class_embedding = nn.Parameter(torch.empty(cfg.hidden_size, dtype=dtype, device=device)) if has_cls else None
        self.patch_embedding = _PatchConv(cfg, dtype, device)
        self.position_embedding = _PosEmb(cfg.num_patches + (1 if has_cls else 0), cfg.hidden_size, dtype, device)

    def forward(self, pixels):
        c = self.cfg
        if pixels.shape[-2] != c.image_size or pixels.shape[-1] != c.image_size:
            raise ValueError(f"Input image size ({pixels.shape[-2]}*{pixels.shape[-1]}) doesn't match model "
                             f"({c.image_size}*{c.image_size}).")
        return Fm.patch_embed(pixels.float(), self.patch_embedding.weight, self.class_embedding,
                              self.position_embedding.weight, c.patch_size, dummy=grad_dummy(self.patch_embedding.weight),
                              b_conv=self.patch_embedding.bias)


class VisionAttention(nn.Module):
    def __init__(self, cfg, dtype, device):
        super().__init__()
        D = cfg.hidden_size
        self.heads = cfg.num_attention_heads
        self.hd = D // self.heads
        self.q_proj = Linear(D, D, dtype=dtype, device=device)
        self.k_proj = Linear(D, D, dtype=dtype, device=device)
        self.v_proj = Linear(D, D, dtype=dtype, device=device)
        self.out_proj = Linear(D, D, dtype=dtype, device=device)
        self._wqkv = Fm.ParamGroup([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight])
        self._bqkv = Fm.ParamGroup([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias])


class VisionMLP(nn.Module):
    def __init__(self, cfg, dtype, device):
        super().__init__()
        self.fc1 = Linear(cfg.hidden_size, cfg.intermediate_size, dtype=dtype, device=device)
        self.fc2 = Linear(cfg.intermediate_size, cfg.hidden_size, dtype=dtype, device=device)
        if cfg.hidden_act not in _ACT:
            raise ValueError(f"unsupported vision hidden_act {cfg.hidden_act!r} (have {sorted(_ACT)})")
        self.act = _ACT[cfg.hidden_act]


class VisionLayer(nn.Module):
    def __init__(self, cfg, dtype, device):
        super().__init__()
        self.self_attn = VisionAttention(cfg, dtype, device)
        self.layer_norm1 = Norm(cfg.hidden_size, cfg.layer_norm_eps, bias=True, dtype=dtype, device=device)
        self.mlp = VisionMLP(cfg, dtype, device)
        self.layer_norm2 = Norm(cfg.hidden_size, cfg.layer_norm_eps, bias=True, dtype=dtype, device=device)

    def forward(self, x, n, T):
        a = self.self_attn
        h, x = self.layer_norm1(x)
        qkv = Fm.linear(h, a._wqkv, a._bqkv, dummy=grad_dummy(a.q_proj.weight))
        hw = Fm.attention_head_width(a.hd, qkv.dtype)
        if hw != a.hd:      # e.g. SigLIP-so400m: 16 heads x 72 run as 16 x 128 with zero columns (exact)
            qkv = Fm.head_pad(qkv, 3 * a.heads, a.hd, hw)
        o = Fm.rope_attention(qkv, None, None, None, n, T, a.heads, a.heads, hw, False, a.hd ** -0.5)
        if hw != a.hd:
            o = Fm.head_strip(o, a.heads, a.hd, hw)
        x = a.out_proj(o, residual=x)
        h, x = self.layer_norm2(x)
        h = self.mlp.fc1(h, act=self.mlp.act)
        return self.mlp.fc2(h, residual=x)


class VisionEncoder(nn.Module):
    def __init__(self, cfg, dtype, device):
        super().__init__()
        self.layers = nn.ModuleList([VisionLayer(cfg, dtype, device) for _ in range(cfg.num_hidden_layers)])


@dataclass
class VisionOutput:
    last_hidden_state: torch.Tensor


class VisionTransformer(nn.Module):
    def __init__(self, cfg: VisionConfig, dtype, device):
        super().__init__()
        self.config = cfg
        self.embeddings = VisionEmbeddings(cfg, dtype, device)
        siglip = cfg.kind == "siglip"
        self.pre_layrnorm = None if siglip else Norm(cfg.hidden_size, cfg.layer_norm_eps, bias=True, dtype=dtype, device=device)
        self.encoder = VisionEncoder(cfg, dtype, device)
        self.post_layernorm = Norm(cfg.hidden_size, cfg.layer_norm_eps, bias=True, dtype=dtype, device=device) if siglip else None

    def forward(self, pixel_values, stages=None) -> VisionOutput:
        n = pixel_values.shape[0]
        T = self.config.num_patches + (0 if self.config.kind == "siglip" else 1)
        x = self.embeddings(pixel_values)
        if stages is not None:
            stages["vit_embeddings"] = x.view(n, T, -1)
        if self.pre_layrnorm is not None:
            x, _ = self.pre_layrnorm(x)
            if stages is not None:
                stages["vit_pre_ln"] = x.view(n, T, -1)
        for i, layer in enumerate(self.encoder.layers):
            x = layer(x, n, T)
            if stages is not None and i == 0:
                stages["vit_layer0"] = x.view(n, T, -1)
        if self.post_layernorm is not None:
            x, _ = self.post_layernorm(x)
        return VisionOutput(last_hidden_state=x.view(n, T, -1))


class CLIPFeatureExtractor(nn.Module):
    """Stands where the reference keeps `AutoModel.from_pretrained(clip_name)` (image_modality.py:124):
    exposes `.vision_model`, `.vision_embed_dim`, `.device`."""

    def __init__(self, cfg: VisionConfig, dtype=torch.bfloat16, device=None):
        super().__init__()
        self.config = cfg
        self.vision_embed_dim = cfg.hidden_size
        self.vision_model = VisionTransformer(cfg, dtype, device)

    @property
    def device(self):
        return self.vision_model.embeddings.patch_embedding.weight.device


class SiglipFeatureExtractor(VisionTransformer):
    """Stands where a SigLIP plug-in keeps `SiglipVisionModel.from_pretrained(...)`: transformers 5.x holds
    embeddings / encoder / post_layernorm directly on that model (no `.vision_model` level), and so do the parameter
    names here.  The attention-pooling head of the HF model is not on the token path and is not instantiated."""

    def __init__(self, cfg: VisionConfig, dtype=torch.bfloat16, device=None):
        assert cfg.kind == "siglip"
        super().__init__(cfg, dtype, device)
        self.vision_embed_dim = cfg.hidden_size

    @property
    def device(self):
        return self.embeddings.patch_embedding.weight.device

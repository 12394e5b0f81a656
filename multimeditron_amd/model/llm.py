"""Llama / Qwen2 decoder on libmmhip kernels: the MI355X replacement for the HF `AutoModelForCausalLM` the
reference constructs at model.py:253-260 and calls at model.py:517-526 (forward) and :595-602 (decode).

Semantics follow HF transformers 5.15.0 (models/llama/modeling_llama.py): RMSNorm :53-70, RoPE :113-160 (+ llama3
inv_freq scaling modeling_rope_utils.py:641-662), GQA softmax attention :191-281, SwiGLU MLP :163-176, pre-norm
residual layer :284-325, final norm + lm_head :413,480, shifted CE loss/loss_utils.py:36-71.  Qwen2 = same graph with
biased q/k/v projections.  Parameter names equal HF's so reference checkpoints interchange.

Data layout: activations are [B*S, features] row-major in HBM; q/k/v come out of ONE fused GEMM as a
[B*S, (Hq+2Hkv)*D] buffer that RoPE rewrites in place and the attention kernel reads through strides; gate/up come
out of one fused GEMM as [B*S, 2I]; the residual add is the GEMM epilogue of o_proj/down_proj."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from .. import functional as Fm
from .. import kernels as K
from ..nn import Embedding, Linear, Norm, grad_dummy


@dataclass
class LLMConfig:
    model_type: str = "llama"
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: Optional[int] = None
    vocab_size: int = 128256
    rms_norm_eps: float = 1e-5
    tie_word_embeddings: bool = False
    attention_bias: bool = False
    max_position_embeddings: int = 131072
    rope_parameters: Dict[str, Any] = field(default_factory=lambda: {"rope_type": "default", "rope_theta": 10000.0})

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "LLMConfig":
        d = dict(d)
        rp = d.get("rope_parameters")
        if not rp:   # older HF layout: rope_theta + rope_scaling
            rp = dict(d.get("rope_scaling") or {})
            rp.setdefault("rope_type", rp.pop("type", "default"))
            rp.setdefault("rope_theta", d.get("rope_theta", 10000.0))
        mt = d.get("model_type", "llama")
        known = {f for f in cls.__dataclass_fields__}
        kw = {k: v for k, v in d.items() if k in known}
        kw["rope_parameters"] = rp
        if mt.startswith("qwen2"):
            kw["attention_bias"] = True          # Qwen2 hard-codes biased q/k/v (HF:models/qwen2)
        if not kw.get("head_dim"):
            kw["head_dim"] = kw.get("hidden_size", 4096) // kw.get("num_attention_heads", 32)
        return cls(**kw)

    def to_dict(self):
        return {k: getattr(self, k) for k in self.__dataclass_fields__}


def rope_inv_freq(cfg: LLMConfig) -> torch.Tensor:
    rp = cfg.rope_parameters
    base = float(rp.get("rope_theta", 10000.0))
    hd = cfg.head_dim
    inv = 1.0 / (base ** (torch.arange(0, hd, 2, dtype=torch.int64).to(torch.float32) / hd))
    if rp.get("rope_type", "default") == "llama3":
        factor, lo, hi = rp["factor"], rp["low_freq_factor"], rp["high_freq_factor"]
        old = rp["original_max_position_embeddings"]
        wavelen = 2 * math.pi / inv
        scaled = torch.where(wavelen > old / lo, inv / factor, inv)
        smooth = (old / wavelen - lo) / (hi - lo)
        mid = (1 - smooth) * scaled / factor + smooth * scaled
        is_mid = ~(wavelen < old / hi) * ~(wavelen > old / lo)
        inv = torch.where(is_mid, mid, scaled)
    return inv


class Attention(nn.Module):
    def __init__(self, cfg: LLMConfig, dtype, device):
        super().__init__()
        H, D = cfg.hidden_size, cfg.head_dim
        self.Hq, self.Hkv, self.D = cfg.num_attention_heads, cfg.num_key_value_heads, D
        b = cfg.attention_bias
        self.q_proj = Linear(H, self.Hq * D, bias=b, dtype=dtype, device=device)
        self.k_proj = Linear(H, self.Hkv * D, bias=b, dtype=dtype, device=device)
        self.v_proj = Linear(H, self.Hkv * D, bias=b, dtype=dtype, device=device)
        self.o_proj = Linear(self.Hq * D, H, bias=False, dtype=dtype, device=device)
        self._wqkv = Fm.ParamGroup([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight])
        self._bqkv = Fm.ParamGroup([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]) if b else None


class MLP(nn.Module):
    def __init__(self, cfg: LLMConfig, dtype, device):
        super().__init__()
        H, I = cfg.hidden_size, cfg.intermediate_size
        self.I = I
        self.gate_proj = Linear(H, I, bias=False, dtype=dtype, device=device)
        self.up_proj = Linear(H, I, bias=False, dtype=dtype, device=device)
        self.down_proj = Linear(I, H, bias=False, dtype=dtype, device=device)
        self._wgu = Fm.ParamGroup([self.gate_proj.weight, self.up_proj.weight])


class DecoderLayer(nn.Module):
    def __init__(self, cfg: LLMConfig, dtype, device):
        super().__init__()
        self.self_attn = Attention(cfg, dtype, device)
        self.mlp = MLP(cfg, dtype, device)
        self.input_layernorm = Norm(cfg.hidden_size, cfg.rms_norm_eps, bias=False, dtype=dtype, device=device)
        self.post_attention_layernorm = Norm(cfg.hidden_size, cfg.rms_norm_eps, bias=False, dtype=dtype, device=device)

    @torch.no_grad()
    def decode_step(self, x, cos, sin, key_mask, B, cache):
        """One new token per sequence (generate's decode loop, reference model.py:595-602): the four linears take the
        weight-streaming GEMM (M <= 16), RoPE and the cache append are one pass, attention is the split-K stream over the
        cache.  Same arithmetic and rounding points as forward().  (Folding RMSNorm / SwiGLU into the GEMM's x operand was
        measured SLOWER: every workgroup redoes the transform on all 64 lanes and the kernel turns VALU-bound, 43 vs 27 us
        and 57 vs 25 us per launch, so they stay separate launches.)"""
        a, m = self.self_attn, self.mlp
        Hq, Hkv, D = a.Hq, a.Hkv, a.D
        h, _ = K.rmsnorm_fwd(x, self.input_layernorm.weight, self.input_layernorm.eps)
        qkv = K.linear_fwd(h, a._wqkv.tensor(), bias=a._bqkv.tensor() if a._bqkv is not None else None)
        K.rope_append_(qkv, B, Hq, Hkv, D, cos, sin, cache.k, cache.v, cache.len)
        cache.len += 1
        o = K.attn_decode(qkv[:, : Hq * D].view(B, Hq, D), cache.k[:, : cache.len], cache.v[:, : cache.len], key_mask, D ** -0.5)
        x = K.linear_fwd(o.view(B, Hq * D), a.o_proj.weight, residual=x)
        h, _ = K.rmsnorm_fwd(x, self.post_attention_layernorm.weight, self.post_attention_layernorm.eps)
        act = K.swiglu_fwd(K.linear_fwd(h, m._wgu.tensor()), m.I)
        return K.linear_fwd(act, m.down_proj.weight, residual=x)

    @torch.no_grad()
    def decode_step_fused(self, x, cos, sin, key_mask, B, cache):
        """The same decode step with the tiny launches folded into the weight-streaming GEMMs (round 3; csrc/mm_gemm.hip
        gemv_stream_kernel): input RMSNorm + q|k|v + RoPE + cache append, o_proj + residual, post-attention RMSNorm + gate|up +
        SwiGLU, down_proj + residual -- 6 launches with the two attention kernels, none of them a norm.  x = residual stream in
        and out.  Same bits as decode_step."""
        a, m = self.self_attn, self.mlp
        Hq, Hkv, D = a.Hq, a.Hkv, a.D
        n1, n2 = self.input_layernorm, self.post_attention_layernorm
        qkv = K.decode_qkv_rope_append(x, a._wqkv.tensor(), a._bqkv.tensor() if a._bqkv is not None else None, Hq, Hkv, D, cos, sin,
                                       cache.k, cache.v, cache.len, norm_w=n1.weight, eps=n1.eps)
        cache.len += 1
        o = K.attn_decode(qkv[:, : Hq * D].view(B, Hq, D), cache.k[:, : cache.len], cache.v[:, : cache.len], key_mask, D ** -0.5)
        x = K.decode_linear(o.view(B, Hq * D), a.o_proj.weight, residual=x)
        act = K.decode_gateup_swiglu(x, m._wgu.tensor(), m.I, norm_w=n2.weight, eps=n2.eps)
        return K.decode_linear(act, m.down_proj.weight, residual=x)

    def can_decode_step(self, x, B, S, cache):
        a = self.self_attn
        return (cache is not None and S == 1 and B <= 16 and x.dtype == torch.bfloat16 and not torch.is_grad_enabled()
                and K.attn_decode_supported(x.dtype, a.Hq, a.Hkv, a.D))

    def can_decode_step_fused(self, x, B, S, cache):
        a = self.self_attn
        return (self.can_decode_step(x, B, S, cache) and a.D == 128 and K.decode_fusions() and x.shape[-1] <= 8192 and self.mlp.I % 4 == 0
                and K.decode_fits(B, x.shape[-1]) and K.decode_fits(B, self.mlp.I) and K.decode_fits(B, a.Hq * a.D)
                and a.o_proj.bias is None and self.mlp.down_proj.bias is None and self.input_layernorm.bias is None)

    def forward(self, x, cos, sin, key_mask, B, S, cache=None, rows=None):
        """rows (functional.LossRows, LAST layer of a training forward only): nothing after this layer's attention reads the
        rows that carry no label -- their hidden states feed no later layer and no loss term -- so o_proj, the post-attention norm
        and the MLP run on the labelled rows alone and the layer returns [rows.n, H].  (Keys and values come from every row, so
        attention itself is computed in full.)  Same loss, same gradients."""
        if self.can_decode_step(x, B, S, cache):
            return self.decode_step(x, cos, sin, key_mask, B, cache)
        a = self.self_attn
        h, x = self.input_layernorm(x)
        if cache is None:      # projection (+ RoPE in its epilogue) + attention as one autograd node
            o = Fm.qkv_rope_attention(h, a._wqkv, a._bqkv, cos, sin, key_mask, B, S, a.Hq, a.Hkv, a.D, True, a.D ** -0.5,
                                      dummy=grad_dummy(a.q_proj.weight))
        else:
            qkv = Fm.linear(h, a._wqkv, a._bqkv, dummy=grad_dummy(a.q_proj.weight))
            o = cache.attend(qkv, cos, sin, key_mask, B, S, a)
        if rows is not None:
            o, x = Fm.rows_select(o, rows), Fm.rows_select(x, rows)
        x = a.o_proj(o, residual=x)
        h, x = self.post_attention_layernorm(x)
        return Fm.swiglu_mlp(h, self.mlp._wgu, self.mlp.down_proj.weight, self.mlp.I, residual=x,
                             dummy=grad_dummy(self.mlp.gate_proj.weight))


class LayerKVCache:
    """Contiguous per-layer KV cache [B, Smax, Hkv, D] for the decode loop of generate (model.py:581-638)."""

    def __init__(self, B, Smax, Hkv, D, dtype, device):
        self.k = torch.zeros((B, Smax, Hkv, D), dtype=dtype, device=device)
        self.v = torch.zeros((B, Smax, Hkv, D), dtype=dtype, device=device)
        self.len = 0

    @torch.no_grad()
    def attend(self, qkv, cos, sin, key_mask, B, S, a: Attention):
        Hq, Hkv, D = a.Hq, a.Hkv, a.D
        W = (Hq + 2 * Hkv) * D
        K.rope_apply_(qkv, B * S, Hq + Hkv, D, W, cos, sin)
        q = qkv[:, : Hq * D].view(B, S, Hq, D)
        kn = qkv[:, Hq * D:(Hq + Hkv) * D].view(B, S, Hkv, D)
        vn = qkv[:, (Hq + Hkv) * D:].view(B, S, Hkv, D)
        self.k[:, self.len:self.len + S].copy_(kn)      # device-side memory plumbing
        self.v[:, self.len:self.len + S].copy_(vn)
        self.len += S
        if S == 1 and K.attn_decode_supported(qkv.dtype, Hq, Hkv, D):      # decode step: stream the cache once (split-K)
            out = K.attn_decode(q.view(B, Hq, D), self.k[:, : self.len], self.v[:, : self.len], key_mask, D ** -0.5)
            return out.view(B, Hq * D)
        out, _ = K.attn_fwd(q, self.k[:, : self.len], self.v[:, : self.len], key_mask, True, D ** -0.5)
        return out.view(B * S, Hq * D)


class DecoderModel(nn.Module):
    def __init__(self, cfg: LLMConfig, dtype, device):
        super().__init__()
        self.embed_tokens = Embedding(cfg.vocab_size, cfg.hidden_size, dtype=dtype, device=device)
        self.layers = nn.ModuleList([DecoderLayer(cfg, dtype, device) for _ in range(cfg.num_hidden_layers)])
        self.norm = Norm(cfg.hidden_size, cfg.rms_norm_eps, bias=False, dtype=dtype, device=device)


@dataclass
class CausalLMOutput:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[Any] = None
    hidden_states: Optional[Any] = None
    attentions: Optional[Any] = None

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else (self.loss, self.logits, self.past_key_values)[k]


class CausalLM(nn.Module):
    """`self.model` of MultiModalModelForCausalLM (reference attribute name, model.py:253-262)."""

    def __init__(self, cfg: LLMConfig, dtype=torch.bfloat16, device=None):
        super().__init__()
        self.config = cfg
        self.model = DecoderModel(cfg, dtype, device)
        self.lm_head = Linear(cfg.hidden_size, cfg.vocab_size, bias=False, dtype=dtype, device=device)
        if cfg.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight
        self._inv_freq = None

    # -- reference surface -------------------------------------------------------------------------
    @property
    def device(self):
        return self.model.embed_tokens.weight.device

    @property
    def dtype(self):
        return self.model.embed_tokens.weight.dtype

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def set_input_embeddings(self, value):
        self.model.embed_tokens = value

    def resize_token_embeddings(self, new_num_tokens: int, mean_resizing: bool = False, std: float = 0.02):
        """model.py:262.  New rows ~ N(0, std) (mean_resizing=False semantics); tied heads stay tied."""
        emb = self.model.embed_tokens
        old = emb.num_embeddings
        if new_num_tokens == old:
            return emb

        def grow(w):
            nw = torch.empty((new_num_tokens, w.shape[1]), dtype=w.dtype, device=w.device)
            n = min(old, new_num_tokens)
            nw[:n] = w.data[:n]
            if new_num_tokens > old:
                nw[old:].normal_(mean=0.0, std=std)
            return nn.Parameter(nw, requires_grad=w.requires_grad)

        tied = self.lm_head.weight is emb.weight
        emb.weight = grow(emb.weight)
        emb.num_embeddings = new_num_tokens
        self.lm_head.weight = emb.weight if tied else grow(self.lm_head.weight)
        self.lm_head.out_features = new_num_tokens
        self.config.vocab_size = new_num_tokens
        return emb

    # -- compute --------------------------------------------------------------------------------------
    def _tables(self, position_ids):
        if self._inv_freq is None or self._inv_freq.device != position_ids.device:
            self._inv_freq = rope_inv_freq(self.config).to(position_ids.device)
        return K.rope_table(position_ids.reshape(-1).contiguous(), self._inv_freq, self.dtype == torch.bfloat16)

    def new_cache(self, B, Smax):
        c = self.config
        return [LayerKVCache(B, Smax, c.num_key_value_heads, c.head_dim, self.dtype, self.device)
                for _ in range(c.num_hidden_layers)]

    def forward(self, input_ids=None, inputs_embeds=None, attention_mask=None, position_ids=None,
                past_key_values=None, labels=None, use_cache=None, return_dict=True, logits_to_keep: int = 0, **kwargs):
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if inputs_embeds is None:
            inputs_embeds = self.model.embed_tokens(input_ids)
        B, S, H = inputs_embeds.shape
        dev = inputs_embeds.device
        cache = past_key_values
        past = cache[0].len if cache else 0
        rows = kwargs.get("loss_rows")
        if rows is not None and (labels is None or cache or use_cache or logits_to_keep or rows.n == 0 or rows.total != B * S):
            rows = None                                      # anything but a plain training forward: every position's logits
        if use_cache and cache is None:
            cache = self.new_cache(B, S + int(kwargs.get("max_new_tokens", 512)))
        if position_ids is None:
            position_ids = (torch.arange(S, device=dev) + past).unsqueeze(0).expand(B, S)
        cos, sin = self._tables(position_ids.to(dev))
        key_mask = None
        if attention_mask is not None and past == 0 and (
                getattr(attention_mask, "_mm_all_ones", False) or (not attention_mask.is_cuda and bool(attention_mask.all()))):
            attention_mask = None       # an all-ones mask masks nothing (HF: _ignore_causal_mask_sdpa).  Known without a device
                                        # sync only for a host tensor, or from the flag DevicePrefetcher computed on the host copy
        if attention_mask is not None:
            key_mask = attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        x = inputs_embeds.reshape(B * S, H)
        if not x.is_contiguous():
            x = x.contiguous()
        layers = self.model.layers
        fused_head = False
        if cache and len(layers) and all(layer.can_decode_step_fused(x, B, S, cache[i]) for i, layer in enumerate(layers)):
            # decode step: every RMSNorm is applied by the projection that consumes it (DecoderLayer.decode_step_fused), the final
            # norm by lm_head below
            for i, layer in enumerate(layers):
                x = layer.decode_step_fused(x, cos, sin, key_mask, B, cache[i])
            fused_head = self.lm_head.bias is None and self.model.norm.bias is None       # S == 1: logits_to_keep keeps the one row
            if not fused_head:
                x, _ = self.model.norm(x)
        else:
            # training step (Trainer.compute_loss hands over `loss_rows`): the loss and every gradient depend only on the rows whose
            # shifted label is not -100 (HF:loss/loss_utils.py:36-71 ignores the others; their dlogits are exactly zero), so the
            # final norm, lm_head and the loss -- and what follows attention in the LAST layer -- run on those rows alone;
            # `logits` is then not returned
            last = len(layers) - 1
            for i, layer in enumerate(layers):
                x = layer(x, cos, sin, key_mask, B, S, cache=cache[i] if cache else None, rows=rows if i == last else None)
            if rows is not None and not len(layers):
                x = Fm.rows_select(x, rows)
            x, _ = self.model.norm(x)
        V = self.config.vocab_size
        if logits_to_keep:
            x = x.view(B, S, H)[:, -logits_to_keep:, :].reshape(-1, H)
            Sk = logits_to_keep
        else:
            Sk = S
        if fused_head:
            logits2d = K.decode_linear(x, self.lm_head.weight, norm_w=self.model.norm.weight, eps=self.model.norm.eps, ldc_pad=True)
        else:
            logits2d = self.lm_head(x, ldc_pad=True)             # [B*Sk, V] view, row stride padded to 64
        loss = None
        if rows is not None:
            loss = Fm.causal_lm_loss(logits2d, V, rows.labels)
            out = CausalLMOutput(loss=loss, logits=None, past_key_values=None)
            return out if return_dict else (loss, None, None)
        if labels is not None:
            shift = torch.nn.functional.pad(labels.to(dev), (0, 1), value=-100)[..., 1:].reshape(-1).contiguous()
            loss = Fm.causal_lm_loss(logits2d, V, shift)
        logits = logits2d.as_strided((B, Sk, V), (Sk * logits2d.stride(0), logits2d.stride(0), 1))
        out = CausalLMOutput(loss=loss, logits=logits, past_key_values=cache if use_cache else None)
        return out if return_dict else (loss, logits, out.past_key_values)

    __call__ = nn.Module.__call__

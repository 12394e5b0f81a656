"""CPU ORACLE — test infrastructure, NOT product code.

Plain-PyTorch (CPU) restatement of the reference's hot path

    modality encoder -> projector -> embed-splice -> LLM decoder (+loss, +greedy generate)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (multimeditron_amd) never does; it fails loudly when the HIP library is
missing.

Parity pin: this file is checked against golden vectors produced by running the real
reference (leagrieder/MultiMeditron @ 2026-01-30 + transformers 5.15.0) on CPU in the build
container (tools/make_golden.py -> tests/golden/*.safetensors; tests/test_oracle_golden.py).

What each function restates (reference file:line, or the third-party HF module the reference
calls at that line -- `HF:` = transformers 5.15.0):

  clip_vision_tower   image_modality.py:133 -> HF:models/clip/modeling_clip.py:138-218 (embeddings),
                      :280-384 (attention/MLP/layer), :594-657 (pre_layrnorm, encoder; last_hidden_state
                      is returned WITHOUT post_layernorm)
  siglip_vision_tower BASELINE config 5 (no reference modality: SURVEY section 0 fact 9) -> HF:models/siglip/modeling_siglip.py
                      SiglipVisionEmbeddings (conv WITH bias + learned positions, no CLS), SiglipEncoderLayer
                      (pre-LN, biased q/k/v/out, fc1 -> gelu_pytorch_tanh -> fc2), post_layernorm on the returned
                      tokens; the pooling head is not on the token path
  mlp_projector       projectors/mlp.py:33-39 (Linear-GELU(erf)-Linear-GELU(erf)-Linear, all biased)
  image_modality      image_modality.py:130-137 (stack -> vision tower -> drop CLS -> projector)
  embed_splice        model.py:433-444 (embedding lookup, then index_put of projected patches)
  rope_inv_freq       HF:modeling_rope_utils.py:641-662 (llama3) / HF:models/llama/modeling_llama.py:98-104
  decoder_forward     model.py:517-526 -> HF:models/llama/modeling_llama.py:53-70,130-160,163-176,191-213,
                      217-325,367-417,480 (Qwen2 = same with q/k/v bias)
  causal_lm_loss      HF:loss/loss_utils.py:36-71 (shift, CE ignore_index=-100, mean over kept tokens)
  multimodal_forward  model.py:449-526
  greedy_generate     model.py:528-640 (argmax(softmax(logits/T)); decode position = padded length + i - 1)
  cross_attention     model/attention.py:48-101 (eval mode: dropout off)
  moe_image_modality  modalities/image_modality_moe.py:152-210 downstream of the gating network (gate weights are an input)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

VIS_PREFIX = "modalities_with_projection.0.feature_extractor.vision_model."
SIGLIP_PREFIX = "modalities_with_projection.0.feature_extractor."      # transformers 5.x SiglipVisionModel layout
PROJ_PREFIX = "modalities_with_projection.0.projector.projection."
LLM_PREFIX = "model.model."


# ------------------------------------------------------------------------------------------
# vision tower
# ------------------------------------------------------------------------------------------
def _layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def _mha_noncausal(x, w, pre, heads):
    n, T, D = x.shape
    hd = D // heads
    q = F.linear(x, w[pre + "q_proj.weight"], w[pre + "q_proj.bias"]).view(n, T, heads, hd).transpose(1, 2)
    k = F.linear(x, w[pre + "k_proj.weight"], w[pre + "k_proj.bias"]).view(n, T, heads, hd).transpose(1, 2)
    v = F.linear(x, w[pre + "v_proj.weight"], w[pre + "v_proj.bias"]).view(n, T, heads, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
    p = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(p, v).transpose(1, 2).reshape(n, T, D)
    return F.linear(o, w[pre + "out_proj.weight"], w[pre + "out_proj.bias"])


def clip_vision_tower(w: Dict[str, torch.Tensor], pixels: torch.Tensor, vis: dict, stages: Optional[dict] = None,
                      prefix: Optional[str] = None):
    """pixels [n,3,H,W] -> last_hidden_state [n,1+P,Dv] (pre post-LN, CLS still present)."""
    pre = VIS_PREFIX if prefix is None else prefix
    pw = w[pre + "embeddings.patch_embedding.weight"]
    ps = vis["patch_size"]
    patches = F.conv2d(pixels.to(pw.dtype), pw, None, stride=ps)          # [n,Dv,g,g]
    patches = patches.flatten(2).transpose(1, 2)                           # [n,P,Dv]
    cls = w[pre + "embeddings.class_embedding"].expand(pixels.shape[0], 1, -1)
    x = torch.cat([cls, patches], dim=1) + w[pre + "embeddings.position_embedding.weight"].unsqueeze(0)
    if stages is not None:
        stages["vit_embeddings"] = x
    eps = vis.get("layer_norm_eps", 1e-5)
    x = _layer_norm(x, w[pre + "pre_layrnorm.weight"], w[pre + "pre_layrnorm.bias"], eps)
    if stages is not None:
        stages["vit_pre_ln"] = x
    for i in range(vis["num_hidden_layers"]):
        lp = f"{pre}encoder.layers.{i}."
        h = _layer_norm(x, w[lp + "layer_norm1.weight"], w[lp + "layer_norm1.bias"], eps)
        x = x + _mha_noncausal(h, w, lp + "self_attn.", vis["num_attention_heads"])
        h = _layer_norm(x, w[lp + "layer_norm2.weight"], w[lp + "layer_norm2.bias"], eps)
        h = F.linear(h, w[lp + "mlp.fc1.weight"], w[lp + "mlp.fc1.bias"])
        act = vis.get("hidden_act", "quick_gelu")
        h = _quick_gelu(h) if act == "quick_gelu" else F.gelu(h)
        x = x + F.linear(h, w[lp + "mlp.fc2.weight"], w[lp + "mlp.fc2.bias"])
        if stages is not None and i == 0:
            stages["vit_layer0"] = x
    if stages is not None:
        stages["vit_last_hidden"] = x
    return x


def siglip_vision_tower(w: Dict[str, torch.Tensor], pixels: torch.Tensor, vis: dict, stages: Optional[dict] = None):
    """pixels [n,3,H,W] -> last_hidden_state [n,P,Dv] = post_layernorm(encoder(embeddings)); no CLS token."""
    pre = SIGLIP_PREFIX
    pw = w[pre + "embeddings.patch_embedding.weight"]
    x = F.conv2d(pixels.to(pw.dtype), pw, w[pre + "embeddings.patch_embedding.bias"], stride=vis["patch_size"])
    x = x.flatten(2).transpose(1, 2) + w[pre + "embeddings.position_embedding.weight"].unsqueeze(0)
    if stages is not None:
        stages["vit_embeddings"] = x
    eps = vis.get("layer_norm_eps", 1e-6)
    for i in range(vis["num_hidden_layers"]):
        lp = f"{pre}encoder.layers.{i}."
        h = _layer_norm(x, w[lp + "layer_norm1.weight"], w[lp + "layer_norm1.bias"], eps)
        x = x + _mha_noncausal(h, w, lp + "self_attn.", vis["num_attention_heads"])
        h = _layer_norm(x, w[lp + "layer_norm2.weight"], w[lp + "layer_norm2.bias"], eps)
        h = F.gelu(F.linear(h, w[lp + "mlp.fc1.weight"], w[lp + "mlp.fc1.bias"]), approximate="tanh")
        x = x + F.linear(h, w[lp + "mlp.fc2.weight"], w[lp + "mlp.fc2.bias"])
        if stages is not None and i == 0:
            stages["vit_layer0"] = x
    x = _layer_norm(x, w[pre + "post_layernorm.weight"], w[pre + "post_layernorm.bias"], eps)
    if stages is not None:
        stages["vit_last_hidden"] = x
    return x


def mlp_projector(w, x, prefix: Optional[str] = None):
    p = PROJ_PREFIX if prefix is None else prefix
    x = F.gelu(F.linear(x, w[p + "0.weight"], w[p + "0.bias"]))
    x = F.gelu(F.linear(x, w[p + "2.weight"], w[p + "2.bias"]))
    return F.linear(x, w[p + "4.weight"], w[p + "4.bias"])


def image_modality(w, pixels, vis, stages=None):
    if vis.get("kind") == "siglip":
        feats = siglip_vision_tower(w, pixels, vis, stages)
    else:
        feats = clip_vision_tower(w, pixels, vis, stages)[:, 1:, :]
    out = mlp_projector(w, feats)
    if stages is not None:
        stages["projector_out"] = out
    return out


# ------------------------------------------------------------------------------------------
# MoE image modality: modalities/image_modality_moe.py:152-210, model/attention.py:48-101
# ------------------------------------------------------------------------------------------
def cross_attention(w, pre, x, experts_ctx, heads):
    """model/attention.py:60-101 in eval mode (attn_drop / proj_drop are identities): queries x [B,Nq,C] attend over the
    contexts concatenated along the sequence; biased q/k/v projections, softmax(q k^T / sqrt(d)) v, output projection."""
    B, Nq, C = x.shape
    ctx = torch.cat(experts_ctx, dim=1)
    hd = C // heads
    q = F.linear(x, w[pre + "q_proj.weight"], w.get(pre + "q_proj.bias")).view(B, Nq, heads, hd).transpose(1, 2)
    k = F.linear(ctx, w[pre + "k_proj.weight"], w.get(pre + "k_proj.bias")).view(B, -1, heads, hd).transpose(1, 2)
    v = F.linear(ctx, w[pre + "v_proj.weight"], w.get(pre + "v_proj.bias")).view(B, -1, heads, hd).transpose(1, 2)
    a = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) * (hd ** -0.5), dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, Nq, C)
    return F.linear(o, w[pre + "proj.weight"], w[pre + "proj.bias"])


def moe_image_modality(w, pixels, gate_weights, vis, num_experts, fusion, generalist_idx=-1, heads=8, perm=None):
    """image_modality_moe.py:152-210 downstream of the gate: every expert tower on every image (CLS dropped), one of the
    three fusions, then the MLP projector.  `gate_weights` [n, E] = the gating network's softmax output (the ResNet-50 gate
    itself is out of scope: torchvision is absent); `perm` maps gate classes to expert order (:118-135)."""
    outs = [clip_vision_tower(w, pixels, vis, prefix=f"experts.{e}.")[:, 1:, :] for e in range(num_experts)]
    st = torch.stack(outs, dim=1)                                     # [n, E, P, C]
    gw = gate_weights if perm is None else gate_weights.index_select(-1, perm)
    if fusion == "sequence_append":
        fused = torch.flatten(st, 1, 2)
    elif fusion == "weighted_average":
        fused = (st * gw.to(st.dtype)[:, :, None, None]).sum(dim=1)
    elif fusion == "cross_attn":
        E = num_experts
        gi = generalist_idx % E
        spec = [i for i in range(E) if i != gi]
        ws = torch.softmax(gw[:, spec], dim=-1).to(st.dtype)
        ctx = [st[:, e] * ws[:, j].view(-1, 1, 1) for j, e in enumerate(spec)]
        fused = cross_attention(w, "cross_attn.", st[:, gi], ctx, heads)
    else:
        raise ValueError(f"Unsupported fusion_method: {fusion}")
    return mlp_projector(w, fused, prefix="projector.projection.")


def moe_image_modality_pep(w, pixels, gate_weights, vis, num_experts, fusion, generalist_idx=-1, heads=8, perm=None):
    """image_modality_moe_pep.py:191-249 (per-expert projection): every expert tower on every image (CLS dropped), each
    through ITS OWN MLP projector (`projectors.{e}.projection.*`), then the fusion in the projected space.  As in the
    reference, `weighted_average` takes the gate's weights as they come (:214-216) and only `cross_attn` aligns them to the
    expert order with `perm` (:229-230)."""
    outs = [mlp_projector(w, clip_vision_tower(w, pixels, vis, prefix=f"experts.{e}.")[:, 1:, :], prefix=f"projectors.{e}.projection.")
            for e in range(num_experts)]
    st = torch.stack(outs, dim=1)                                     # [n, E, P, H]
    if fusion == "sequence_append":
        return torch.flatten(st, 1, 2)
    if fusion == "weighted_average":
        return (st * gate_weights.to(st.dtype)[:, :, None, None]).sum(dim=1)
    if fusion == "cross_attn":
        E = num_experts
        gw = gate_weights if perm is None else gate_weights.index_select(-1, perm)
        spec = [i for i in range(E) if i != generalist_idx]           # the reference indexes with generalist_idx as given (:220-223)
        ws = torch.softmax(gw[:, spec], dim=-1).to(st.dtype)
        ctx = [st[:, e] * ws[:, j].view(-1, 1, 1) for j, e in enumerate(spec)]
        return cross_attention(w, "cross_attn.", st[:, generalist_idx], ctx, heads)
    raise ValueError(f"Unsupported fusion_method: {fusion}")


# ------------------------------------------------------------------------------------------
# splice
# ------------------------------------------------------------------------------------------
def embed_splice(emb_weight, input_ids, projected, batch_idx, token_range):
    e = F.embedding(input_ids, emb_weight)
    if projected is not None and batch_idx is not None and batch_idx.numel() > 0:
        e = e.clone()
        e[batch_idx, token_range] = projected.reshape(-1, projected.shape[-1]).to(e.dtype)
    return e


# ------------------------------------------------------------------------------------------
# decoder
# ------------------------------------------------------------------------------------------
def rope_inv_freq(llm: dict) -> torch.Tensor:
    rp = llm.get("rope_parameters") or {}
    if not rp and llm.get("rope_scaling"):
        rp = dict(llm["rope_scaling"])
        rp.setdefault("rope_theta", llm.get("rope_theta", 10000.0))
    base = float(rp.get("rope_theta", llm.get("rope_theta", 10000.0)))
    hd = llm.get("head_dim") or llm["hidden_size"] // llm["num_attention_heads"]
    inv = 1.0 / (base ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    if rp.get("rope_type", "default") == "llama3":
        factor, lo, hi = rp["factor"], rp["low_freq_factor"], rp["high_freq_factor"]
        old = rp["original_max_position_embeddings"]
        wavelen = 2 * math.pi / inv
        inv_l = torch.where(wavelen > old / lo, inv / factor, inv)
        smooth = (old / wavelen - lo) / (hi - lo)
        smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
        medium = ~(wavelen < old / hi) * ~(wavelen > old / lo)
        inv = torch.where(medium, smoothed, inv_l)
    return inv


def _rms_norm(x, weight, eps):
    dt = x.dtype
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return weight * xf.to(dt)


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def decoder_forward(w, embeds, attention_mask, position_ids, llm: dict, cache: Optional[list] = None,
                    stages: Optional[dict] = None):
    """embeds [B,S,H] -> final-norm hidden [B,S,H].  cache: list (per layer) of [k,v] or None; updated in place."""
    pre = LLM_PREFIX
    B, S, H = embeds.shape
    nh, nkv = llm["num_attention_heads"], llm["num_key_value_heads"]
    hd = llm.get("head_dim") or H // nh
    eps = llm["rms_norm_eps"]
    inv = rope_inv_freq(llm)
    freqs = position_ids[:, :, None].float() * inv[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos().to(embeds.dtype)[:, None], emb.sin().to(embeds.dtype)[:, None]
    past = 0 if cache is None or cache[0] is None else cache[0][0].shape[2]
    T = past + S
    # additive mask [B,1,S,T]: causal AND key padding (HF create_causal_mask, eager)
    qpos = torch.arange(past, T)[:, None]
    kpos = torch.arange(T)[None, :]
    allowed = (kpos <= qpos)[None, None]
    if attention_mask is not None:
        allowed = allowed & attention_mask[:, None, None, :T].bool()
    neg = torch.finfo(embeds.dtype).min
    amask = torch.zeros(B, 1, S, T, dtype=embeds.dtype).masked_fill(~allowed, neg)
    x = embeds
    for i in range(llm["num_hidden_layers"]):
        lp = f"{pre}layers.{i}."
        h = _rms_norm(x, w[lp + "input_layernorm.weight"], eps)
        q = F.linear(h, w[lp + "self_attn.q_proj.weight"], w.get(lp + "self_attn.q_proj.bias")).view(B, S, nh, hd).transpose(1, 2)
        k = F.linear(h, w[lp + "self_attn.k_proj.weight"], w.get(lp + "self_attn.k_proj.bias")).view(B, S, nkv, hd).transpose(1, 2)
        v = F.linear(h, w[lp + "self_attn.v_proj.weight"], w.get(lp + "self_attn.v_proj.bias")).view(B, S, nkv, hd).transpose(1, 2)
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        if cache is not None:
            if cache[i] is not None:
                k = torch.cat([cache[i][0], k], dim=2)
                v = torch.cat([cache[i][1], v], dim=2)
            cache[i] = [k, v]
        rep = nh // nkv
        kk = k[:, :, None].expand(B, nkv, rep, T, hd).reshape(B, nh, T, hd)
        vv = v[:, :, None].expand(B, nkv, rep, T, hd).reshape(B, nh, T, hd)
        s = torch.matmul(q, kk.transpose(2, 3)) * (hd ** -0.5) + amask
        p = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(p, vv).transpose(1, 2).reshape(B, S, nh * hd)
        x = x + F.linear(o, w[lp + "self_attn.o_proj.weight"], w.get(lp + "self_attn.o_proj.bias"))
        h = _rms_norm(x, w[lp + "post_attention_layernorm.weight"], eps)
        g = F.linear(h, w[lp + "mlp.gate_proj.weight"])
        u = F.linear(h, w[lp + "mlp.up_proj.weight"])
        x = x + F.linear(F.silu(g) * u, w[lp + "mlp.down_proj.weight"])
        if stages is not None and i == 0:
            stages["llm_layer0"] = x
    x = _rms_norm(x, w[pre + "norm.weight"], eps)
    if stages is not None:
        stages["llm_final_norm"] = x
    return x


def lm_head_weight(w, llm=None):
    tied = bool(llm and llm.get("tie_word_embeddings", False))
    if tied or "model.lm_head.weight" not in w:
        return w[LLM_PREFIX + "embed_tokens.weight"]
    return w["model.lm_head.weight"]


def causal_lm_loss(logits, labels, ignore_index=-100):
    logits = logits.float()
    shift = F.pad(labels, (0, 1), value=ignore_index)[..., 1:].contiguous()
    return F.cross_entropy(logits.view(-1, logits.shape[-1]), shift.view(-1), ignore_index=ignore_index)


def multimodal_embed(w, batch, meta, stages=None):
    pmi = batch.get("processed_multimodal_inputs") or {}
    proj = bi = tr = None
    if pmi.get("stacked", {}).get("image"):
        pixels = torch.stack(list(pmi["stacked"]["image"]), dim=0)
        proj = image_modality(w, pixels, meta["vision"], stages)
        bi, tr = pmi["batch_idx"]["image"], pmi["token_range"]["image"]
    e = embed_splice(w[LLM_PREFIX + "embed_tokens.weight"], batch["input_ids"], proj, bi, tr)
    if stages is not None:
        stages["spliced_embeds"] = e
    return e


def multimodal_forward(w, batch, meta, stages=None):
    """Returns (logits [B,S,V], loss or None).  meta["truncation"] + meta["max_sequence_length"]: the reference's truncation branch
    (model.py:505-514): AFTER the splice, the embeddings, labels, mask and position ids are cut to the first max_sequence_length
    positions (S becomes that length); pinned by tests/golden/tiny_clip_llama_trunc.*."""
    e = multimodal_embed(w, batch, meta, stages)
    mask, pos, labels = batch.get("attention_mask"), batch["position_ids"], batch.get("labels")
    msl = meta.get("max_sequence_length")
    if meta.get("truncation") and msl is not None and e.shape[1] > msl:
        e = e[:, :msl, :]
        labels = labels[:, :msl] if labels is not None else None
        mask = mask[:, :msl] if mask is not None else None
        pos = pos[:, :msl] if pos is not None else None
    h = decoder_forward(w, e, mask, pos, meta["llm"], stages=stages)
    logits = F.linear(h, lm_head_weight(w, meta["llm"]))
    loss = causal_lm_loss(logits, labels) if labels is not None else None
    return logits, loss


@torch.no_grad()
def greedy_generate(w, batch, meta, max_new_tokens=8, temperature=0.1, return_logits=False) -> torch.Tensor:
    """return_logits: also return the per-step raw logits [B, n, V] (tests use them to show that a bf16 arg-max flip is a
    near-tie of the reference's own scores)."""
    temperature = max(temperature, 1e-6)
    llm = meta["llm"]
    eos = meta["eos_token_idx"]
    nxt = multimodal_embed(w, batch, meta)
    mask = batch["attention_mask"]
    pos = batch["position_ids"]
    B, S = mask.shape
    cache: List = [None] * llm["num_hidden_layers"]
    finished = torch.zeros(B, dtype=torch.bool)
    toks, step_logits = [], []
    for i in range(max_new_tokens):
        if i > 0:
            pos = (S + i - 1) * torch.ones(B, 1, dtype=torch.long)
            mask = torch.cat([mask, torch.ones(B, 1, dtype=mask.dtype)], dim=-1)
        h = decoder_forward(w, nxt, mask, pos, llm, cache=cache)
        raw = F.linear(h[:, -1, :], lm_head_weight(w, llm))
        step_logits.append(raw.float())
        logits = raw / temperature
        tok = torch.argmax(torch.softmax(logits, dim=-1), dim=-1)
        tok = torch.where(finished, torch.full_like(tok, eos), tok)
        toks.append(tok)
        finished = finished | (tok == eos)
        if bool(finished.all()):
            break
        nxt = F.embedding(tok, w[LLM_PREFIX + "embed_tokens.weight"])[:, None, :]
    if return_logits:
        return torch.stack(toks, dim=1), torch.stack(step_logits, dim=1)
    return torch.stack(toks, dim=1)


# ------------------------------------------------------------------------------------------
# helpers for tests / bench
# ------------------------------------------------------------------------------------------
def load_golden(name: str, golden_dir: str):
    import json
    import os
    from safetensors.torch import load_file
    meta = json.load(open(os.path.join(golden_dir, f"{name}.meta.json")))
    weights = load_file(os.path.join(golden_dir, f"{name}.weights.safetensors"))
    vectors = load_file(os.path.join(golden_dir, f"{name}.vectors.safetensors"))
    return meta, weights, vectors


def golden_batch(vectors, case):
    b = {k: vectors[f"{case}.in.{k}"] for k in ("input_ids", "attention_mask", "position_ids", "labels")}
    if f"{case}.in.pixels" in vectors:
        px = vectors[f"{case}.in.pixels"]
        b["processed_multimodal_inputs"] = {"batch_idx": {"image": vectors[f"{case}.in.batch_idx"]},
                                            "token_range": {"image": vectors[f"{case}.in.token_range"]},
                                            "stacked": {"image": [px[i] for i in range(px.shape[0])]}}
    else:
        b["processed_multimodal_inputs"] = {"batch_idx": {}, "token_range": {}, "stacked": {}}
    return b

#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (read-only, /root/reference) on CPU.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py

Only OUTPUTS are written (tests/golden/*.safetensors / *.json): weights (bf16-rounded, random
init), inputs, per-stage activations, logits, loss, selected grads, greedy token ids, and the
collator's output for mock_dataset/cat.jpg.  No reference source text is copied.

Import recipe = SURVEY.md Appendix A (three harness-side shims, none edits the reference).
The numbers pin the semantics of transformers==5.15.0 CLIP / Llama / Qwen2 modules as called
by the reference at model.py:433-444,517-526,595-640 and image_modality.py:130-137.
"""
import io
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ.setdefault("HF_HUB_OFFLINE", "1")

import numpy as np
import torch
import transformers
from transformers import (AutoConfig, AutoModel, AutoModelForCausalLM, AutoProcessor,  # noqa: F401
                          CLIPConfig, CLIPImageProcessorPil, CLIPModel, LlamaConfig,
                          PreTrainedTokenizerFast, Qwen2Config)
from safetensors.torch import save_file

REF_SRC = "/root/reference/src/multimeditron"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

# ---- shims (harness side) ----------------------------------------------------------------
pkg = types.ModuleType("multimeditron")
pkg.__path__ = [REF_SRC]
sys.modules["multimeditron"] = pkg
tv = types.ModuleType("torchvision")
tvm = types.ModuleType("torchvision.models")
tv.models = tvm
sys.modules["torchvision"] = tv
sys.modules["torchvision.models"] = tvm

from multimeditron.model.model import MultimodalConfig, MultiModalModelForCausalLM, ChatTemplate  # noqa: E402
import multimeditron.model.modalities.image_modality as im  # noqa: E402

im.AutoImageProcessor = types.SimpleNamespace(from_pretrained=CLIPImageProcessorPil.from_pretrained)
from multimeditron.model.modalities import ImageConfig  # noqa: E402
from multimeditron.model.data_loader import DataCollatorForMultimodal  # noqa: E402
from multimeditron.dataset.loader import FileSystemImageLoader, RawImageLoader  # noqa: E402

torch.set_num_threads(8)


def bf16_round_(t: torch.Tensor) -> torch.Tensor:
    return t.copy_(t.to(torch.bfloat16).to(t.dtype))


# ---- tiny model directories ----------------------------------------------------------------
VIS = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
           image_size=56, patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=32)
TXT = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
           vocab_size=64, max_position_embeddings=16, projection_dim=32)
VOCAB = 130
EOS = 129
IMG_START, IMG_END, ATTACH = 126, 127, 125
P = (VIS["image_size"] // VIS["patch_size"]) ** 2  # 16


def make_clip_dir(d, seed):
    torch.manual_seed(seed)
    clip = CLIPModel(CLIPConfig(text_config=TXT, vision_config=VIS, projection_dim=32))
    # HF inits LayerNorm to (1, 0) and biases to 0; perturb them so the fixtures exercise them.
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in clip.named_parameters():
            if p.ndim == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    clip.save_pretrained(d)
    CLIPImageProcessorPil(size={"shortest_edge": VIS["image_size"]},
                          crop_size={"height": VIS["image_size"], "width": VIS["image_size"]}).save_pretrained(d)


def llama_cfg():
    return LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                       num_key_value_heads=1, head_dim=64, vocab_size=128, rms_norm_eps=1e-5,
                       max_position_embeddings=131072, tie_word_embeddings=False,
                       rope_parameters={"rope_type": "llama3", "rope_theta": 500000.0, "factor": 8.0,
                                        "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                        "original_max_position_embeddings": 8192})


def llama_d128_cfg():
    """The HEADLINE attention geometry at a fixture-sized width: head_dim 128, GQA 4:1, llama3 RoPE (Llama-3.1-8B runs
    32/8 heads x 128).  hidden stays 128, so q_proj is 128 -> 512 (HF allows head_dim * heads != hidden)."""
    return LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                       num_key_value_heads=1, head_dim=128, vocab_size=128, rms_norm_eps=1e-5,
                       max_position_embeddings=131072, tie_word_embeddings=False,
                       rope_parameters={"rope_type": "llama3", "rope_theta": 500000.0, "factor": 8.0,
                                        "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                        "original_max_position_embeddings": 8192})


def qwen2_cfg():
    return Qwen2Config(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                       num_key_value_heads=1, vocab_size=128, rms_norm_eps=1e-6, max_position_embeddings=32768,
                       tie_word_embeddings=True, use_sliding_window=False,
                       rope_parameters={"rope_type": "default", "rope_theta": 1000000.0})


def randomize_(model, seed):
    """Re-randomise everything deterministically (post_init policies differ between HF versions), then round to
    bf16-representable values so bf16 and fp32 consumers see identical weights."""
    g = torch.Generator().manual_seed(seed + 11)
    with torch.no_grad():
        seen = set()
        for n, p in model.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            if "layernorm" in n.lower() or "layer_norm" in n or "layrnorm" in n or n.endswith("norm.weight"):
                if n.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif p.ndim == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.06 * torch.randn(p.shape, generator=g))
            bf16_round_(p)


def build_model(llm_cfg, seed, tmp):
    d1 = os.path.join(tmp, "clip")
    d2 = os.path.join(tmp, "llm")
    os.makedirs(d1, exist_ok=True)
    os.makedirs(d2, exist_ok=True)
    make_clip_dir(d1, seed)
    llm_cfg.save_pretrained(d2)
    torch.manual_seed(seed + 7)
    cfg = MultimodalConfig(vocab_size=VOCAB, modalities=[ImageConfig(hidden_size=128, clip_name=d1)],
                           llm_path=d2, dtype="float32", eos_token_idx=EOS, hidden_size=128)
    model = MultiModalModelForCausalLM(cfg)
    randomize_(model, seed)
    return model.eval()


# ---- synthetic batch (hand-built; the collator has its own fixture) ---------------------------
def make_batch(seed, S, layout, pad_side):
    """layout: per sample list of image start positions (index of the first attachment token)."""
    g = torch.Generator().manual_seed(seed)
    B = len(layout)
    lengths = [S if b == 0 else S - 7 for b in range(B)] if pad_side != "none" else [S] * B
    ids = torch.randint(0, 120, (B, S), generator=g)
    mask = torch.ones(B, S, dtype=torch.long)
    labels = ids.clone()
    batch_idx, token_range, pixels = [], [], []
    for b, starts in enumerate(layout):
        L = lengths[b]
        off = S - L if pad_side == "left" else 0
        if L < S:
            if pad_side == "left":
                mask[b, :off] = 0
                ids[b, :off] = EOS
            else:
                mask[b, L:] = 0
                ids[b, L:] = EOS
        for s in starts:
            s = s + off
            ids[b, s - 1] = IMG_START
            ids[b, s:s + P] = ATTACH
            ids[b, s + P] = IMG_END
            labels[b, s - 1:s + P + 1] = -100
            batch_idx += [b] * P
            token_range += list(range(s, s + P))
            pixels.append(bf16_round_(torch.randn(3, VIS["image_size"], VIS["image_size"], generator=g)))
        labels[b, off:off + 6] = -100  # masked "user" prefix
    labels = torch.where(mask == 0, torch.full_like(labels, -100), torch.where(labels == ATTACH, -100, labels))
    labels[ids == IMG_START] = -100
    labels[ids == IMG_END] = -100
    pos = (mask.cumsum(-1) - 1).masked_fill(mask == 0, 0)
    pmi = {"batch_idx": {"image": torch.tensor(batch_idx, dtype=torch.long)},
           "token_range": {"image": torch.tensor(token_range, dtype=torch.long)},
           "stacked": {"image": pixels}}
    if not pixels:
        pmi = {"batch_idx": {}, "token_range": {}, "stacked": {}}
    return dict(input_ids=ids, attention_mask=mask, position_ids=pos, labels=labels,
                processed_multimodal_inputs=pmi)


GRAD_KEYS = ("projector", "model.model.layers.0.", "model.model.embed_tokens", "model.lm_head", "model.model.norm",
             "vision_model.encoder.layers.1.", "vision_model.embeddings", "vision_model.pre_layrnorm")


def run_case(model, batch, tag, out, do_grads=True, do_generate=False):
    acts = {}
    hooks = []
    mod = model.modalities_with_projection[0]
    vm = mod.feature_extractor.vision_model

    def grab(name, pick=lambda o: o):
        def fn(_m, _i, o):
            o = pick(o)
            acts[name] = o.detach().float().clone()
        return fn

    has_img = len(batch["processed_multimodal_inputs"]["stacked"]) > 0
    hooks.append(vm.embeddings.register_forward_hook(grab("vit_embeddings")))
    hooks.append(vm.pre_layrnorm.register_forward_hook(grab("vit_pre_ln")))
    hooks.append(vm.encoder.layers[0].register_forward_hook(grab("vit_layer0", lambda o: o[0] if isinstance(o, tuple) else o)))
    hooks.append(vm.encoder.register_forward_hook(grab("vit_last_hidden", lambda o: o.last_hidden_state)))
    hooks.append(mod.projector.register_forward_hook(grab("projector_out")))
    llm = model.model.model
    hooks.append(llm.layers[0].register_forward_hook(grab("llm_layer0", lambda o: o[0] if isinstance(o, tuple) else o)))
    hooks.append(llm.norm.register_forward_hook(grab("llm_final_norm")))

    model.unfreeze()
    model.zero_grad(set_to_none=True)
    with torch.no_grad():
        spliced = model.embed_modalities_with_text(batch["input_ids"], batch["processed_multimodal_inputs"])
    acts["spliced_embeds"] = spliced.float().clone()
    acts.clear() if False else None
    o = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
              position_ids=batch["position_ids"], labels=batch["labels"],
              processed_multimodal_inputs=batch["processed_multimodal_inputs"])
    for h in hooks:
        h.remove()
    out[f"{tag}.logits"] = o.logits.detach().float()
    out[f"{tag}.loss"] = o.loss.detach().float().reshape(1)
    for k, v in acts.items():
        if not has_img and k.startswith(("vit", "proj")):
            continue
        out[f"{tag}.act.{k}"] = v
    if do_grads:
        o.loss.backward()
        for n, p in model.named_parameters():
            if p.grad is not None and any(k in n for k in GRAD_KEYS):
                out[f"{tag}.grad.{n}"] = p.grad.detach().float().clone()
        model.zero_grad(set_to_none=True)
    for k in ("input_ids", "attention_mask", "position_ids", "labels"):
        out[f"{tag}.in.{k}"] = batch[k].clone()
    pmi = batch["processed_multimodal_inputs"]
    if has_img:
        out[f"{tag}.in.batch_idx"] = pmi["batch_idx"]["image"].clone()
        out[f"{tag}.in.token_range"] = pmi["token_range"]["image"].clone()
        out[f"{tag}.in.pixels"] = torch.stack(pmi["stacked"]["image"]).clone()
    if do_generate:
        for T in (0.1, 0.7):
            with torch.no_grad():
                ids = model.generate(batch, max_new_tokens=8, temperature=T, do_sample=False)
            out[f"{tag}.greedy_T{T}"] = ids.clone()


def model_fixture(name, llm_cfg, seed, long_seq=False):
    with tempfile.TemporaryDirectory() as tmp:
        model = build_model(llm_cfg, seed, tmp)
        out = {}
        if long_seq:    # sequences that cross the D = 128 kernels' 64-key tiles and 256-row query blocks
            run_case(model, make_batch(seed + 1, 300, [[3], [2, 150]], "right"), "right", out)
            run_case(model, make_batch(seed + 2, 200, [[4], [2, 122]], "left"), "left", out, do_grads=False, do_generate=True)
            run_case(model, make_batch(seed + 4, 330, [[2, 81, 160, 260]], "none"), "interleaved4", out, do_grads=False)
        else:
            run_case(model, make_batch(seed + 1, 48, [[3], [2, 24]], "right"), "right", out)
            run_case(model, make_batch(seed + 2, 48, [[4], [2, 22]], "left"), "left", out, do_grads=False, do_generate=True)
            run_case(model, make_batch(seed + 3, 40, [[], []], "none"), "textonly", out, do_grads=False, do_generate=True)
            run_case(model, make_batch(seed + 4, 80, [[2, 21, 40, 60]], "none"), "interleaved4", out, do_grads=False)
        # weights (only what the hot path uses: vision tower, projector, LLM)
        w = {}
        for n, p in model.state_dict().items():
            if "text_model" in n or "text_projection" in n or "visual_projection" in n or "logit_scale" in n \
                    or "position_ids" in n or "post_layernorm" in n:
                continue
            w[n] = p.detach().to(torch.bfloat16).contiguous().clone()
        save_file(w, os.path.join(OUT, f"{name}.weights.safetensors"))
        save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(OUT, f"{name}.vectors.safetensors"))
        meta = dict(name=name, vision=VIS, llm=llm_cfg.to_dict(), vocab_size=VOCAB, eos_token_idx=EOS,
                    image_start=IMG_START, image_end=IMG_END, attachment=ATTACH, num_patches=P,
                    transformers=transformers.__version__, torch=torch.__version__,
                    cases=sorted({k.split(".")[0] for k in out}))
        meta["llm"] = {k: v for k, v in meta["llm"].items()
                       if isinstance(v, (int, float, str, bool, dict, type(None), list))}
        with open(os.path.join(OUT, f"{name}.meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True, default=str)
        print(name, "weights", sum(v.numel() for v in w.values()), "vector tensors", len(out))



def trunc_fixture(name="tiny_clip_llama_trunc", seed=100, msl=40):
    """The truncation branch of the reference's forward (model.py:505-514; the shipped MoE recipes set `truncation: true`): the model of
    `tiny_clip_llama` (same seed -> same weights, tiny_clip_llama.weights.safetensors) with config.truncation = True and
    max_sequence_length = msl < S.  Stored: the batch, the TRUNCATED logits, the loss and selected gradients."""
    with tempfile.TemporaryDirectory() as tmp:
        model = build_model(llama_cfg(), seed, tmp)
        model.config.truncation = True
        model.config.max_sequence_length = msl
        out = {}
        run_case(model, make_batch(seed + 11, 56, [[3], [2, 28]], "right"), "trunc_right", out)        # second image span 28..37 ends before msl
        run_case(model, make_batch(seed + 12, 64, [[2, 30]], "none"), "trunc_cut_image", out)             # ... and one cut in the middle of its span
        keep = {k: v.contiguous() for k, v in out.items() if ".act." not in k}
        save_file(keep, os.path.join(OUT, f"{name}.vectors.safetensors"))
        with open(os.path.join(OUT, f"{name}.meta.json"), "w") as f:
            json.dump(dict(name=name, weights="tiny_clip_llama.weights.safetensors", max_sequence_length=msl, truncation=True,
                           cases=sorted({k.split(".")[0] for k in keep}), transformers=transformers.__version__, torch=torch.__version__),
                      f, indent=1, sort_keys=True)
        print(name, "vector tensors", len(keep), {k: tuple(v.shape) for k, v in keep.items() if k.endswith("logits")})


# ---- SigLIP plug-in (BASELINE config 5) ---------------------------------------------------------
# The reference ships no SigLIP modality: config 5 plugs an alternate embedder in through
# the reference's OWN plug-in protocol (BaseModalityConfig / BaseModalityProcessor / BaseModality + AutoModality.register,
# base.py:10-196).  The class below is harness code written against that protocol; the arithmetic it pins is HF
# transformers' SiglipVisionModel (.last_hidden_state = post_layernorm(encoder(...)), no CLS token, conv bias,
# gelu_pytorch_tanh) followed by the reference's MLPProjector and the reference's splice + LLM.
SIG = dict(hidden_size=144, intermediate_size=264, num_hidden_layers=2, num_attention_heads=2,   # head_dim 72, as so400m
           image_size=56, patch_size=14, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)


def register_siglip_plug():
    from transformers import SiglipVisionModel
    from multimeditron.model.modalities.base import AutoModality, BaseModality, BaseModalityConfig, BaseModalityProcessor
    from multimeditron.model.projectors.mlp import MLPProjector
    if "meditron_siglip" in AutoModality._registry:
        return AutoModality._registry["meditron_siglip"].config_class

    class SiglipImageConfig(BaseModalityConfig):
        def __init__(self, hidden_size=4096, clip_name="google/siglip-so400m-patch14-384", **kwargs):
            super().__init__(modality_type="image", hidden_size=hidden_size, **kwargs)
            self.clip_name = clip_name

    class SiglipImageProcessor(BaseModalityProcessor):
        def process(self, modality):
            raise NotImplementedError("fixtures feed pixel tensors directly")

    @AutoModality.register("meditron_siglip")
    class SiglipImageModality(BaseModality):
        config_class = SiglipImageConfig
        preprocessor_class = SiglipImageProcessor

        def __init__(self, config):
            super().__init__(config)
            self.feature_extractor = SiglipVisionModel.from_pretrained(config.clip_name)
            self.embedding_size = self.feature_extractor.config.hidden_size
            self.projector = MLPProjector(self.embedding_size, config.hidden_size, dtype=self.dtype)

        def forward(self, inputs):
            x = torch.stack(inputs, dim=0).to(self.feature_extractor.device)
            return self.projector(self.feature_extractor(pixel_values=x).last_hidden_state)

        def freeze_modality_embedder(self):
            for q in self.feature_extractor.parameters():
                q.requires_grad = False

        def unfreeze_modality_embedder(self):
            for q in self.feature_extractor.parameters():
                q.requires_grad = True

        def unfreeze_projection(self):
            for q in self.projector.parameters():
                q.requires_grad = True

    return SiglipImageConfig


def siglip_fixture(name="tiny_siglip_qwen2", seed=300):
    from transformers import SiglipVisionConfig, SiglipVisionModel
    import multimeditron.model.model as mm
    SiglipImageConfig = register_siglip_plug()
    with tempfile.TemporaryDirectory() as tmp:
        d1, d2 = os.path.join(tmp, "siglip"), os.path.join(tmp, "llm")
        torch.manual_seed(seed)
        SiglipVisionModel(SiglipVisionConfig(**SIG)).save_pretrained(d1)
        qwen2_cfg().save_pretrained(d2)
        cfg = MultimodalConfig(vocab_size=VOCAB, modalities=[SiglipImageConfig(hidden_size=128, clip_name=d1)],
                               llm_path=d2, dtype="float32", eos_token_idx=EOS, hidden_size=128)
        # the reference builds processors with AutoModality.preprocessor_from_name(..., config); ours takes the config
        model = MultiModalModelForCausalLM(cfg)
        g = torch.Generator().manual_seed(seed + 11)
        with torch.no_grad():
            seen = set()
            for n, q in model.named_parameters():
                if id(q) in seen:
                    continue
                seen.add(id(q))
                if "layernorm" in n.lower() or "layer_norm" in n or n.endswith("norm.weight"):
                    q.copy_((1.0 if n.endswith("weight") else 0.0) + 0.1 * torch.randn(q.shape, generator=g))
                elif "probe" in n:
                    q.copy_(0.06 * torch.randn(q.shape, generator=g))
                elif q.ndim == 1:
                    q.copy_(0.1 * torch.randn(q.shape, generator=g))
                else:
                    q.copy_(0.06 * torch.randn(q.shape, generator=g))
                bf16_round_(q)
        model = model.eval()
        out = {}
        mod = model.modalities_with_projection[0]
        vm = mod.feature_extractor      # transformers 5.x: SiglipVisionModel holds embeddings/encoder/post_layernorm itself

        def run(batch, tag, do_grads, do_generate):
            acts, hooks = {}, []

            def grab(key, pick=lambda o: o):
                def fn(_m, _i, o):
                    acts[key] = pick(o).detach().float().clone()
                return fn
            hooks.append(vm.embeddings.register_forward_hook(grab("vit_embeddings")))
            hooks.append(vm.encoder.layers[0].register_forward_hook(grab("vit_layer0", lambda o: o[0] if isinstance(o, tuple) else o)))
            hooks.append(vm.post_layernorm.register_forward_hook(grab("vit_last_hidden")))
            hooks.append(mod.projector.register_forward_hook(grab("projector_out")))
            hooks.append(model.model.model.layers[0].register_forward_hook(grab("llm_layer0", lambda o: o[0] if isinstance(o, tuple) else o)))
            hooks.append(model.model.model.norm.register_forward_hook(grab("llm_final_norm")))
            model.unfreeze()
            model.zero_grad(set_to_none=True)
            with torch.no_grad():
                acts_sp = model.embed_modalities_with_text(batch["input_ids"], batch["processed_multimodal_inputs"]).float().clone()
            o = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], position_ids=batch["position_ids"],
                      labels=batch["labels"], processed_multimodal_inputs=batch["processed_multimodal_inputs"])
            for hk in hooks:
                hk.remove()
            out[f"{tag}.logits"] = o.logits.detach().float()
            out[f"{tag}.loss"] = o.loss.detach().float().reshape(1)
            out[f"{tag}.act.spliced_embeds"] = acts_sp
            for k, v in acts.items():
                out[f"{tag}.act.{k}"] = v
            if do_grads:
                o.loss.backward()
                keys = ("projector", "model.model.layers.0.", "model.model.embed_tokens", "model.model.norm",
                        "feature_extractor.encoder.layers.1.", "feature_extractor.embeddings", "feature_extractor.post_layernorm")
                for n, q in model.named_parameters():
                    if q.grad is not None and any(k in n for k in keys):
                        out[f"{tag}.grad.{n}"] = q.grad.detach().float().clone()
                model.zero_grad(set_to_none=True)
            for k in ("input_ids", "attention_mask", "position_ids", "labels"):
                out[f"{tag}.in.{k}"] = batch[k].clone()
            pmi = batch["processed_multimodal_inputs"]
            out[f"{tag}.in.batch_idx"] = pmi["batch_idx"]["image"].clone()
            out[f"{tag}.in.token_range"] = pmi["token_range"]["image"].clone()
            out[f"{tag}.in.pixels"] = torch.stack(pmi["stacked"]["image"]).clone()
            if do_generate:
                for T in (0.1, 0.7):
                    with torch.no_grad():
                        ids = model.generate(batch, max_new_tokens=8, temperature=T, do_sample=False)
                    out[f"{tag}.greedy_T{T}"] = ids.clone()

        run(make_batch(seed + 1, 48, [[3], [2, 24]], "right"), "right", True, False)
        run(make_batch(seed + 2, 48, [[4], [2, 22]], "left"), "left", False, True)
        w = {}
        for n, q in model.state_dict().items():
            if ".head." in n or "position_ids" in n:      # pooling head: not on the token path
                continue
            w[n] = q.detach().to(torch.bfloat16).contiguous().clone()
        save_file(w, os.path.join(OUT, f"{name}.weights.safetensors"))
        save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(OUT, f"{name}.vectors.safetensors"))
        llm = {k: v for k, v in qwen2_cfg().to_dict().items() if isinstance(v, (int, float, str, bool, dict, type(None), list))}
        meta = dict(name=name, vision=dict(SIG, kind="siglip"), llm=llm, vocab_size=VOCAB, eos_token_idx=EOS, image_start=IMG_START,
                    image_end=IMG_END, attachment=ATTACH, num_patches=P, transformers=transformers.__version__,
                    torch=torch.__version__, cases=sorted({k.split(".")[0] for k in out}),
                    note="SigLIP plug-in written against the reference's modality protocol (harness code in tools/make_golden.py); "
                         "vision arithmetic = HF SiglipVisionModel, projector/splice/LLM = the reference")
        with open(os.path.join(OUT, f"{name}.meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True, default=str)
        print(name, "weights", sum(v.numel() for v in w.values()), "vector tensors", len(out))


# ---- MoE image modality -------------------------------------------------------------
# reference modalities/image_modality_moe.py:152-210: E CLIP vision towers + a gating network + one of three fusions
# (sequence_append / weighted_average / cross_attn over model/attention.py:48-101) + the MLP projector.  The gate is a
# torchvision ResNet-50 (absent here: its arithmetic stays "parity unpinned"); the harness plugs a stub with the gate's
# OUTPUT CONTRACT (logits, top-k indices, softmax weights) so that everything downstream of the gate is pinned:
#     weights = softmax(mean_hw(pixels) @ Wg^T + bg)
def moe_fixture(name="tiny_moe_clip", seed=500, pep=False):
    """pep=False: the reference's MOEImageModality (one projector after the fusion); pep=True: MOEImageModalityPEP
    (image_modality_moe_pep.py: one projector PER EXPERT, fusion and cross-attention in the projected space)."""
    if pep:
        import multimeditron.model.modalities.image_modality_moe_pep as moe
        moe.MOEImageConfig, moe.MOEImageModality = moe.MOEImageConfigPEP, moe.MOEImageModalityPEP
        moe.AutoModel = types.SimpleNamespace(from_pretrained=lambda path, trust_remote_code=True: transformers.CLIPModel.from_pretrained(path))
    else:
        import multimeditron.model.modalities.image_modality_moe as moe
    moe.AutoImageProcessor = types.SimpleNamespace(from_pretrained=CLIPImageProcessorPil.from_pretrained)
    E = 3

    class StubGate(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(seed + 5)
            self.w = torch.nn.Parameter(bf16_round_(torch.randn(E, 3, generator=g)))
            self.b = torch.nn.Parameter(bf16_round_(0.3 * torch.randn(E, generator=g)))
            self.config = types.SimpleNamespace(class_names=[])

        def forward(self, px):
            logits = px.float().mean(dim=(2, 3)) @ self.w.t() + self.b
            return logits, logits.topk(1, dim=-1).indices, torch.softmax(logits, dim=-1)

    moe.GatingNetwork = types.SimpleNamespace(from_pretrained=lambda path: StubGate())
    out, weights = {}, None
    with tempfile.TemporaryDirectory() as tmp:
        dirs = []
        for e in range(E):
            d = os.path.join(tmp, f"clip{e}")
            os.makedirs(d)
            make_clip_dir(d, seed + 10 * e)
            dirs.append(d)
        g = torch.Generator().manual_seed(seed + 1)
        n = 3
        pixels = [bf16_round_(torch.randn(3, VIS["image_size"], VIS["image_size"], generator=g)) for _ in range(n)]
        for fusion in ("weighted_average", "sequence_append", "cross_attn"):
            torch.manual_seed(seed + 7)
            cfg = moe.MOEImageConfig(hidden_size=128, expert_clip_names=dirs, image_processor=dirs[0], gating_path="stub",
                                     top_k_experts=E, generalist_idx=E - 1, fusion_method=fusion, cross_attn_heads=2)
            m = moe.MOEImageModality(cfg).float()
            import zlib
            with torch.no_grad():
                for nme, q in sorted(m.named_parameters()):
                    if nme.startswith("gating_network."):
                        continue
                    gi = torch.Generator().manual_seed(seed + zlib.crc32(nme.encode()) % 100003)   # per name: every fusion variant
                                                                                                    # gets the same shared weights
                    if "layer_norm" in nme or "layrnorm" in nme:
                        q.copy_((1.0 if nme.endswith("weight") else 0.0) + 0.1 * torch.randn(q.shape, generator=gi))
                    elif q.ndim == 1:
                        q.copy_(0.1 * torch.randn(q.shape, generator=gi))
                    else:
                        q.copy_(0.06 * torch.randn(q.shape, generator=gi))
                    bf16_round_(q)
            m.eval()                                           # dropout (attn_drop / proj_drop 0.1) off: no RNG in the fixture
            for q in m.parameters():
                q.requires_grad_(True)
            y = m(pixels)
            G = bf16_round_(torch.randn(y.shape, generator=torch.Generator().manual_seed(seed + 13)))
            (y * G).sum().backward()
            out[f"{fusion}.out"] = y.detach().float().clone()
            out[f"{fusion}.dout"] = G
            for nme, q in m.named_parameters():
                if q.grad is not None and not nme.startswith("gating_network.") and (
                        "projector" in nme or "cross_attn" in nme or "experts.0.encoder.layers.1" in nme or "experts.2.embeddings" in nme
                        or "experts.1.encoder.layers.0.mlp" in nme):     # "projector" also matches PEP's projectors.{e}.*
                    out[f"{fusion}.grad.{nme}"] = q.grad.detach().float().clone()
            with torch.no_grad():
                _, _, gw = m.gating_network(torch.stack(pixels))
            out[f"{fusion}.gate_weights"] = gw.float().clone()
            sd = {k: v.detach().to(torch.bfloat16).contiguous().clone() for k, v in m.state_dict().items()
                  if not k.startswith("gating_network.") and "position_ids" not in k and "post_layernorm" not in k
                  and not k.startswith("_gating")}
            if weights is None:
                weights = {}
            for k, v in sd.items():        # cross_attn.* exists only in the cross_attn variant; the rest is shared
                if k in weights:
                    assert torch.equal(weights[k], v), k
                weights[k] = v
        out["pixels"] = torch.stack(pixels)
        gate = StubGate()
        out["gate.w"], out["gate.b"] = gate.w.detach().clone(), gate.b.detach().clone()
    save_file(weights, os.path.join(OUT, f"{name}.weights.safetensors"))
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(OUT, f"{name}.vectors.safetensors"))
    with open(os.path.join(OUT, f"{name}.meta.json"), "w") as f:
        json.dump(dict(name=name, vision=VIS, num_experts=E, generalist_idx=E - 1, cross_attn_heads=2, hidden_size=128, per_expert_projection=bool(pep),
                       fusions=["weighted_average", "sequence_append", "cross_attn"], transformers=transformers.__version__,
                       torch=torch.__version__,
                       note="reference MOEImageModality in eval mode (dropout off) with a stub gate standing for the torchvision "
                            "ResNet-50 GatingNetwork: weights = softmax(mean_hw(pixels) @ Wg^T + bg)"), f, indent=1, sort_keys=True)
    print(name, "weights", sum(v.numel() for v in weights.values()), "vector tensors", len(out))

# ---- collator fixture ------------------------------------------------------------------------
LLAMA3_TEMPLATE = (
    "{% for message in messages %}"
    "{{ '<|start_header_id|> ' + message['role'] + ' <|end_header_id|> ' + message['content'] + ' <|eot_id|> ' }}"
    "{% endfor %}"
    "{% if add_generation_prompt %}{{ '<|start_header_id|> assistant <|end_header_id|> ' }}{% endif %}")

WORDS = ("<|eot_id|> <|start_header_id|> <|end_header_id|> <|image_start|> <|image_end|> <|attachment|> <unk> "
         "system user assistant describe the image in detail what is this a cat sitting on grass it looks at "
         "camera you are helpful and there two pictures first second nice").split()


def ckpt_fixture(name="ckpt_ref", seed=700):
    """A checkpoint directory written by the REFERENCE's own `save_pretrained` (HF PreTrainedModel.save_pretrained on the reference
    class: config.json in `MultimodalConfig.to_dict` layout, model.py:152-202, + model.safetensors under the reference's parameter
    names, the unused CLIP text tower included), committed as data under tests/golden/ckpt_ref/ together with the logits the
    reference computes from it.  `clip_name` / `llm_path` are the relative directories clip/ and llm/ inside it (config files only).

    Second direction, checked here because the reference only runs in this container: the BUILD loads that directory, writes its
    own `save_pretrained` output, and the REFERENCE's `from_pretrained` loads that and must reproduce the same logits."""
    import shutil
    d = os.path.join(OUT, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    cwd = os.getcwd()
    os.chdir(d)
    try:
        make_clip_dir("clip", seed)
        llama_cfg().save_pretrained("llm")
        torch.manual_seed(seed + 7)
        cfg = MultimodalConfig(vocab_size=VOCAB, modalities=[ImageConfig(hidden_size=128, clip_name="clip")], llm_path="llm",
                               dtype="float32", eos_token_idx=EOS, hidden_size=128)
        model = MultiModalModelForCausalLM(cfg)
        randomize_(model, seed)
        model.eval()
        model.save_pretrained(".", safe_serialization=True)
        batch = make_batch(seed + 1, 40, [[3], [1, 20]], "right")
        with torch.no_grad():
            o = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], position_ids=batch["position_ids"],
                      labels=batch["labels"], processed_multimodal_inputs=batch["processed_multimodal_inputs"])
        vec = {"logits": o.logits.float(), "loss": o.loss.float().reshape(1)}
        for k in ("input_ids", "attention_mask", "position_ids", "labels"):
            vec[f"in.{k}"] = batch[k].clone()
        pmi = batch["processed_multimodal_inputs"]
        vec["in.batch_idx"], vec["in.token_range"] = pmi["batch_idx"]["image"].clone(), pmi["token_range"]["image"].clone()
        vec["in.pixels"] = torch.stack(pmi["stacked"]["image"]).clone()
        save_file({k: v.contiguous() for k, v in vec.items()}, "vectors.safetensors")
        # ---- build -> reference
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from multimeditron_amd.model.model import MultiModalModelForCausalLM as Ours
        ours = Ours.from_pretrained(".", device="cpu", strict=True)
        back = tempfile.mkdtemp(prefix="build_ckpt_")
        ours.save_pretrained(back)
        # The reference's own `from_pretrained` cannot run under transformers 5.15, not even on the directory it has just written:
        # HF builds the model under a meta-device context and ImageModality.__init__ calls AutoModel.from_pretrained inside it
        # (image_modality.py:124) -> RuntimeError.  Recorded in the fixture's meta; the reverse direction is therefore checked
        # the way HF's loader ends: a reference-constructed model + load_state_dict of the file the build wrote.
        own_error = None
        try:
            MultiModalModelForCausalLM.from_pretrained(".")
        except Exception as e:                                   # noqa: BLE001
            own_error = f"{type(e).__name__}: {str(e).splitlines()[0]}"[:200]
        from safetensors.torch import load_file
        torch.manual_seed(seed + 99)                             # different init: every tensor that matters must come from the file
        # (return_unused_kwargs=True: the only branch of the reference's from_dict that works)
        ref2 = MultiModalModelForCausalLM(MultimodalConfig.from_dict(json.load(open(os.path.join(back, "config.json"))),
                                                                      return_unused_kwargs=True)[0])
        res = ref2.load_state_dict(load_file(os.path.join(back, "model.safetensors")), strict=False)
        info = {"missing_keys": list(res.missing_keys), "unexpected_keys": list(res.unexpected_keys)}
        ref2.eval()
        with torch.no_grad():
            o2 = ref2(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], position_ids=batch["position_ids"],
                      labels=batch["labels"], processed_multimodal_inputs=batch["processed_multimodal_inputs"])
        diff = float((o2.logits.float() - o.logits.float()).abs().max())
        missing = sorted(info.get("missing_keys", []))
        unexpected = sorted(info.get("unexpected_keys", []))
        assert diff == 0.0, diff
        assert not unexpected, unexpected[:5]
        assert all(("text_model" in k or "text_projection" in k or "visual_projection" in k or "logit_scale" in k or "post_layernorm" in k)
                   for k in missing), [k for k in missing if "text_model" not in k][:5]
        shutil.rmtree(back, ignore_errors=True)
        meta = {"written_by": "reference MultiModalModelForCausalLM.save_pretrained (transformers %s)" % transformers.__version__,
                "files": sorted(f for f in os.listdir(".") if os.path.isfile(f)),
                "reference_loads_build_checkpoint": True, "max_abs_logit_diff_reference_vs_reference_via_build": diff,
                "reference_from_pretrained_on_its_own_directory": own_error or "ok",
                "keys_the_build_does_not_write": len(missing), "vocab_size": VOCAB, "eos_token_idx": EOS,
                "llm": json.load(open("llm/config.json")), "vision": VIS}
        with open("fixture.meta.json", "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        # the CLIP weights live in the checkpoint itself; the tower directory keeps its config files only
        for fn in os.listdir("clip"):
            if fn.endswith((".safetensors", ".bin")):
                os.remove(os.path.join("clip", fn))
        print(name, "files", meta["files"], "| reference <- build: max |dlogits| =", diff, "| keys left to re-init:", len(missing))
    finally:
        os.chdir(cwd)


def make_tokenizer():
    from tokenizers import Tokenizer, models, pre_tokenizers
    vocab = {w: i for i, w in enumerate(WORDS)}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    t = PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|eot_id|>", unk_token="<unk>",
                                additional_special_tokens=["<|start_header_id|>", "<|end_header_id|>",
                                                           "<|image_start|>", "<|image_end|>", "<|attachment|>"],
                                chat_template=LLAMA3_TEMPLATE)
    t.pad_token = t.eos_token
    return t


def llama_spaced_template():
    # The synthetic WordLevel tokenizer splits on whitespace, so role tags carry spaces.
    ct = ChatTemplate.llama()
    for role in ct.delimiters:
        ct.delimiters[role] = {"start": f"<|start_header_id|> {role} <|end_header_id|>", "end": "<|eot_id|>"}
    return ct


def collator_fixture():
    from PIL import Image
    with tempfile.TemporaryDirectory() as tmp:
        make_clip_dir(tmp, 5)
        proc = im.ImageProcessor(ImageConfig(hidden_size=128, clip_name=tmp))
        cat = "/root/reference/mock_dataset/cat.jpg"
        with open(cat, "rb") as f:
            cat_bytes = f.read()
        out = {}
        meta = {"words": WORDS, "chat_template": LLAMA3_TEMPLATE, "attachment_token": "<|attachment|>",
                "image_size": VIS["image_size"], "patch_size": VIS["patch_size"], "cases": {}}
        samples_conv = [
            {"conversations": [{"role": "system", "content": "you are helpful"},
                               {"role": "user", "content": "<|attachment|> describe the image in detail"},
                               {"role": "assistant", "content": "a cat sitting on grass"}],
             "modalities": [{"type": "image", "value": "cat.jpg"}]},
            {"conversations": [{"role": "user", "content": "what is this"},
                               {"role": "assistant", "content": "nice"}],
             "modalities": []},
            {"conversations": [{"role": "user", "content": "first <|attachment|> and second <|attachment|> what is this"},
                               {"role": "assistant", "content": "two pictures"},
                               {"role": "user", "content": "describe the first"},
                               {"role": "assistant", "content": "it looks at the camera"}],
             "modalities": [{"type": "image", "value": "cat.jpg"}, {"type": "image", "value": "EPFL_campus_2017.jpg"}]},
        ]
        for side in ("right", "left"):
            for gen in (False, True):
                tok = make_tokenizer()
                tok.padding_side = side
                coll = DataCollatorForMultimodal(tokenizer=tok, modality_processors={"image": proc},
                                                 modality_loaders={"image": FileSystemImageLoader("/root/reference/mock_dataset")},
                                                 attachment_token="<|attachment|>", chat_template=llama_spaced_template(),
                                                 add_generation_prompt=gen)
                import copy
                b = coll(copy.deepcopy(samples_conv))
                tag = f"conv_{side}_gen{int(gen)}"
                for k in ("input_ids", "labels", "attention_mask", "position_ids"):
                    out[f"{tag}.{k}"] = b[k].clone()
                out[f"{tag}.batch_idx"] = b["processed_multimodal_inputs"]["batch_idx"]["image"].clone()
                out[f"{tag}.token_range"] = b["processed_multimodal_inputs"]["token_range"]["image"].clone()
                out[f"{tag}.pixels"] = torch.stack(b["processed_multimodal_inputs"]["stacked"]["image"]).clone()
                meta["cases"][tag] = {"padding_side": side, "add_generation_prompt": gen, "kind": "conversations"}
        # 2-D position ids (data_loader.py:159-188 + image_modality.py:99-108): [B, S, 2]; no supported LLM consumes them, the
        # collator branch is pinned all the same
        proc2d = im.ImageProcessor(ImageConfig(hidden_size=128, clip_name=tmp, use_2d_position_ids=True))
        for side in ("right", "left"):
            tok = make_tokenizer()
            tok.padding_side = side
            coll = DataCollatorForMultimodal(tokenizer=tok, modality_processors={"image": proc2d},
                                             modality_loaders={"image": FileSystemImageLoader("/root/reference/mock_dataset")},
                                             attachment_token="<|attachment|>", chat_template=llama_spaced_template(),
                                             use_2d_position_ids=True)
            import copy
            b = coll(copy.deepcopy(samples_conv))
            tag = f"pos2d_{side}"
            for k in ("input_ids", "attention_mask", "position_ids"):
                out[f"{tag}.{k}"] = b[k].clone()
            assert out[f"{tag}.position_ids"].dim() == 3
            meta["cases"][tag] = {"padding_side": side, "kind": "conversations", "use_2d_position_ids": True}
        # text samples through the raw-image (bytes) loader
        tok = make_tokenizer()
        tok.padding_side = "right"
        coll = DataCollatorForMultimodal(tokenizer=tok, modality_processors={"image": proc},
                                         modality_loaders={"image": RawImageLoader()},
                                         attachment_token="<|attachment|>", chat_template=llama_spaced_template())
        samples_text = [{"text": "a cat <|attachment|> sitting on grass", "modalities": [{"type": "image", "value": {"bytes": cat_bytes}}]},
                        {"text": "what is this", "modalities": []}]
        try:
            b = coll(samples_text)
            tag = "text_right"
            for k in ("input_ids", "labels", "attention_mask", "position_ids"):
                out[f"{tag}.{k}"] = b[k].clone()
            out[f"{tag}.batch_idx"] = b["processed_multimodal_inputs"]["batch_idx"]["image"].clone()
            out[f"{tag}.token_range"] = b["processed_multimodal_inputs"]["token_range"]["image"].clone()
            out[f"{tag}.pixels"] = torch.stack(b["processed_multimodal_inputs"]["stacked"]["image"]).clone()
            meta["cases"][tag] = {"padding_side": "right", "kind": "text"}
        except Exception as e:  # the reference's text branch tokenizes a ragged batch with return_tensors="pt"
            meta["text_branch_error"] = f"{type(e).__name__}: {e}"[:300]
            # one-sample text batch (no raggedness)
            b = coll(samples_text[:1])
            tag = "text_single"
            for k in ("input_ids", "labels", "attention_mask", "position_ids"):
                out[f"{tag}.{k}"] = b[k].clone()
            out[f"{tag}.batch_idx"] = b["processed_multimodal_inputs"]["batch_idx"]["image"].clone()
            out[f"{tag}.token_range"] = b["processed_multimodal_inputs"]["token_range"]["image"].clone()
            out[f"{tag}.pixels"] = torch.stack(b["processed_multimodal_inputs"]["stacked"]["image"]).clone()
            meta["cases"][tag] = {"padding_side": "right", "kind": "text"}
        meta["samples_conv"] = samples_conv
        meta["samples_text"] = [{"text": s["text"], "n_images": len(s["modalities"])} for s in samples_text]
        save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(OUT, "collator.vectors.safetensors"))
        with open(os.path.join(OUT, "collator.meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        # the two mock images are DATA held by the reference's repo; tests need them as inputs
        import shutil
        os.makedirs(os.path.join(OUT, "mock_dataset"), exist_ok=True)
        for fn in ("cat.jpg", "EPFL_campus_2017.jpg"):
            shutil.copyfile(os.path.join("/root/reference/mock_dataset", fn), os.path.join(OUT, "mock_dataset", fn))
        print("collator cases", list(meta["cases"]))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["llama", "trunc", "qwen2", "llama_d128", "siglip", "moe", "moe_pep", "collator", "ckpt"]
    if "llama" in which:
        model_fixture("tiny_clip_llama", llama_cfg(), 100)
    if "trunc" in which:
        trunc_fixture()
    if "qwen2" in which:
        model_fixture("tiny_clip_qwen2", qwen2_cfg(), 200)
    if "llama_d128" in which:
        model_fixture("tiny_clip_llama_d128", llama_d128_cfg(), 400, long_seq=True)
    if "siglip" in which:
        siglip_fixture()
    if "moe" in which:
        moe_fixture()
    if "moe_pep" in which:
        moe_fixture("tiny_moe_clip_pep", 600, pep=True)
    if "collator" in which:
        collator_fixture()
    if "ckpt" in which:
        ckpt_fixture()

"""Drop-in for the reference's `multimeditron.model.model` (model.py:17-673): `ChatTemplate`, `MultimodalConfig`,
`MultiModalModelForCausalLM` (forward / generate / freeze policies / processors) and `bootstrap`, with the whole
numerical path -- modality encoder, projector, embed-splice, decoder, loss -- on libmmhip (gfx950) kernels.

Differences from the reference that a caller can observe (all documented in DESIGN.md):
  * the LLM and the CLIP tower are this package's own modules (model/llm.py, model/vision.py), not HF classes;
    parameter names are identical so checkpoints interchange; hub names resolve to built-in shape presets;
  * `forward` returns a small dataclass with `.loss`, `.logits`, `.past_key_values` (HF `CausalLMOutputWithPast` shape);
    `.logits` is a [B,S,V] view whose row stride is padded to a multiple of 64 elements;
  * there is no eager/CPU fallback: without libmmhip.so and a GPU every compute call raises."""
from __future__ import annotations

import json
import logging
import os
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Union

import torch
import torch.nn as nn

from .. import functional as Fm
from .. import kernels as K
from ..nn import FlatParams, grad_dummy
from ..utils import get_torch_dtype, trace_range
from .llm import CausalLM, CausalLMOutput, LLMConfig
from .modalities import AutoModality, BaseModality, BaseModalityConfig, BaseModalityProcessor
from .presets import resolve_llm_config

logger = logging.getLogger(__name__)


@dataclass
class ChatTemplate:
    """Role delimiters + special tokens per LLM family (reference model.py:17-99)."""
    name: str = "custom"
    delimiters: Dict[str, Dict[str, str]] = field(default_factory=dict)
    special_tokens: Dict[str, str] = field(default_factory=dict)

    _IMAGE_TOKENS = {"image_start": "<|image_start|>", "image_end": "<|image_end|>"}

    @staticmethod
    def from_name(name: str) -> "ChatTemplate":
        makers = {"llama": ChatTemplate.llama, "apertus": ChatTemplate.apertus, "qwen3": ChatTemplate.qwen3}
        if name not in makers:
            raise ValueError(f"Unknown chat template name: {name}")
        return makers[name]()

    @staticmethod
    def _make(name, roles):
        return ChatTemplate(name=name, delimiters={r: {"start": s, "end": e} for r, (s, e) in roles.items()},
                            special_tokens=dict(ChatTemplate._IMAGE_TOKENS))

    @staticmethod
    def llama() -> "ChatTemplate":
        hdr = "<|start_header_id|>{}<|end_header_id|>"
        return ChatTemplate._make("llama", {r: (hdr.format(r), "<|eot_id|>") for r in ("system", "user", "assistant")})

    @staticmethod
    def apertus() -> "ChatTemplate":
        return ChatTemplate._make("apertus", {r: (f"<|{r}_start|>", f"<|{r}_end|>")
                                              for r in ("system", "developer", "user", "assistant")})

    @staticmethod
    def qwen3() -> "ChatTemplate":
        return ChatTemplate._make("qwen3", {r: (f"<|im_start|>{r}", "<|im_end|>") for r in ("system", "user", "assistant")})


class MultimodalConfig:
    """reference model.py:103-202 (same constructor arguments and dict layout)."""
    model_type = "multimodal"

    def __init__(self, vocab_size: Optional[int] = None, modalities: List[BaseModalityConfig] = (), pad_token_idx: int = 0,
                 eos_token_idx: int = 0, padding_side: str = "left", initializer_range: float = 0.02,
                 llm_path: str = "meta-llama/Llama-3.1-8B-Instruct", truncation: bool = False,
                 max_sequence_length: Optional[int] = None, dtype="bfloat16", **kwargs):
        self.vocab_size = vocab_size
        self.modalities = list(modalities)
        self.pad_token_idx = pad_token_idx
        self.eos_token_idx = eos_token_idx
        self.padding_side = padding_side
        self.initializer_range = initializer_range
        self.llm_path = llm_path
        self.dtype = dtype
        self.truncation = truncation
        self.max_sequence_length = max_sequence_length
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to_dict(self):
        out = {k: v for k, v in self.__dict__.items() if not k.startswith("_") and k != "modalities"}
        out["dtype"] = str(out["dtype"]).replace("torch.", "")
        out["model_type"] = self.model_type
        out["modalities"] = [m.to_dict() for m in self.modalities]
        return out

    @classmethod
    def from_dict(cls, config_dict, **kwargs):
        d = dict(config_dict)
        d.pop("model_type", None)
        d.pop("torch_dtype", None)
        mods = [AutoModality.config_from_dict(m) for m in d.pop("modalities", [])]
        d.update(kwargs)
        return cls(modalities=mods, **d)

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True, default=str)


class MultiModalModelForCausalLM(nn.Module):
    config_class = MultimodalConfig
    base_model_prefix = "model"
    supports_gradient_checkpointing = True

    def __init__(self, config: MultimodalConfig, bootstrap=False, device=None, llm_config: Optional[dict] = None):
        """`bootstrap` is accepted for signature parity (model.py:226-230): there is no hub here, so both branches build
        the LLM from `config.llm_path`'s config.json / preset with random init; local weights are loaded by
        `load_state_dict` / `from_pretrained`."""
        super().__init__()
        self.config = config
        dtype = get_torch_dtype(config.dtype)
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self._llm_cfg = LLMConfig.from_dict(llm_config if llm_config is not None else resolve_llm_config(config.llm_path))
        self.model = CausalLM(self._llm_cfg, dtype=dtype, device=device)
        self.modalities_by_type: Dict[str, BaseModality] = {}
        self.processors_by_type: Dict[str, BaseModalityProcessor] = {}
        self.modalities_with_projection = nn.ModuleList()
        for mc in config.modalities:
            modality = AutoModality.model_from_config(mc, dtype=dtype, device=device)
            processor = AutoModality.preprocessor_from_name(mc.model_type, mc)
            if mc.modality_type in self.modalities_by_type:
                raise ValueError(f"Modality type {mc.modality_type} has already been registered")
            self.modalities_by_type[mc.modality_type] = modality
            self.processors_by_type[mc.modality_type] = processor
            self.modalities_with_projection.append(modality)
        self.apply(self._init_weights)
        if config.vocab_size is not None:
            self.model.resize_token_embeddings(config.vocab_size, mean_resizing=False, std=config.initializer_range)
        self._flat: Optional[FlatParams] = None

    # ------------------------------------------------------------------ init / packing
    def _init_weights(self, module):
        """model.py:287-308: N(0, initializer_range) weights, zero biases; norms keep (1, 0)."""
        from ..nn import Embedding, Linear, Norm
        std = self.config.initializer_range
        with torch.no_grad():
            if isinstance(module, Linear):
                module.weight.normal_(mean=0.0, std=std)
                if module.bias is not None:
                    module.bias.zero_()
            elif isinstance(module, Embedding):
                module.weight.normal_(mean=0.0, std=std)
            elif isinstance(module, Norm):
                module.weight.fill_(1.0)
                if module.bias is not None:
                    module.bias.zero_()
            else:
                for n, p in module.named_parameters(recurse=False):   # class/position embeddings, patch conv
                    p.normal_(mean=0.0, std=std)

    def _component_params(self):
        """(name, parameter, component, weight_decay) for every parameter; weight_decay follows HF Trainer's grouping."""
        from ..nn import hf_decays
        owner = {}
        for mod in self.modules():
            for p in mod._parameters.values():
                if p is not None:
                    owner.setdefault(id(p), mod)
        seen, out = set(), []
        for n, p in self.model.named_parameters():
            if id(p) not in seen:
                seen.add(id(p))
                out.append(("model." + n, p, "llm", hf_decays("model." + n, owner.get(id(p)))))
        for i, m in enumerate(self.modalities_with_projection):
            for n, p in m.named_parameters():
                if id(p) in seen:
                    continue
                seen.add(id(p))
                comp = f"projector{i}" if n.startswith("projector.") else f"encoder{i}"
                full = f"modalities_with_projection.{i}.{n}"
                out.append((full, p, comp, hf_decays(full, owner.get(id(p)))))
        return out

    def pack_parameters(self) -> FlatParams:
        """Lay all parameters out in one flat buffer (fused q/k/v and gate/up operands, flat grads, flat AdamW)."""
        params = self._component_params()
        dtype = get_torch_dtype(self.config.dtype)
        dev = params[0][1].device
        self._flat = FlatParams(params, dev, dtype)
        return self._flat

    def flat_params(self) -> FlatParams:
        if self._flat is None:
            self.pack_parameters()
        return self._flat

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._flat = None            # views into the old flat buffer are gone: re-pack lazily
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = dict(state_dict)
        own = dict(self.named_parameters())
        # the reference keeps the whole CLIPModel; accept and ignore its unused towers / buffers
        unused = [k for k in sd if k not in own and any(t in k for t in ("text_model", "text_projection", "visual_projection",
                                                                         "logit_scale", "position_ids", "post_layernorm"))]
        for k in unused:
            sd.pop(k)
        if "model.lm_head.weight" in sd and self._llm_cfg.tie_word_embeddings:
            sd.pop("model.lm_head.weight")
        missing = [k for k in own if k not in sd]
        extra = [k for k in sd if k not in own]
        if strict and (missing or extra):
            raise RuntimeError(f"load_state_dict: missing={missing[:5]}... unexpected={extra[:5]}...")
        with torch.no_grad():
            for k, v in sd.items():
                if k in own:
                    own[k].copy_(v.to(device=own[k].device, dtype=own[k].dtype).reshape(own[k].shape))
        return missing, extra

    # known-ignorable keys of a reference checkpoint: the reference keeps the whole CLIPModel (text tower, projections,
    # logit_scale: SURVEY Appendix B) and HF buffers; a tied lm_head is stored under both names
    _IGNORABLE = ("text_model", "text_projection", "visual_projection", "logit_scale", "position_ids", "post_layernorm",
                  "rotary_emb.inv_freq")

    def save_pretrained(self, path: str, max_shard_size: int = 5 * 1024 ** 3, safe_serialization: bool = True):
        """HF layout (reference model.py:152-202 config + `PreTrainedModel.save_pretrained` weights): config.json with
        `MultimodalConfig.to_dict`, and the weights under the reference's parameter names as `model.safetensors`, or as
        `model-0000i-of-0000N.safetensors` + `model.safetensors.index.json` when they exceed `max_shard_size` (what an 8B
        checkpoint looks like).  Waits for in-flight device work first (the trainer's AdamW runs on a side stream)."""
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        self.config.save_pretrained(path)
        try:                                       # the LLM shape travels with the checkpoint only when llm_path cannot name it
            resolve_llm_config(self.config.llm_path)
        except (ValueError, OSError):
            with open(os.path.join(path, "llm_config.json"), "w") as f:
                json.dump(self._llm_cfg.to_dict(), f, indent=2)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        names = [(k, v) for k, v in self.named_parameters()]
        shards, cur, cur_bytes = [], [], 0
        for k, v in names:
            nb = v.numel() * v.element_size()
            if cur and cur_bytes + nb > max_shard_size:
                shards.append(cur)
                cur, cur_bytes = [], 0
            cur.append((k, v))
            cur_bytes += nb
        if cur:
            shards.append(cur)
        if len(shards) == 1:
            save_file({k: v.detach().cpu().contiguous().clone() for k, v in shards[0]}, os.path.join(path, "model.safetensors"),
                      metadata={"format": "pt"})
            return
        weight_map, total = {}, 0
        for i, sh in enumerate(shards):
            fn = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file({k: v.detach().cpu().contiguous().clone() for k, v in sh}, os.path.join(path, fn), metadata={"format": "pt"})
            for k, v in sh:
                weight_map[k] = fn
                total += v.numel() * v.element_size()
        with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2, sort_keys=True)

    def load_checkpoint_weights(self, path: str, strict: bool = True):
        """Stream `model.safetensors` or the shards listed by `model.safetensors.index.json` into the parameters, tensor by
        tensor (no whole-checkpoint host copy).  -> (missing, unexpected); raises when strict and either is non-empty
        after the known-ignorable keys (CLIP text tower etc.) are set aside."""
        from safetensors import safe_open
        idx = os.path.join(path, "model.safetensors.index.json")
        if os.path.exists(idx):
            files = sorted(set(json.load(open(idx))["weight_map"].values()))
        elif os.path.exists(os.path.join(path, "model.safetensors")):
            files = ["model.safetensors"]
        else:
            raise FileNotFoundError(f"{path}: neither model.safetensors nor model.safetensors.index.json")
        own = dict(self.named_parameters())
        seen, unexpected = set(), []
        tied = self._llm_cfg.tie_word_embeddings
        with torch.no_grad():
            for fn in files:
                with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as f:
                    for k in f.keys():
                        if k == "model.lm_head.weight" and tied:
                            continue
                        if k not in own:
                            if not any(t in k for t in self._IGNORABLE):
                                unexpected.append(k)
                            continue
                        p = own[k]
                        p.copy_(f.get_tensor(k).to(dtype=p.dtype).reshape(p.shape))
                        seen.add(k)
        missing = [k for k in own if k not in seen]
        if strict and (missing or unexpected):
            raise RuntimeError(f"{path}: missing keys {missing[:8]}{'...' if len(missing) > 8 else ''}; "
                               f"unexpected keys {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
        if missing or unexpected:
            logger.warning(f"{path}: {len(missing)} missing / {len(unexpected)} unexpected keys (left at their initial values)")
        return missing, unexpected

    @classmethod
    def from_pretrained(cls, path: str, device=None, strict: bool = True, **kwargs):
        """`kwargs` override fields of the stored config, as HF's `from_pretrained(path, truncation=..., max_sequence_length=...)`
        does for the reference (cli/train.py:131-137)."""
        raw = json.load(open(os.path.join(path, "config.json")))

        def local(name):      # a relative model directory stored beside the checkpoint (the reference resolves it against the cwd)
            if isinstance(name, str) and not os.path.isabs(name) and not os.path.isdir(name) and os.path.isdir(os.path.join(path, name)):
                return os.path.join(path, name)
            return name

        raw["llm_path"] = local(raw.get("llm_path"))
        for m in raw.get("modalities", []):
            for key in ("clip_name", "image_processor"):
                if key in m:
                    m[key] = local(m[key])
            if "expert_clip_names" in m:
                m["expert_clip_names"] = [local(x) for x in m["expert_clip_names"]]
        cfg = MultimodalConfig.from_dict(raw, **{k: v for k, v in kwargs.items() if k in ("truncation", "max_sequence_length", "dtype")})
        llm_cfg = None
        p = os.path.join(path, "llm_config.json")
        if os.path.exists(p):
            llm_cfg = json.load(open(p))
        model = cls(cfg, device=device, llm_config=llm_cfg)
        model.load_checkpoint_weights(path, strict=strict)
        return model

    # ------------------------------------------------------------------ freeze policies (model.py:310-377)
    def freeze_for_alignment(self):
        for m in self.modalities_with_projection:
            m.unfreeze_projection()
            m.freeze_modality_embedder()
        for p in self.model.parameters():
            p.requires_grad = False

    def freeze_for_lm(self):
        for m in self.modalities_with_projection:
            m.freeze_all()
        for p in self.model.parameters():
            p.requires_grad = True

    def freeze_for_end2end(self):
        for m in self.modalities_with_projection:
            m.unfreeze_projection()
            m.freeze_modality_embedder()
        for p in self.model.parameters():
            p.requires_grad = True

    def unfreeze(self):
        for m in self.modalities_with_projection:
            m.unfreeze_all()
        for p in self.model.parameters():
            p.requires_grad = True

    # ------------------------------------------------------------------ accessors (model.py:379-408)
    def processors(self) -> Dict[str, BaseModalityProcessor]:
        return self.processors_by_type

    def get_model(self):
        return self.model

    def _get_modality_by_name(self, name: str) -> BaseModality:
        if name not in self.modalities_by_type:
            raise KeyError(f"No modality registered in the model that can handle modality named: {name}")
        modality = self.modalities_by_type[name]
        if not isinstance(modality, BaseModality):
            raise TypeError(f"Registered modality {name} is not of type ModalityWithProjection")
        return modality

    def get_input_embeddings(self):
        return self.model.get_input_embeddings()

    def set_input_embeddings(self, value):
        self.model.set_input_embeddings(value)

    @property
    def device(self):
        return self.model.device

    @property
    def dtype(self):
        return self.model.dtype

    # ------------------------------------------------------------------ the hot path
    def embed_modalities_with_text(self, input_ids: torch.Tensor, processed_multimodal_inputs, stages=None):
        """model.py:410-446 as ONE gather pass: out[b,s] = projected modality row if (b,s) is in
        (batch_idx, token_range) else embedding[input_ids[b,s]]."""
        if self._flat is None:
            self.pack_parameters()
        dev = self.device
        input_ids = input_ids.to(dev)
        B, S = input_ids.shape
        projs, bis, trs = [], [], []
        pmi = processed_multimodal_inputs or {}
        for name, stack in (pmi.get("stacked") or {}).items():
            if len(stack) == 0:
                continue
            modality = self._get_modality_by_name(name)
            with trace_range(f"modality:{name}"):
                e = modality(stack, stages=stages) if stages is not None else modality(stack)
            projs.append(e.reshape(-1, e.shape[-1]))
            bis.append(pmi["batch_idx"][name].to(dev))
            trs.append(pmi["token_range"][name].to(dev))
        proj = bi = tr = None
        if projs:
            proj = projs[0] if len(projs) == 1 else torch.cat(projs, dim=0)
            bi = (bis[0] if len(bis) == 1 else torch.cat(bis)).to(torch.int64).contiguous()
            tr = (trs[0] if len(trs) == 1 else torch.cat(trs)).to(torch.int64).contiguous()
        with trace_range("embed_splice"):
            out = self.model.get_input_embeddings()(input_ids, proj, bi, tr)    # through __call__: parameter-read hooks fire
        return out.view(B, S, -1)

    def forward(self, input_ids: torch.LongTensor = None, inputs_embeds: Optional[torch.Tensor] = None,
                attention_mask: Optional[torch.Tensor] = None, position_ids: Optional[torch.LongTensor] = None,
                past_key_values=None, labels: Optional[torch.LongTensor] = None, use_cache: Optional[bool] = None,
                multimodal_inputs=None, processed_multimodal_inputs=None, return_dict: Optional[bool] = True,
                cache_position=None, **kwargs) -> Union[tuple, CausalLMOutput]:
        if self._flat is None:
            self.pack_parameters()
        if inputs_embeds is None:
            inputs_embeds = self.embed_modalities_with_text(input_ids, processed_multimodal_inputs)
        msl = self.config.max_sequence_length
        if self.config.truncation and msl is not None and inputs_embeds.shape[1] > msl:
            logger.warning(f"Truncating input to {msl} tokens.")
            inputs_embeds = inputs_embeds[:, :msl, :]
            labels = labels[:, :msl] if labels is not None else None
            kwargs.pop("loss_rows", None)                    # built from the untruncated labels
            attention_mask = attention_mask[:, :msl] if attention_mask is not None else None
            position_ids = position_ids[:, :msl] if position_ids is not None else None
        with trace_range("decoder+loss"):
            return self.model(inputs_embeds=inputs_embeds, attention_mask=attention_mask, position_ids=position_ids,
                              past_key_values=past_key_values, use_cache=use_cache, labels=labels, return_dict=return_dict,
                              **kwargs)

    def generate(self, batch: Dict[str, Any], max_new_tokens=512, temperature=0.1, do_sample=True, sync_every: int = 16,
                 **kwargs) -> torch.Tensor:
        """model.py:528-640: KV-cache decode.  Token choice = argmax(softmax(logits/T)) (one kernel) or a per-row
        multinomial draw; decode position = padded prompt length + i - 1 for every row (reference quirk :582-586);
        rows that already emitted eos keep emitting eos; returns [B, n_new] int64 on CPU without the prompt.

        The reference synchronises with the host after EVERY token (`.cpu()`, `finished.all()`, model.py:618-625,637-638).
        Here the chosen id, the eos bookkeeping and the next embedding lookup stay on the device (mm_decode_select +
        the embedding gather), so the host queues steps ahead of the GPU; it looks at `finished` only every `sync_every`
        tokens.  Steps issued past the point where every row had finished are cut off afterwards: the returned ids are
        exactly the reference's (same length, same values)."""
        if self._flat is None:
            self.pack_parameters()
        dev = self.device
        input_ids = batch["input_ids"].to(dev)
        temperature = max(temperature, 1e-6)
        B = input_ids.shape[0]
        V = self._llm_cfg.vocab_size
        eos = self.config.eos_token_idx
        if max_new_tokens <= 0:
            return torch.empty((B, 0), dtype=torch.int64)
        with torch.no_grad():
            nxt = self.embed_modalities_with_text(input_ids, batch["processed_multimodal_inputs"])
            attention_mask = batch["attention_mask"].to(dev)
            position_ids = batch["position_ids"].to(dev)
            seq_length = attention_mask.shape[1]
            cache = self.model.new_cache(B, seq_length + max_new_tokens)
            # device-resident loop state (plumbing): the growing mask is a view of one preallocated buffer
            full_mask = torch.ones((B, seq_length + max_new_tokens), dtype=attention_mask.dtype, device=dev)
            full_mask[:, :seq_length] = attention_mask
            out_ids = torch.full((B, max_new_tokens), eos, dtype=torch.int64, device=dev)
            finished = torch.zeros(B, dtype=torch.uint8, device=dev)
            next_ids = torch.empty(B, dtype=torch.int64, device=dev)
            steps = 0
            emb = self.model.get_input_embeddings()
            for i in range(max_new_tokens):
                if i > 0:
                    position_ids = torch.full((B, 1), seq_length + i - 1, dtype=torch.long, device=dev)
                    attention_mask = full_mask[:, : seq_length + i]
                out = self.model(inputs_embeds=nxt, attention_mask=attention_mask, position_ids=position_ids,
                                 past_key_values=cache, use_cache=True, logits_to_keep=1)
                logits2d = out.logits[:, -1, :]                       # [B, V] view, stride padded
                if do_sample:
                    probs = torch.softmax(logits2d.float() / temperature, dim=-1)
                    tok = torch.multinomial(probs, num_samples=1).view(B)
                else:
                    tok = K.argmax_softmax(logits2d, V, temperature)
                K.decode_select(tok, finished, eos, out_ids, i, next_ids)
                steps = i + 1
                if steps == max_new_tokens:
                    break
                if steps % max(1, sync_every) == 0 and bool(finished.all()):      # the only host sync of the loop
                    break
                nxt = emb(next_ids.view(B, 1))
            ids = out_ids[:, :steps].cpu()
            K.embed_check_pending()          # the host has just synchronised: an out-of-range input id raises here (IndexError)
        # the reference stops right after the first step at which every row has emitted eos
        done = (ids == eos).to(torch.int8).cummax(dim=1).values.bool().all(dim=0)
        if bool(done.any()):
            ids = ids[:, : int(torch.nonzero(done)[0]) + 1]
        return ids


def bootstrap(config, tokenizer, modalities_config):
    """reference model.py:643-671"""
    mc = MultimodalConfig(hidden_size=config["token_size"], vocab_size=len(tokenizer),
                          eos_token_idx=tokenizer.convert_tokens_to_ids(tokenizer.eos_token), modalities=modalities_config,
                          llm_path=config["base_llm"], truncation=config.get("truncation", False),
                          max_sequence_length=config.get("max_sequence_length", None))
    return MultiModalModelForCausalLM(mc, bootstrap=True)

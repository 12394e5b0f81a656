"""Training recipes -> (model, collator, trainer): the key mapping of the reference's `multimeditron train -c cfg.yaml`
(cli/train.py:83-157) without its CLI / wandb / DeepSpeed plumbing (out of scope).

`from_training_config(cfg, tokenizer)` consumes the dict a recipe YAML parses to (config/config_alignment.yaml:1-59 and the
cookbook/sft/*/config.yaml files have this layout):

    base_llm            hub name or local directory of the LLM (-> MultimodalConfig.llm_path; shapes from model/presets.py)
    base_model          None: bootstrap a fresh model (model.py:643-671) | a checkpoint directory: from_pretrained (train.py:131-137)
    token_size          LLM hidden size (-> MultimodalConfig.hidden_size)
    tokenizer_type      ChatTemplate name: llama | apertus | qwen3 (train.py:97)
    attachment_token    placeholder token of an attachment in the text (train.py:101, data_loader.py:26)
    loaders[]           {loader_type, modality_type, **kwargs} -> AutoModalityLoader.from_name (train.py:113-118)
    modalities[]        {model_type, ...} -> AutoModality.config_from_dict (train.py:109-111); the alternate embedder / LLM of
                        BASELINE config 5 plug in HERE (model_type: meditron_siglip, base_llm: Qwen/Qwen2-7B-Instruct)
    training_mode       ALIGNMENT | END2END | LM_ONLY | FULL (trainer.py:16-23)
    truncation, max_sequence_length, use_2d_position_ids
    training_args{}     the HF TrainingArguments keys that act on this path: learning_rate, weight_decay, max_grad_norm,
                        gradient_accumulation_steps, lr_scheduler_type, lr_scheduler_kwargs.min_lr, warmup_steps | warmup_ratio,
                        max_steps | num_train_epochs, per_device_train_batch_size, adam_beta1, adam_beta2, adam_epsilon,
                        dataloader_num_workers (-> the collator's thread pool).  Everything else there (output_dir, run_name,
                        report_to, save_*, logging_*, deepspeed, bf16, gradient_checkpointing, accelerator_config ...) belongs to
                        the reference's control plane or to ZeRO-3 memory saving and is reported back, not acted upon.

The tokenizer is an argument: `AutoTokenizer.from_pretrained(base_llm)` needs the hub (train.py:94); what the reference then does
to it (pad = eos, the chat template's special tokens and the attachment token added, train.py:95-104) happens here."""
from __future__ import annotations

import logging
import math
from typing import Any, Dict, List, NamedTuple, Optional

logger = logging.getLogger(__name__)

# HF TrainingArguments defaults (transformers 5.15) for the keys this path consumes
_TA_DEFAULTS = dict(learning_rate=5e-5, weight_decay=0.0, max_grad_norm=1.0, gradient_accumulation_steps=1,
                    lr_scheduler_type="linear", lr_scheduler_kwargs=None, warmup_steps=0, warmup_ratio=0.0, max_steps=-1,
                    num_train_epochs=3.0, per_device_train_batch_size=8, adam_beta1=0.9, adam_beta2=0.999, adam_epsilon=1e-8,
                    dataloader_num_workers=0)


class TrainingSetup(NamedTuple):
    model: Any
    collator: Any
    trainer: Any
    ignored_training_args: List[str]      # keys of training_args that belong to the reference's control plane


def prepare_tokenizer(tokenizer, chat_template, attachment_token: str):
    """reference cli/train.py:95-104"""
    tokenizer.pad_token = tokenizer.eos_token
    special = list(chat_template.special_tokens.values()) + [attachment_token]
    tokenizer.add_special_tokens({"additional_special_tokens": special})
    return tokenizer


def from_training_config(cfg: Dict[str, Any], tokenizer, train_dataset=None, device=None, dtype: Optional[str] = None,
                         llm_config: Optional[dict] = None, process_group=None) -> TrainingSetup:
    """-> TrainingSetup(model, collator, trainer, ignored_training_args).  `dtype` / `llm_config` override what the recipe implies
    (tests run tiny fp32 models on the CPU); the defaults are the reference's: bf16 (`torch.set_default_dtype(bfloat16)`,
    train.py:107) and the shape preset of `base_llm`."""
    from ..dataset.loader import AutoModalityLoader
    from ..model.data_loader import DataCollatorForMultimodal
    from ..model.modalities import AutoModality
    from ..model.model import ChatTemplate, MultimodalConfig, MultiModalModelForCausalLM
    from .trainer import TRAINING_MAPPING, MultimodalTrainer

    for key in ("base_llm", "tokenizer_type", "attachment_token", "loaders", "training_mode"):
        if key not in cfg:
            raise KeyError(f"training config: missing key '{key}' (reference cli/train.py reads it unconditionally)")
    chat_template = ChatTemplate.from_name(cfg["tokenizer_type"])
    prepare_tokenizer(tokenizer, chat_template, cfg["attachment_token"])

    modalities_config = [AutoModality.config_from_dict(dict(m)) for m in cfg.get("modalities", [])]
    loaders = {}
    for loader in cfg["loaders"]:
        kw = dict(loader)
        loader_type, modality_type = kw.pop("loader_type"), kw.pop("modality_type")
        loaders[modality_type] = AutoModalityLoader.from_name(loader_type, **kw)

    if cfg.get("base_model") is None:                                    # bootstrap (model.py:643-671)
        if "token_size" not in cfg:
            raise KeyError("training config: 'token_size' is required to bootstrap a model")
        mc = MultimodalConfig(hidden_size=cfg["token_size"], vocab_size=len(tokenizer),
                              eos_token_idx=tokenizer.convert_tokens_to_ids(tokenizer.eos_token), modalities=modalities_config,
                              llm_path=cfg["base_llm"], truncation=cfg.get("truncation", False),
                              max_sequence_length=cfg.get("max_sequence_length", None), **({"dtype": dtype} if dtype else {}))
        model = MultiModalModelForCausalLM(mc, bootstrap=True, device=device, llm_config=llm_config)
    else:                                                                # start from a checkpoint directory (train.py:131-137)
        model = MultiModalModelForCausalLM.from_pretrained(cfg["base_model"], device=device, truncation=cfg.get("truncation", False),
                                                           max_sequence_length=cfg.get("max_sequence_length", None))
    model.train()

    ta = dict(_TA_DEFAULTS)
    given = dict(cfg.get("training_args") or {})
    ignored = sorted(k for k in given if k not in _TA_DEFAULTS)
    ta.update({k: v for k, v in given.items() if k in _TA_DEFAULTS})
    collator = DataCollatorForMultimodal(tokenizer=tokenizer, modality_processors=model.processors(), modality_loaders=loaders,
                                         chat_template=chat_template, attachment_token=cfg["attachment_token"],
                                         use_2d_position_ids=cfg.get("use_2d_position_ids", False),
                                         num_threads=int(ta["dataloader_num_workers"] or 0))

    mode = cfg["training_mode"]
    if mode not in TRAINING_MAPPING:
        raise KeyError(f"training_mode {mode!r}: expected one of {sorted(TRAINING_MAPPING)}")
    import torch.distributed as dist
    world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
    accum = max(1, int(ta["gradient_accumulation_steps"]))
    max_steps = int(ta["max_steps"])
    if max_steps <= 0:                                                   # HF: epochs x ceil(batches per epoch / accumulation)
        n = len(train_dataset) if train_dataset is not None and hasattr(train_dataset, "__len__") else 0
        per_epoch = max(1, math.ceil(math.ceil(n / (int(ta["per_device_train_batch_size"]) * world)) / accum)) if n else 0
        max_steps = int(math.ceil(float(ta["num_train_epochs"]) * per_epoch))
    warmup = int(ta["warmup_steps"]) or int(math.ceil(float(ta["warmup_ratio"]) * max_steps))
    sched_kw = dict(ta["lr_scheduler_kwargs"] or {})
    trainer = MultimodalTrainer(model, training_mode=TRAINING_MAPPING[mode], learning_rate=float(ta["learning_rate"]),
                                weight_decay=float(ta["weight_decay"]), betas=(float(ta["adam_beta1"]), float(ta["adam_beta2"])),
                                eps=float(ta["adam_epsilon"]), max_grad_norm=float(ta["max_grad_norm"] or 0.0),
                                gradient_accumulation_steps=accum, max_steps=max_steps, min_lr=sched_kw.get("min_lr"),
                                warmup_steps=warmup, lr_scheduler_type=str(ta["lr_scheduler_type"]), process_group=process_group,
                                data_collator=collator, train_dataset=train_dataset,
                                per_device_train_batch_size=int(ta["per_device_train_batch_size"]))
    if ignored:
        logger.info("training_args keys left to the caller's control plane: %s", ", ".join(ignored))
    return TrainingSetup(model, collator, trainer, ignored)

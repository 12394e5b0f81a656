"""`multimeditron_amd.train.from_training_config`: the key mapping of the reference's `multimeditron train -c cfg.yaml`
(cli/train.py:83-157) driven from dicts with the keys of config/config_alignment.yaml:1-59 and of a MoE recipe
(cookbook/sft/moe/full/attn/shared/config.yaml) -- written inline here, tiny local model directories standing in for the hub names.
CPU only: construction, freezing, collator wiring, optimiser recipe; no kernel runs."""
import json
import os

import pytest
import torch

from multimeditron_amd.train import from_training_config
from multimeditron_amd.train.trainer import TrainingMode, scheduled_lr

ATTACH = "<|reserved_special_token_0|>"
TEMPLATE = ("{% for m in messages %}<|start_header_id|> {{ m['role'] }} <|end_header_id|> {{ m['content'] }} <|eot_id|>{% endfor %}"
            "{% if add_generation_prompt %}<|start_header_id|> assistant <|end_header_id|>{% endif %}")


def make_tokenizer():
    pytest.importorskip("transformers")
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    words = ["<unk>", "<|eot_id|>", "<|start_header_id|>", "<|end_header_id|>", "user", "assistant", "system", "describe", "the",
             "image", "a", "cat", "."]
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    return PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|eot_id|>", unk_token="<unk>", chat_template=TEMPLATE)


def tiny_dirs(tmp, n_clip=1):
    llm = os.path.join(str(tmp), "llm")
    os.makedirs(llm, exist_ok=True)
    json.dump(dict(model_type="llama", hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                   num_key_value_heads=2, head_dim=16, vocab_size=32, rms_norm_eps=1e-5, tie_word_embeddings=False,
                   rope_parameters={"rope_type": "default", "rope_theta": 10000.0}), open(os.path.join(llm, "config.json"), "w"))
    clips = []
    for i in range(n_clip):
        d = os.path.join(str(tmp), f"clip{i}")
        os.makedirs(d, exist_ok=True)
        json.dump({"vision_config": dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, image_size=32,
                                         patch_size=16)}, open(os.path.join(d, "config.json"), "w"))
        json.dump({"size": {"shortest_edge": 32}, "crop_size": {"height": 32, "width": 32}}, open(os.path.join(d, "preprocessor_config.json"), "w"))
        clips.append(d)
    return llm, clips


def alignment_recipe(llm, clip):
    """the keys of reference config/config_alignment.yaml:1-59"""
    return {
        "base_llm": llm, "base_model": None, "attachment_token": ATTACH, "tokenizer_type": "llama", "token_size": 64,
        "loaders": [{"loader_type": "raw-image", "modality_type": "image"}],
        "modalities": [{"model_type": "meditron_clip", "clip_name": clip, "hidden_size": 64}],
        "training_mode": "ALIGNMENT",
        "datasets": [{"packed_path": "/nonexistent"}],
        "training_args": {
            "output_dir": "models/x", "dataloader_num_workers": 16, "dataloader_prefetch_factor": 4, "remove_unused_columns": False,
            "ddp_find_unused_parameters": False, "learning_rate": 1.0e-4, "bf16": True, "per_device_train_batch_size": 4,
            "gradient_accumulation_steps": 8, "num_train_epochs": 1, "gradient_checkpointing": True,
            "gradient_checkpointing_kwargs": {"use_reentrant": True}, "save_strategy": "steps", "save_steps": 0.25, "max_grad_norm": 1.0,
            "run_name": "x", "deepspeed": "./config/deepspeed.json", "accelerator_config": {"dispatch_batches": False},
            "lr_scheduler_type": "cosine_with_min_lr", "lr_scheduler_kwargs": {"min_lr": 3.0e-5}, "report_to": "wandb", "logging_steps": 1,
            "weight_decay": 0.01},
    }


def test_alignment_recipe_builds_model_collator_trainer(tmp_path):
    llm, clips = tiny_dirs(tmp_path)
    tok = make_tokenizer()
    n0 = len(tok)
    ds = [{"text": "a cat ."}] * 1000
    setup = from_training_config(alignment_recipe(llm, clips[0]), tok, train_dataset=ds, device="cpu", dtype="float32")
    model, coll, tr = setup.model, setup.collator, setup.trainer
    # tokenizer: pad = eos, image delimiters + attachment token added (train.py:95-104); the embedding follows len(tokenizer)
    assert tok.pad_token == tok.eos_token and len(tok) == n0 + 3
    assert model.config.vocab_size == len(tok) and model.model.model.embed_tokens.weight.shape == (len(tok), 64)
    assert model.config.eos_token_idx == tok.convert_tokens_to_ids("<|eot_id|>") and model.config.hidden_size == 64
    assert model.training and list(model.modalities_by_type) == ["image"]
    # ALIGNMENT: exactly the projector trains (trainer.py:132-144, model.py:310-322)
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    assert trainable and all(".projector." in n for n in trainable)
    assert tr.training_mode == TrainingMode.ALIGNMENT
    # optimiser recipe (config_alignment.yaml:38-59)
    assert (tr.lr, tr.wd, tr.max_grad_norm, tr.accum, tr.min_lr, tr.lr_scheduler_type) == (1e-4, 0.01, 1.0, 8, 3e-5, "cosine_with_min_lr")
    assert tr.max_steps == 32 and tr.warmup == 0             # 1 epoch x ceil(ceil(1000 / 4) / 8)
    assert tr.per_device_train_batch_size == 4 and tr.data_collator is coll and tr.train_dataset is ds
    # collator wiring (train.py:153-160)
    assert coll.attachment_token == ATTACH and coll.chat_template.name == "llama" and coll.num_threads == 16
    assert set(coll.modality_loaders) == {"image"} and type(coll.modality_loaders["image"]).__name__ == "RawImageLoader"
    assert coll.modality_processors is model.processors()
    # control-plane keys are reported, not silently dropped
    assert {"deepspeed", "report_to", "output_dir", "gradient_checkpointing", "bf16"} <= set(setup.ignored_training_args)
    assert "learning_rate" not in setup.ignored_training_args


def test_moe_recipe_and_checkpoint_start(tmp_path):
    """cookbook/sft/moe/full/attn/shared/config.yaml: five experts, cross_attn fusion, FULL mode, started from `base_model`."""
    from multimeditron_amd.model.modalities.image_modality_moe import MOEImageModality
    llm, clips = tiny_dirs(tmp_path, n_clip=3)
    recipe = {
        "base_llm": llm, "base_model": None, "resume_from_checkpoint": False, "wandb_run_id": None, "attachment_token": ATTACH,
        "tokenizer_type": "llama", "token_size": 64, "truncation": True, "max_sequence_length": 48,
        "loaders": [{"loader_type": "raw-image", "modality_type": "image"}],
        "modalities": [{"model_type": "moe_meditron_clip_shared", "image_processor": clips[0], "hidden_size": 64, "expert_clip_names": clips,
                        "generalist_idx": -1, "gating_path": "stub", "fusion_method": "cross_attn", "top_k_experts": 3,
                        "cross_attn_heads": 2}],
        "training_mode": "FULL",
        "training_args": {"learning_rate": 1.0e-5, "bf16": True, "per_device_train_batch_size": 2, "gradient_accumulation_steps": 8,
                          "max_steps": 50, "warmup_ratio": 0.1, "max_grad_norm": 1.0, "lr_scheduler_type": "cosine_with_min_lr",
                          "lr_scheduler_kwargs": {"min_lr": 1.0e-6}, "weight_decay": 0.01, "deepspeed": "./config/deepspeed.json"},
    }
    tok = make_tokenizer()
    setup = from_training_config(recipe, tok, device="cpu", dtype="float32")
    m = setup.model
    mod = m.modalities_by_type["image"]
    assert isinstance(mod, MOEImageModality) and len(mod.experts) == 3 and mod.fusion_method == "cross_attn"
    assert mod.cross_attn.num_heads == 2 and mod.cross_attn.head_dim == 16
    assert all(p.requires_grad for p in m.parameters())                      # FULL
    assert m.config.truncation is True and m.config.max_sequence_length == 48
    assert (setup.trainer.max_steps, setup.trainer.warmup, setup.trainer.min_lr) == (50, 5, 1e-6)
    setup.trainer.close()
    # the same recipe started from a checkpoint directory (train.py:131-137): weights come from it, config overrides apply
    ck = tmp_path / "ckpt"
    m.save_pretrained(str(ck))
    recipe2 = dict(recipe, base_model=str(ck), max_sequence_length=40, training_mode="END2END")
    s2 = from_training_config(recipe2, make_tokenizer(), device="cpu")
    for (k1, p1), (k2, p2) in zip(m.named_parameters(), s2.model.named_parameters()):
        assert k1 == k2 and torch.equal(p1.detach(), p2.detach()), k1
    assert s2.model.config.max_sequence_length == 40
    froz = {n for n, p in s2.model.named_parameters() if not p.requires_grad}
    assert froz and all(".experts." in n for n in froz)                      # END2END: the towers freeze, projector + LLM train


def test_missing_or_bad_keys_raise(tmp_path):
    llm, clips = tiny_dirs(tmp_path)
    good = alignment_recipe(llm, clips[0])
    for key in ("base_llm", "tokenizer_type", "attachment_token", "loaders", "training_mode"):
        bad = {k: v for k, v in good.items() if k != key}
        with pytest.raises(KeyError):
            from_training_config(bad, make_tokenizer(), device="cpu", dtype="float32")
    with pytest.raises(KeyError):
        from_training_config(dict(good, training_mode="HALF"), make_tokenizer(), device="cpu", dtype="float32")
    with pytest.raises(ValueError):
        from_training_config(dict(good, tokenizer_type="gpt2"), make_tokenizer(), device="cpu", dtype="float32")
    with pytest.raises(ValueError):
        from_training_config(dict(good, training_args=dict(good["training_args"], lr_scheduler_type="polynomial")), make_tokenizer(),
                             device="cpu", dtype="float32")


@pytest.mark.parametrize("kind", ["linear", "cosine", "constant", "constant_with_warmup", "cosine_with_min_lr"])
@pytest.mark.parametrize("warmup,total", [(0, 20), (3, 20)])
def test_lr_schedules_equal_hf(kind, warmup, total):
    from transformers.optimization import get_scheduler
    base, mn = 1e-4, 3e-5
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=base)
    kw = {"scheduler_specific_kwargs": {"min_lr": mn}} if kind == "cosine_with_min_lr" else {}
    sch = get_scheduler(kind, optimizer=opt, num_warmup_steps=warmup, num_training_steps=total, **kw)
    for step in range(total + 2):
        hf = opt.param_groups[0]["lr"]
        ours = scheduled_lr(kind, step, total, base, mn, warmup)
        assert abs(hf - ours) <= 1e-12 + 1e-9 * abs(hf), (kind, step, hf, ours)
        opt.step()
        sch.step()

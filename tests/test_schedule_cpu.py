"""Optimiser recipe details against HF transformers itself (the reference trains through HF Trainer:
config/config_alignment.yaml:38-59): the `cosine_with_min_lr` schedule step by step, and which parameters get weight decay
(`Trainer.get_decay_parameter_names`)."""
import re

import pytest
import torch

from oracle import ref_cpu as R
from tests.model_utils import build_from_golden


@pytest.mark.parametrize("warmup,total", [(0, 20), (3, 20), (5, 7), (0, 1)])
def test_cosine_with_min_lr_equals_hf_scheduler(warmup, total):
    from transformers.optimization import get_scheduler
    from multimeditron_amd.train.trainer import cosine_with_min_lr
    base, mn = 1e-4, 3e-5
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=base)
    sch = get_scheduler("cosine_with_min_lr", optimizer=opt, num_warmup_steps=warmup, num_training_steps=total,
                        scheduler_specific_kwargs={"min_lr": mn})
    for step in range(total + 3):
        hf = opt.param_groups[0]["lr"]                      # the rate HF applies at optimiser step `step`
        ours = cosine_with_min_lr(step, total, base, mn, warmup)
        assert abs(hf - ours) <= 1e-12 + 1e-9 * abs(hf), (step, hf, ours)
        opt.step()
        sch.step()


def test_weight_decay_grouping_equals_hf_trainer(golden_dir, tmp_path):
    """HF decays everything except LayerNorm-module parameters and names matching bias|layernorm|rmsnorm|.norm.|_norm: so
    CLIP's 1-D class_embedding IS decayed (it was not).  Checked against HF's own function on HF modules of the
    same structure (CLIPVisionModel / LlamaForCausalLM built from the fixture's config)."""
    import torch.nn as nn
    from transformers import CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM
    from transformers.trainer_pt_utils import get_parameter_names
    meta, w, v = R.load_golden("tiny_clip_llama", golden_dir)
    m = build_from_golden(meta, w, tmp_path, "float32", device="cpu")
    flat = m.flat_params()
    ours = {sg.name for sg in flat.segments if sg.decay}
    patterns = [r"bias", r"layernorm", r"rmsnorm", r"(?:^|\.)norm(?:$|\.)", r"_norm(?:$|\.)"]
    vis_cfg = {k: meta["vision"][k] for k in ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                                                "image_size", "patch_size")}
    hf_vis = CLIPVisionModel(CLIPVisionConfig(**vis_cfg))
    llm_cfg = {k: val for k, val in meta["llm"].items() if k in ("hidden_size", "intermediate_size", "num_hidden_layers",
                                                                  "num_attention_heads", "num_key_value_heads", "head_dim", "vocab_size")}
    hf_llm = LlamaForCausalLM(LlamaConfig(**llm_cfg))
    # transformers 5.x CLIPVisionModel holds the tower's modules directly; the reference keeps a CLIPModel (`.vision_model.`)
    want = {"modalities_with_projection.0.feature_extractor.vision_model." + n.removeprefix("vision_model.")
            for n in get_parameter_names(hf_vis, [nn.LayerNorm], patterns) if "post_layernorm" not in n}
    want |= {"model." + n for n in get_parameter_names(hf_llm, [nn.LayerNorm], patterns)}
    want |= {n for n in dict(m.named_parameters()) if ".projector." in n and n.endswith("weight")}
    own = set(dict(m.named_parameters()))
    assert want <= own, sorted(want - own)[:5]
    assert ours == want, (sorted(ours - want)[:5], sorted(want - ours)[:5])
    assert "modalities_with_projection.0.feature_extractor.vision_model.embeddings.class_embedding" in ours
    assert not any(re.search(r"bias|norm", n.lower()) for n in ours)


def test_loss_rows_follow_hf_shift_and_ignore_rule():
    """functional.LossRows.host_parts: the rows the Trainer runs lm_head on are exactly the rows HF's loss keeps
    (HF:loss/loss_utils.py:52-56: labels shifted left by one, the last position and every -100 ignored)."""
    import torch
    from multimeditron_amd.functional import LossRows
    labels = torch.tensor([[5, -100, 7, 8], [-100, -100, -100, 3], [-100, -100, -100, -100]])
    idx, inv, lab = LossRows.host_parts(labels)
    # sample 0: position 0 predicts -100 (ignored), 1 -> 7, 2 -> 8, 3 -> nothing; sample 1: position 2 -> 3; sample 2: none
    assert idx.tolist() == [1, 2, 6] and lab.tolist() == [7, 8, 3]
    assert inv.tolist() == [-1, 0, 1, -1, -1, -1, 2, -1, -1, -1, -1, -1]
    logits = torch.randn(12, 11)
    full = torch.nn.functional.cross_entropy(logits, torch.nn.functional.pad(labels, (0, 1), value=-100)[..., 1:].reshape(-1),
                                             ignore_index=-100)
    rows = torch.nn.functional.cross_entropy(logits[idx.long()], lab)
    assert abs(float(full) - float(rows)) < 1e-6


def test_cu_mask_spreads_the_disabled_cus_evenly():
    """kernels.cu_mask_words ("hash"): whichever way the driver numbers CUs -- round-robin over the 8 XCDs (what tools/cumask_probe.py
    found on MI355X) or XCD by XCD -- every XCD loses the same number of CUs."""
    from multimeditron_amd.kernels import cu_mask_words
    for enabled in (248, 240, 224, 192, 128):
        words = cu_mask_words(enabled, 256, "hash")
        bits = [(words[i // 32] >> (i % 32)) & 1 for i in range(256)]
        assert sum(bits) == enabled
        per_rr = [sum(bits[i] for i in range(256) if i % 8 == x) for x in range(8)]
        per_blk = [sum(bits[32 * x:32 * x + 32]) for x in range(8)]
        assert per_rr == [enabled // 8] * 8, (enabled, per_rr)
        assert max(per_blk) - min(per_blk) <= 4, (enabled, per_blk)      # contiguous numbering: roughly even (the probe found round-robin)

"""Checkpoint interchange, CPU only: the reference's parameter names load into the product model
(strict), extra CLIP text-tower keys the reference keeps are ignored, and save_pretrained/from_pretrained round-trips."""
import json
import os

import torch

from oracle import ref_cpu as R
from tests.model_utils import build_from_golden


def test_reference_named_weights_roundtrip(golden_dir, tmp_path):
    meta, w, v = R.load_golden("tiny_clip_qwen2", golden_dir)
    m = build_from_golden(meta, w, tmp_path / "a", "float32", device="cpu")
    own = dict(m.named_parameters())
    for k, t in w.items():
        if k == "model.lm_head.weight" and meta["llm"].get("tie_word_embeddings"):
            assert m.model.lm_head.weight is m.model.model.embed_tokens.weight
            continue
        assert torch.equal(own[k].detach(), t.float()), k
    # q/k/v and gate/up live adjacently in the flat buffer (single fused GEMM operands) yet keep their own names
    a = m.model.model.layers[0].self_attn
    assert a.k_proj.weight.data_ptr() == a.q_proj.weight.data_ptr() + a.q_proj.weight.numel() * 4
    assert a._wqkv.tensor().shape == (a.q_proj.weight.shape[0] + 2 * a.k_proj.weight.shape[0], a.q_proj.weight.shape[1])
    # keys of the unused CLIP text tower (the reference keeps the whole CLIPModel) are accepted and ignored
    extra = dict(w)
    extra["modalities_with_projection.0.feature_extractor.text_model.embeddings.token_embedding.weight"] = torch.zeros(4, 4)
    extra["modalities_with_projection.0.feature_extractor.logit_scale"] = torch.zeros(())
    m.load_state_dict(extra, strict=True)
    out = tmp_path / "ckpt"
    m.save_pretrained(str(out))
    cfg = json.load(open(out / "config.json"))
    assert cfg["model_type"] == "multimodal" and cfg["modalities"][0]["model_type"] == "meditron_clip"
    from multimeditron_amd.model.model import MultiModalModelForCausalLM
    m2 = MultiModalModelForCausalLM.from_pretrained(str(out), device="cpu")
    for (k1, p1), (k2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert k1 == k2 and torch.equal(p1.detach(), p2.detach()), k1
    assert set(dict(m.named_parameters())) == {k for k in w if not (k == "model.lm_head.weight" and meta["llm"].get("tie_word_embeddings"))}

"""Shared helpers for model-level tests: build the product model from a golden fixture's meta + weights."""
import json
import os

import torch

from multimeditron_amd.model.model import MultimodalConfig, MultiModalModelForCausalLM
from multimeditron_amd.model.modalities import ImageConfig, SiglipImageConfig


def build_from_golden(meta, weights, tmpdir, dtype="float32", device="cuda"):
    clip_dir = os.path.join(str(tmpdir), "clip")
    os.makedirs(clip_dir, exist_ok=True)
    size = meta["vision"]["image_size"]
    if meta["vision"].get("kind") == "siglip":       # BASELINE config 5: the alternate embedder plugs in by modality class
        json.dump(dict(meta["vision"], model_type="siglip_vision_model"), open(os.path.join(clip_dir, "config.json"), "w"))
        mod_cfg = SiglipImageConfig(hidden_size=meta["llm"]["hidden_size"], clip_name=clip_dir)
    else:
        json.dump({"vision_config": meta["vision"]}, open(os.path.join(clip_dir, "config.json"), "w"))
        json.dump({"size": {"shortest_edge": size}, "crop_size": {"height": size, "width": size}},
                  open(os.path.join(clip_dir, "preprocessor_config.json"), "w"))
        mod_cfg = ImageConfig(hidden_size=meta["llm"]["hidden_size"], clip_name=clip_dir)
    cfg = MultimodalConfig(vocab_size=meta["vocab_size"], modalities=[mod_cfg],
                           llm_path="unused", dtype=dtype, eos_token_idx=meta["eos_token_idx"], hidden_size=meta["llm"]["hidden_size"])
    model = MultiModalModelForCausalLM(cfg, device=device, llm_config=meta["llm"])
    model.load_state_dict(weights, strict=True)
    model.pack_parameters()
    return model


def to_device(batch, device="cuda"):
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            out[k] = v.to(device)
            if k == "labels" and not v.is_cuda and str(device) != "cpu":      # what train/prefetch.py attaches to a staged batch
                from multimeditron_amd.functional import LossRows
                out[k]._mm_loss_rows = LossRows.from_host_labels(v, device)
        elif k == "processed_multimodal_inputs":
            out[k] = {"batch_idx": {t: x.to(device) for t, x in v["batch_idx"].items()},
                      "token_range": {t: x.to(device) for t, x in v["token_range"].items()},
                      "stacked": v["stacked"]}
        else:
            out[k] = v
    return out

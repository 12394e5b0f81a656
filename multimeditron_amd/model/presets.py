"""Architecture presets for the hub names the reference's recipes use (no network here: shapes only, weights are
random-init unless a local checkpoint directory is given).  Public model-card values."""
import json
import os
from typing import Any, Dict

_LLAMA3_ROPE = {"rope_type": "llama3", "rope_theta": 500000.0, "factor": 8.0, "low_freq_factor": 1.0,
                "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}

LLM_PRESETS: Dict[str, Dict[str, Any]] = {
    "meta-llama/Llama-3.1-8B-Instruct": dict(model_type="llama", hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                                              num_attention_heads=32, num_key_value_heads=8, head_dim=128, vocab_size=128256,
                                              rms_norm_eps=1e-5, tie_word_embeddings=False, rope_parameters=_LLAMA3_ROPE),
    "meta-llama/Llama-3.2-1B-Instruct": dict(model_type="llama", hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
                                              num_attention_heads=32, num_key_value_heads=8, head_dim=64, vocab_size=128256,
                                              rms_norm_eps=1e-5, tie_word_embeddings=True,
                                              rope_parameters=dict(_LLAMA3_ROPE, factor=32.0)),
    "Qwen/Qwen2-7B-Instruct": dict(model_type="qwen2", hidden_size=3584, intermediate_size=18944, num_hidden_layers=28,
                                    num_attention_heads=28, num_key_value_heads=4, head_dim=128, vocab_size=152064,
                                    rms_norm_eps=1e-6, tie_word_embeddings=False,
                                    rope_parameters={"rope_type": "default", "rope_theta": 1000000.0}),
}
for _alias, _name in (("meta-llama/Llama-3.1-8B", "meta-llama/Llama-3.1-8B-Instruct"),
                      ("meta-llama/Llama-3.2-1B", "meta-llama/Llama-3.2-1B-Instruct"),
                      ("Qwen/Qwen2-7B", "Qwen/Qwen2-7B-Instruct")):
    LLM_PRESETS[_alias] = LLM_PRESETS[_name]

VISION_PRESETS: Dict[str, Dict[str, Any]] = {
    "openai/clip-vit-large-patch14": dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                                          image_size=224, patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5),
    "openai/clip-vit-base-patch32": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                                         image_size=224, patch_size=32, hidden_act="quick_gelu", layer_norm_eps=1e-5),
}

VISION_PRESETS["google/siglip-so400m-patch14-384"] = dict(
    kind="siglip", hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16, image_size=384,
    patch_size=14, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)

CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


def _read_json(path):
    with open(path) as f:
        return json.load(f)


def resolve_llm_config(llm_path: str) -> Dict[str, Any]:
    if os.path.isdir(llm_path):
        return _read_json(os.path.join(llm_path, "config.json"))
    if llm_path in LLM_PRESETS:
        return dict(LLM_PRESETS[llm_path])
    raise ValueError(f"Unknown llm_path {llm_path!r}: give a local directory with config.json or one of {sorted(LLM_PRESETS)} "
                     "(no hub access in this build)")


def resolve_vision_config(clip_name: str) -> Dict[str, Any]:
    if os.path.isdir(clip_name):
        cfg = _read_json(os.path.join(clip_name, "config.json"))
        top_type = cfg.get("model_type", "")
        cfg = dict(cfg.get("vision_config", cfg))
        if "siglip" in str(cfg.get("model_type", top_type)):
            cfg["kind"] = "siglip"
        return cfg
    if clip_name in VISION_PRESETS:
        return dict(VISION_PRESETS[clip_name])
    raise ValueError(f"Unknown clip_name {clip_name!r}: give a local directory with config.json or one of {sorted(VISION_PRESETS)}")


def resolve_preprocessor_config(clip_name: str, image_size: int) -> Dict[str, Any]:
    base = dict(do_resize=True, size={"shortest_edge": image_size}, resample=3, do_center_crop=True,
                crop_size={"height": image_size, "width": image_size}, do_rescale=True, rescale_factor=1 / 255,
                do_normalize=True, image_mean=CLIP_MEAN, image_std=CLIP_STD, do_convert_rgb=True)
    if resolve_vision_config(clip_name).get("kind") == "siglip":      # HF SiglipImageProcessor defaults: plain resize, mean = std = 0.5
        base.update(size={"height": image_size, "width": image_size}, do_center_crop=False,
                    image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])
    p = os.path.join(clip_name, "preprocessor_config.json") if os.path.isdir(clip_name) else None
    if p and os.path.exists(p):
        base.update({k: v for k, v in _read_json(p).items() if v is not None})
    return base

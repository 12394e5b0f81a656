"""`meditron_siglip` image modality: BASELINE config 5's alternate modality embedder (SigLIP-so400m + Qwen2-7B).

The reference has no SigLIP modality (only `meditron_clip`, whose `vision_embed_dim` /
`[:, 1:, :]` are wrong for SigLIP); config 5 goes through the reference's plug-in protocol instead
(modalities/base.py:10-196: BaseModalityConfig / BaseModalityProcessor / BaseModality + AutoModality.register) and the
YAML keys `modalities[].model_type / clip_name / hidden_size` (config/config_alignment.yaml:1-14).  This class is that
plug-in on libmmhip kernels: SigLIP processor -> pixel tensor, SigLIP vision tower (no CLS token, conv bias, tanh-GELU,
post_layernorm; 72-wide heads run zero-padded on the D = 128 attention kernels) -> MLP projector over ALL tokens."""
from __future__ import annotations

from typing import Any, Dict

import torch

from ..constants import MODALITY_VALUE_KEY, NUM_EMBEDDINGS_KEY, POSITION_IDS_KEY
from ..presets import resolve_preprocessor_config, resolve_vision_config
from ..projectors.mlp import MLPProjector
from ..vision import SiglipFeatureExtractor, VisionConfig
from .base import AutoModality, BaseModality, BaseModalityConfig, BaseModalityProcessor
from .image_modality import ClipImagePreprocessor


def _siglip_vision_config(clip_name: str) -> VisionConfig:
    d = dict(resolve_vision_config(clip_name))
    d["kind"] = "siglip"
    return VisionConfig.from_dict(d)


class SiglipImageConfig(BaseModalityConfig):
    def __init__(self, hidden_size: int = 4096, clip_name: str = "google/siglip-so400m-patch14-384",
                 projection_type: str = "mlp", use_2d_position_ids: bool = False, **kwargs):
        super().__init__(modality_type="image", hidden_size=hidden_size)
        self.clip_name = clip_name
        self.projection_type = projection_type
        self.use_2d_position_ids = use_2d_position_ids


class SiglipImageProcessor(BaseModalityProcessor):
    def __init__(self, config: SiglipImageConfig):
        super().__init__(config)
        assert config.clip_name is not None, "clip_name must be specified in the config"
        vis = _siglip_vision_config(config.clip_name)
        self.image_processor = ClipImagePreprocessor(resolve_preprocessor_config(config.clip_name, vis.image_size))
        self._image_size = vis.image_size // vis.patch_size
        self._num_patches_per_entry = self._image_size ** 2

    def process(self, modality: Dict[str, Any]) -> Dict[str, Any]:
        out = modality.copy()
        out[MODALITY_VALUE_KEY] = self.image_processor(modality[MODALITY_VALUE_KEY])
        out[NUM_EMBEDDINGS_KEY] = self._num_patches_per_entry
        if self.config.use_2d_position_ids:
            g = torch.arange(self._image_size, dtype=torch.long)
            out[POSITION_IDS_KEY] = torch.stack(torch.meshgrid(g, g, indexing="ij"), dim=-1).reshape(self._num_patches_per_entry, 2)
        return out


@AutoModality.register("meditron_siglip")
class SiglipImageModality(BaseModality):
    config_class = SiglipImageConfig
    preprocessor_class = SiglipImageProcessor

    def __init__(self, config: SiglipImageConfig, dtype: torch.dtype = torch.bfloat16, device=None):
        super().__init__(config, dtype=dtype)
        self.vision_tower_name = config.clip_name
        vis = _siglip_vision_config(config.clip_name)
        self.feature_extractor = SiglipFeatureExtractor(vis, dtype=dtype, device=device)
        self.embedding_size = self.feature_extractor.vision_embed_dim
        self._num_patches_per_entry = vis.num_patches
        self.projector = MLPProjector(self.embedding_size, config.hidden_size, dtype=dtype, device=device)

    def forward(self, inputs, stages=None) -> torch.Tensor:
        """list of n pixel tensors [3,H,W] (or a stacked [n,3,H,W]) -> [n, num_patches, hidden_size]."""
        pixels = torch.stack(list(inputs), dim=0) if not torch.is_tensor(inputs) else inputs
        pixels = pixels.to(self.feature_extractor.device, non_blocking=True)
        hs = self.feature_extractor(pixels, stages=stages).last_hidden_state            # [n, P, Dv]: no CLS to drop
        out = self.projector(hs)
        if stages is not None:
            stages["vit_last_hidden"] = hs
            stages["projector_out"] = out
        return out

    def freeze_modality_embedder(self):
        for p in self.feature_extractor.parameters():
            p.requires_grad = False

    def unfreeze_modality_embedder(self):
        for p in self.feature_extractor.parameters():
            p.requires_grad = True

    def unfreeze_projection(self):
        for p in self.projector.parameters():
            p.requires_grad = True

"""`meditron_clip` image modality (reference model/modalities/image_modality.py:10-150): CLIP image processor ->
pixel tensor, CLIP vision tower -> drop CLS -> MLP projector.  The tower and projector run on libmmhip kernels."""
from __future__ import annotations

from typing import Any, Dict

import numpy as np
import torch

from ..constants import MODALITY_VALUE_KEY, NUM_EMBEDDINGS_KEY, POSITION_IDS_KEY
from ..presets import resolve_preprocessor_config, resolve_vision_config
from ..projectors.mlp import MLPProjector
from ..vision import CLIPFeatureExtractor, VisionConfig
from ... import functional as Fm
from .base import AutoModality, BaseModality, BaseModalityConfig, BaseModalityProcessor


class ImageConfig(BaseModalityConfig):
    def __init__(self, hidden_size: int = 4096, clip_name: str = "openai/clip-vit-large-patch14",
                 projection_type: str = "mlp", use_2d_position_ids: bool = False, gpu_preprocess: bool = False, **kwargs):
        """reference image_modality.py:14-44, plus `gpu_preprocess` (not in the reference): the processor then hands the decoded
        uint8 RGB image through the collator and `DevicePrefetcher(image_preprocessors=...)` resizes / crops / normalises it on the
        device (dataset/gpu_image.py: bit-identical pixels)."""
        super().__init__(modality_type="image", hidden_size=hidden_size)
        self.clip_name = clip_name
        self.projection_type = projection_type
        self.use_2d_position_ids = use_2d_position_ids
        self.gpu_preprocess = gpu_preprocess


class ClipImagePreprocessor:
    """PIL + numpy restatement of the CLIP image processor the reference obtains from
    `AutoImageProcessor.from_pretrained(clip_name)` (image_modality.py:77): RGB -> resize shortest edge (bicubic) ->
    center crop -> rescale 1/255 -> normalise."""

    def __init__(self, cfg: Dict[str, Any]):
        self.cfg = cfg

    def __call__(self, image) -> torch.Tensor:
        from PIL import Image
        c = self.cfg
        if not isinstance(image, Image.Image):
            image = Image.fromarray(np.asarray(image))
        if c.get("do_convert_rgb", True):
            image = image.convert("RGB")
        if c.get("do_resize", True):
            size = c["size"]
            w, h = image.size
            if "shortest_edge" in size:
                s = size["shortest_edge"]
                short, long = (w, h) if w <= h else (h, w)
                new_short, new_long = s, int(s * long / short)
                nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
            else:
                nh, nw = size["height"], size["width"]
            image = image.resize((nw, nh), resample=Image.Resampling(c.get("resample", 3)))
        arr = np.asarray(image)
        if c.get("do_center_crop", True):
            ch, cw = c["crop_size"]["height"], c["crop_size"]["width"]
            h, w = arr.shape[:2]
            top, left = (h - ch) // 2, (w - cw) // 2
            arr = arr[top:top + ch, left:left + cw]
        x = torch.from_numpy(np.array(arr, copy=True)).permute(2, 0, 1).to(torch.float32)
        if c.get("do_rescale", True):
            x = x * c.get("rescale_factor", 1 / 255)
        if c.get("do_normalize", True):
            mean = torch.tensor(c["image_mean"], dtype=torch.float32).view(3, 1, 1)
            std = torch.tensor(c["image_std"], dtype=torch.float32).view(3, 1, 1)
            x = (x - mean) / std
        return x.contiguous()


class ImageProcessor(BaseModalityProcessor):
    def __init__(self, config: ImageConfig):
        super().__init__(config)
        assert config.clip_name is not None, "clip_name must be specified in the config"
        vis = VisionConfig.from_dict(resolve_vision_config(config.clip_name))
        self.preprocessor_config = resolve_preprocessor_config(config.clip_name, vis.image_size)
        self.image_processor = ClipImagePreprocessor(self.preprocessor_config)
        self.gpu_preprocess = bool(getattr(config, "gpu_preprocess", False))
        self._image_size = vis.image_size // vis.patch_size
        self._num_patches_per_entry = self._image_size ** 2

    def process(self, modality: Dict[str, Any]) -> Dict[str, Any]:
        out = modality.copy()
        if self.gpu_preprocess:            # decoded RGB bytes only: the rest happens on the device (train/prefetch.py)
            from ...dataset.gpu_image import GpuClipPreprocessor
            out[MODALITY_VALUE_KEY] = GpuClipPreprocessor.to_rgb_uint8(modality[MODALITY_VALUE_KEY])
        else:
            out[MODALITY_VALUE_KEY] = self.image_processor(modality[MODALITY_VALUE_KEY])
        out[NUM_EMBEDDINGS_KEY] = self._num_patches_per_entry
        if self.config.use_2d_position_ids:
            g = torch.arange(self._image_size, dtype=torch.long)
            out[POSITION_IDS_KEY] = torch.stack(torch.meshgrid(g, g, indexing="ij"), dim=-1).reshape(self._num_patches_per_entry, 2)
        return out


@AutoModality.register("meditron_clip")
class ImageModality(BaseModality):
    config_class = ImageConfig
    preprocessor_class = ImageProcessor

    def __init__(self, config: ImageConfig, dtype: torch.dtype = torch.bfloat16, device=None):
        super().__init__(config, dtype=dtype)
        self.vision_tower_name = config.clip_name
        assert self.vision_tower_name is not None, "vision_tower_name must be specified in the config"
        vis = VisionConfig.from_dict(resolve_vision_config(config.clip_name))
        self.feature_extractor = CLIPFeatureExtractor(vis, dtype=dtype, device=device)
        self.embedding_size = self.feature_extractor.vision_embed_dim
        self._num_patches_per_entry = vis.num_patches
        self.projector = MLPProjector(self.embedding_size, config.hidden_size, dtype=dtype, device=device)

    def forward(self, inputs, stages=None) -> torch.Tensor:
        """list of n pixel tensors [3,H,W] (or a stacked [n,3,H,W]) -> [n, num_patches, hidden_size]."""
        pixels = torch.stack(list(inputs), dim=0) if not torch.is_tensor(inputs) else inputs
        pixels = pixels.to(self.feature_extractor.device, non_blocking=True)
        n = pixels.shape[0]
        hs = self.feature_extractor.vision_model(pixels, stages=stages).last_hidden_state     # [n, 1+P, Dv]
        T = hs.shape[1]
        feats = Fm.drop_cls(hs.reshape(n * T, -1), n, T)                                       # [n, P, Dv]
        out = self.projector(feats)
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

from .base import AutoModality, BaseModality, BaseModalityConfig, BaseModalityProcessor
from .image_modality import ImageConfig, ImageModality, ImageProcessor
from .siglip_modality import SiglipImageConfig, SiglipImageModality, SiglipImageProcessor
from .image_modality_moe import (CrossAttention, MOEImageConfig, MOEImageConfigPEP, MOEImageModality, MOEImageModalityPEP,
                                 MOEImageProcessor, MOEImageProcessorPEP)

__all__ = ["BaseModality", "BaseModalityConfig", "BaseModalityProcessor", "AutoModality", "ImageConfig", "ImageModality",
           "ImageProcessor", "SiglipImageConfig", "SiglipImageModality", "SiglipImageProcessor", "MOEImageConfig", "MOEImageModality",
           "MOEImageProcessor", "CrossAttention", "MOEImageConfigPEP", "MOEImageModalityPEP", "MOEImageProcessorPEP"]

from .base import AutoModality, BaseModality, BaseModalityConfig, BaseModalityProcessor
from .image_modality import ImageConfig, ImageModality, ImageProcessor

__all__ = ["BaseModality", "BaseModalityConfig", "BaseModalityProcessor", "AutoModality", "ImageConfig", "ImageModality",
           "ImageProcessor"]

"""Modality plug-in API (reference model/modalities/base.py:10-222): config / processor / model base classes and the
`AutoModality` registry.  A new modality (e.g. a SigLIP tower) plugs in by subclassing these three and decorating the
model class with `@AutoModality.register("<model_type>")` -- the same protocol as the reference; what is different is
that the model's forward runs on libmmhip kernels instead of HF modules."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict, Optional

import torch
import torch.nn as nn


class BaseModalityConfig:
    model_type: str = "base_modality"

    def __init__(self, hidden_size: int = 1024, modality_type: Optional[str] = None, **kwargs):
        self.modality_type = modality_type
        self.hidden_size = hidden_size
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to_dict(self) -> Dict[str, Any]:
        d = {k: v for k, v in self.__dict__.items() if not k.startswith("_")}
        d["model_type"] = self.model_type
        return d

    @classmethod
    def from_dict(cls, config: Dict[str, Any], **kwargs):
        cfg = {k: v for k, v in config.items() if k not in ("model_type", "modality_type", "kwargs")}
        cfg.update(kwargs)
        return cls(**cfg)


class BaseModalityProcessor(ABC):
    def __init__(self, config: BaseModalityConfig):
        self.config = config

    @abstractmethod
    def process(self, modality: Dict[str, Any]) -> Dict[str, Any]:
        """Returns the modality dict with "value" replaced by a tensor and "num_embeddings" added."""
        raise NotImplementedError

    def __call__(self, modality: Dict[str, Any]) -> Dict[str, Any]:
        return self.process(modality)


class BaseModality(ABC, nn.Module):
    config_class = BaseModalityConfig
    preprocessor_class: type = None

    def __init__(self, config: BaseModalityConfig, dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.config = config
        self.tokenizer = None
        self._dtype = dtype

    def get_config(self) -> BaseModalityConfig:
        return self.config

    @property
    def dtype(self):
        for p in self.parameters():
            return p.dtype
        return self._dtype

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    @abstractmethod
    def freeze_modality_embedder(self): ...

    @abstractmethod
    def unfreeze_modality_embedder(self): ...

    @abstractmethod
    def unfreeze_projection(self): ...

    def freeze_all(self):
        self.freeze_modality_embedder()
        for p in self.parameters():
            p.requires_grad = False

    def unfreeze_all(self):
        self.unfreeze_projection()
        self.unfreeze_modality_embedder()
        for p in self.parameters():
            p.requires_grad = True


class AutoModality:
    _registry: Dict[str, type] = {}

    def __init__(self):
        raise RuntimeError("AutoModality should not be instantiated directly. Please use the 'from_name' method.")

    @classmethod
    def register(c, name: str):
        def decorator(cls):
            if not issubclass(cls, BaseModality):
                raise ValueError(f"Class {cls.__name__} must inherit from BaseModality to be registered.")
            if name in c._registry:
                raise ValueError(f"Modality name '{name}' is already registered.")
            if getattr(cls, "preprocessor_class", None) is None:
                raise ValueError(f"Modality class '{cls.__name__}' must define a 'preprocessor_class' attribute.")
            c._registry[name] = cls
            setattr(cls.config_class, "model_type", name)
            return cls
        return decorator

    @classmethod
    def _cls(c, name):
        if name not in c._registry:
            raise ValueError(f"Modality name '{name}' is not registered. Available values are {list(c._registry.keys())}")
        return c._registry[name]

    @classmethod
    def model_from_config(c, config: BaseModalityConfig, **kwargs) -> BaseModality:
        return c._cls(config.model_type)(config, **kwargs)

    @classmethod
    def preprocessor_from_name(c, name: str, *args, **kwargs) -> BaseModalityProcessor:
        return c._cls(name).preprocessor_class(*args, **kwargs)

    @classmethod
    def config_from_dict(c, config: dict, **kwargs) -> BaseModalityConfig:
        assert "model_type" in config, "Config dictionary must contain a 'model_type' key."
        return c._cls(config["model_type"]).config_class.from_dict(config, **kwargs)

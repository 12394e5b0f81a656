"""`raw-image` loader: {"value": {"bytes": ...}} or a PIL image -> RGB PIL image (reference loader/image/bytes.py:13-51)."""
import io
from typing import Any, Dict

import PIL.Image

from .. import AutoModalityLoader, BaseModalityLoader
from ....model.constants import MODALITY_VALUE_KEY


@AutoModalityLoader.register("raw-image")
class RawImageLoader(BaseModalityLoader):
    def load(self, sample: Dict[str, Any]) -> PIL.Image.Image:
        v = sample[MODALITY_VALUE_KEY]
        if isinstance(v, PIL.Image.Image):
            return v
        return PIL.Image.open(io.BytesIO(v["bytes"])).convert("RGB")

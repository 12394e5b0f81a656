"""CLIP / SigLIP image preprocessing on the device (the optional row): the decoded uint8 RGB image goes to HBM as it
is and `mm_image_resample_h` / `mm_image_resample_v_norm` produce the fp32 pixel tensor -- bit for bit what
`ClipImagePreprocessor` (model/modalities/image_modality.py: PIL resize BICUBIC -> center crop -> rescale -> normalise, itself the
restatement of the reference's `AutoImageProcessor`, image_modality.py:77,88-93) produces on the CPU.

What stays on the host: JPEG decoding and RGB conversion (PIL), and Pillow's weight tables: `pillow_bicubic_coeffs` follows
src/libImaging/Resample.c (`precompute_coeffs` with the bicubic filter, support 2, a = -0.5, then `normalize_coeffs_8bpc`: weights
as int32 in units of 2^-22) in float64, operation for operation, so the device's integer dot products reproduce Pillow's."""
from __future__ import annotations

import ctypes
import math
from functools import lru_cache
from typing import Any, Dict, List, Sequence, Tuple

import numpy as np
import torch

from .. import _lib

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


@lru_cache(maxsize=256)
def pillow_bicubic_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """-> (bounds int32 [out, 2] = (first source index, tap count), coef int32 [out, ksize]) of Pillow's BICUBIC resize of a line
    of `in_size` pixels to `out_size` (box = the whole line)."""
    scale = float(np.float64(np.float32(in_size) - np.float32(0.0))) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    # all output positions at once; every float64 operation is the one Resample.c performs for that position, in its order
    center = 0.0 + (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)               # (int) truncates toward zero
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    t = np.arange(ksize, dtype=np.float64)[None, :]
    live = np.arange(ksize)[None, :] < xmax[:, None]
    w = np.where(live, _bicubic((t + xmin[:, None].astype(np.float64) - center[:, None] + 0.5) * ss), 0.0)
    ww = np.zeros(out_size, np.float64)
    for j in range(ksize):                               # the C loop's summation order (taps in sequence), vectorised over positions
        ww = np.where(live[:, j], ww + w[:, j], ww)
    w = np.where(live & (ww != 0.0)[:, None], w / np.where(ww != 0.0, ww, 1.0)[:, None], w)
    pre = w * float(1 << PRECISION_BITS)
    coef = np.where(live, np.where(w < 0, np.trunc(-0.5 + pre), np.trunc(0.5 + pre)), 0.0).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, coef


def resized_size(w: int, h: int, cfg: Dict[str, Any]) -> Tuple[int, int]:
    """(new width, new height) of the resize step, as ClipImagePreprocessor computes it."""
    if not cfg.get("do_resize", True):
        return w, h
    size = cfg["size"]
    if "shortest_edge" in size:
        s = size["shortest_edge"]
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = s, int(s * long / short)
        return (new_short, new_long) if w <= h else (new_long, new_short)
    return size["width"], size["height"]


class GpuClipPreprocessor:
    """Device-side twin of ClipImagePreprocessor for resample = BICUBIC (3): `pp(list of PIL images / uint8 HWC arrays) ->
    fp32 tensor [n, 3, crop_h, crop_w]` on `device`, enqueued on the current stream."""

    def __init__(self, cfg: Dict[str, Any], device=None):
        if cfg.get("do_resize", True) and int(cfg.get("resample", 3)) != 3:
            raise ValueError("GpuClipPreprocessor implements Pillow's BICUBIC resampling (resample = 3) only")
        if not torch.cuda.is_available():
            raise RuntimeError("GpuClipPreprocessor runs on the GPU (no CPU fallback: use ClipImagePreprocessor)")
        self.cfg = cfg
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._tables: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor, np.ndarray, int]] = {}
        self._mean = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in cfg.get("image_mean", [0, 0, 0])])
        self._std = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in cfg.get("image_std", [1, 1, 1])])

    def _table(self, n_in: int, n_out: int):
        key = (n_in, n_out)
        t = self._tables.get(key)
        if t is None:
            b, k = pillow_bicubic_coeffs(n_in, n_out)
            t = (torch.from_numpy(b.copy()).to(self.device), torch.from_numpy(k.copy()).to(self.device), b, k.shape[1])
            if len(self._tables) > 64:
                self._tables.clear()
            self._tables[key] = t
        return t

    @staticmethod
    def to_rgb_uint8(image) -> np.ndarray:
        from PIL import Image
        if isinstance(image, Image.Image):
            return np.asarray(image.convert("RGB"))
        a = np.asarray(image)
        if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
            a = np.asarray(Image.fromarray(a).convert("RGB"))
        return a

    def one(self, rgb: np.ndarray, out: torch.Tensor):
        """rgb: uint8 [H, W, 3] on the host; out: fp32 [3, ch, cw] on the device (written)."""
        c = self.cfg
        H, W = int(rgb.shape[0]), int(rgb.shape[1])
        nw, nh = resized_size(W, H, c)
        if c.get("do_center_crop", True):
            ch, cw = int(c["crop_size"]["height"]), int(c["crop_size"]["width"])
            top, left = (nh - ch) // 2, (nw - cw) // 2
            if top < 0 or left < 0:
                raise ValueError(f"crop {ch}x{cw} larger than the resized image {nh}x{nw}")       # numpy slicing would wrap around
        else:
            ch, cw, top, left = nh, nw, 0, 0
        assert tuple(out.shape) == (3, ch, cw) and out.dtype == torch.float32 and out.is_cuda and out.is_contiguous()
        xb, xk, _, kx = self._table(W, nw)
        yb, yk, yb_host, ky = self._table(H, nh)
        r0 = int(yb_host[top, 0])
        r1 = int(yb_host[top + ch - 1, 0] + yb_host[top + ch - 1, 1])
        src = torch.from_numpy(np.require(rgb, dtype=np.uint8, requirements=["C", "W"])).to(self.device, non_blocking=False)
        tmp = torch.empty((r1 - r0, cw, 3), dtype=torch.uint8, device=self.device)
        st = torch.cuda.current_stream().cuda_stream
        _lib.call("mm_image_resample_h", src.data_ptr(), H, W, 3 * W, r0, r1 - r0, xb.data_ptr(), xk.data_ptr(), kx, left, cw,
                  tmp.data_ptr(), st)
        _lib.call("mm_image_resample_v_norm", tmp.data_ptr(), r0, r1 - r0, cw, yb.data_ptr(), yk.data_ptr(), ky, top, ch,
                  float(np.float32(c.get("rescale_factor", 1 / 255))), int(bool(c.get("do_rescale", True))), self._mean, self._std,
                  int(bool(c.get("do_normalize", True))), out.data_ptr(), st)

    def crop_shape(self, rgb_shape) -> Tuple[int, int]:
        c = self.cfg
        if c.get("do_center_crop", True):
            return int(c["crop_size"]["height"]), int(c["crop_size"]["width"])
        nw, nh = resized_size(int(rgb_shape[1]), int(rgb_shape[0]), c)
        return nh, nw

    def __call__(self, images: Sequence[Any]) -> torch.Tensor:
        arrs: List[np.ndarray] = [self.to_rgb_uint8(im) for im in images]
        shapes = {self.crop_shape(a.shape) for a in arrs}
        if len(shapes) != 1:
            raise ValueError(f"images of one batch must preprocess to one size, got {sorted(shapes)}")
        ch, cw = next(iter(shapes))
        out = torch.empty((len(arrs), 3, ch, cw), dtype=torch.float32, device=self.device)
        for i, a in enumerate(arrs):
            self.one(a, out[i])
        return out

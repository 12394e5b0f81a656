"""Host half of the device image preprocessing (dataset/gpu_image.py), CPU only: the weight tables are Pillow's.  A numpy
two-pass resample driven by `pillow_bicubic_coeffs` must equal `PIL.Image.resize(..., BICUBIC)` bit for bit on random images of
many shapes (up- and down-scaling, one pass being the identity) and on the reference's own test images."""
import os

import numpy as np
import pytest
from PIL import Image

from multimeditron_amd.dataset.gpu_image import PRECISION_BITS, pillow_bicubic_coeffs


def _resample(img, ow, oh):
    out = img
    for axis, n_out in ((1, ow), (0, oh)):
        b, k = pillow_bicubic_coeffs(out.shape[axis], n_out)
        res = []
        for i in range(n_out):
            lo, n = int(b[i, 0]), int(b[i, 1])
            sl = out[:, lo:lo + n] if axis == 1 else out[lo:lo + n]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(sl.astype(np.int64), k[i, :n].astype(np.int64), axes=([axis], [0]))
            res.append(np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8))
        out = np.stack(res, axis=axis)
    return out


@pytest.mark.parametrize("shape", [(480, 640, 224, 298), (640, 480, 298, 224), (300, 300, 224, 224), (100, 150, 224, 336),
                                   (1000, 777, 224, 224), (224, 224, 224, 224), (50, 60, 224, 268), (225, 224, 225, 224),
                                   (97, 1300, 384, 384)])
def test_tables_reproduce_pillow_bicubic(shape):
    h, w, oh, ow = shape
    img = np.random.default_rng(h * 7 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), resample=Image.Resampling.BICUBIC))
    assert np.array_equal(_resample(img, ow, oh), ref)


def test_tables_on_the_reference_test_images(golden_dir):
    for name in ("cat.jpg", "EPFL_campus_2017.jpg"):
        im = Image.open(os.path.join(golden_dir, "mock_dataset", name)).convert("RGB")
        w, h = im.size
        short, long = (w, h) if w <= h else (h, w)
        nw, nh = (224, int(224 * long / short)) if w <= h else (int(224 * long / short), 224)
        ref = np.asarray(im.resize((nw, nh), resample=Image.Resampling.BICUBIC))
        assert np.array_equal(_resample(np.asarray(im), nw, nh), ref), name

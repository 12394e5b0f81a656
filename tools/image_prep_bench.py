#!/usr/bin/env python3
"""Image preprocessing (resize shortest edge 224 bicubic -> center crop -> rescale -> normalise) of already decoded images: the CPU
path (ClipImagePreprocessor = the reference's AutoImageProcessor, PIL) against the device path (GpuClipPreprocessor)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from multimeditron_amd.dataset.gpu_image import GpuClipPreprocessor
from multimeditron_amd.model.modalities.image_modality import ClipImagePreprocessor
from multimeditron_amd.model.presets import resolve_preprocessor_config

cfg = resolve_preprocessor_config("openai/clip-vit-large-patch14", 224)
cpu, gpu = ClipImagePreprocessor(cfg), GpuClipPreprocessor(cfg)
rng = np.random.default_rng(0)
for (h, w) in ((480, 640), (1024, 1024), (2048, 1536)):
    imgs = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for _ in range(8)]
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    ref = torch.stack([cpu(im) for im in imgs])
    t_cpu = (time.perf_counter() - t0) / len(imgs) * 1e3
    gpu(imgs[:2]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = gpu(imgs)
    torch.cuda.synchronize()
    t_gpu = (time.perf_counter() - t0) / len(imgs) * 1e3
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    arrs = [gpu.to_rgb_uint8(im) for im in imgs]
    dev = [torch.from_numpy(np.array(a)).cuda() for a in arrs]
    torch.cuda.synchronize()
    print(f"{h}x{w}: CPU (PIL, 1 thread) {t_cpu:.2f} ms/image; device path incl. host tables + H2D of the raw image {t_gpu:.2f} ms/image; "
          f"identical: {bool(torch.equal(got.cpu(), ref))}", flush=True)

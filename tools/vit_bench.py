#!/usr/bin/env python3
"""The headline workload's image modality ALONE (CLIP ViT-L/14 + projector, n images, trainable): forward and forward + backward
wall time with nothing else on the chip -- what the step's two overlap windows would cost un-overlapped.
   python tools/vit_bench.py [n_images]            (under rocprofv3 --kernel-trace --stats for the per-kernel table)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd.model.modalities import ImageConfig, ImageModality
from multimeditron_amd.nn import FlatParams

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
m = ImageModality(ImageConfig(hidden_size=4096, clip_name="openai/clip-vit-large-patch14"), dtype=torch.bfloat16, device="cuda")
flat = FlatParams([(k, p, "projector" if k.startswith("projector") else "encoder") for k, p in m.named_parameters()], "cuda", torch.bfloat16)
for p in m.parameters():
    p.requires_grad_(True)
flat.attach_grads(fresh=True)
px = torch.randn(n, 3, 224, 224, device="cuda")
for mode in ("fwd", "fwd+bwd"):
    for it in range(8):
        if it == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if mode == "fwd":
            with torch.no_grad():
                y = m(px)
        else:
            y = m(px)
            y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    print(f"ViT-L/14 + projector, {n} images, {mode}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)

#!/usr/bin/env python3
"""Stand-alone rate of the split-master AdamW update (mm_adamw_step_split: 26 B per parameter): python tools/adamw_bench.py [n_millions]
(the step's update moves 217 GB in 42-44 ms = 5.0-5.2 TB/s while the ViT forward runs beside it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd import kernels as K
from multimeditron_amd._lib import lib

n = (int(sys.argv[1]) if len(sys.argv) > 1 else 512) * (1 << 20)
dev = "cuda"
p = torch.randn(n, device=dev, dtype=torch.bfloat16)
g = torch.randn(n, device=dev, dtype=torch.bfloat16) * 0.01
lo = torch.zeros(n, device=dev, dtype=torch.int16)
m = torch.zeros(n, device=dev, dtype=torch.float32)
v = torch.zeros(n, device=dev, dtype=torch.float32)
clip = torch.tensor([1.0, 1.0], device=dev, dtype=torch.float32)
for blocks in (2048, 16384, 0, 1048576, 2048, 0):
    assert lib().mm_set_option(b"adamw_blocks", blocks) == 0
    for _ in range(2):
        K.adamw_step_split(p, g, lo, m, v, 1e-4, 0.9, 0.999, 1e-8, 0.01, 1, clip)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(5):
        K.adamw_step_split(p, g, lo, m, v, 1e-4, 0.9, 0.999, 1e-8, 0.01, 2 + i, clip)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"adamw_blocks={blocks:7d} (0 = default 262144): {ms:7.3f} ms for {n / 2**20:.0f} Mi parameters = {26.0 * n / ms / 1e9:6.2f} TB/s", flush=True)
lib().mm_set_option(b"adamw_blocks", 0)

# the gradient-norm sweep (mm_gradnorm_partial: one read of the gradients)
part = torch.empty(1024, device=dev, dtype=torch.float32)
for _ in range(2):
    K.gradnorm_partial(g, part)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    K.gradnorm_partial(g, part)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"gradnorm_partial, 1024 workgroups: {ms:7.3f} ms for {n / 2**20:.0f} Mi bf16 gradients = {2.0 * n / ms / 1e9:6.2f} TB/s; sum {float(part.sum()):.6e} (torch {float((g.float() ** 2).sum()):.6e})")

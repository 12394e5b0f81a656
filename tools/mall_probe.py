#!/usr/bin/env python3
"""Does a weight matrix that was just read (by any kernel) serve the decode GEMV faster than one coming from HBM?  For each shape:
GEMV on cold weights (16 other layers' weights were streamed in between) vs GEMV right after a pass that read the same weights.
Feasibility probe for a weight prefetcher running ahead of the decode chain in the 256 MB Infinity Cache.  python tools/mall_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd import kernels as K

B, L = 4, 12
H, I = 4096, 14336
bf = torch.bfloat16
dev = "cuda"
torch.manual_seed(0)
shapes = {"o_proj 34 MB": (H, H), "q|k|v 50 MB": (6144, H), "down 117 MB": (H, I), "gate|up 235 MB": (2 * I, H)}
for name, (N, Kd) in shapes.items():
    ws = [torch.randn(N, Kd, device=dev, dtype=bf) * 0.02 for _ in range(L)]
    x = torch.randn(B, Kd, device=dev, dtype=bf)
    res = {}
    for mode in ("cold", "touched"):
        ts = []
        for rep in range(3):
            for i in range(L):
                # a pass that reads every byte of a weight matrix right before the GEMV (any reader will do for the probe): the same
                # matrix ("touched") or another one of the same size ("cold": same launch pattern, nothing of W_i cached)
                ws[i if mode == "touched" else (i + L // 2) % L].view(torch.int16).sum()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                K.linear_fwd(x, ws[i])
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    mb = N * Kd * 2 / 1e6
    print(f"{name:16s} cold {res['cold']:6.1f} us ({mb / res['cold'] / 1e6 * 1e6 / 1e6:.2f} TB/s)   just read {res['touched']:6.1f} us ({mb / res['touched']:.2f} MB/us)", flush=True)
    del ws
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Small HBM-bound kernels in isolation at the shapes of the 8B step (decoder rows 8192 x 4096, ViT-L rows 1028 x 1024/4096):
time per launch and the bytes each must move, so that an in-step duration (rocprofv3, other streams sharing the chip) can be
told from the kernel's own speed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd import kernels as K


def timeit(name, fn, nbytes, it=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / it * 1e3
    print(f"{name:44s} {us:8.1f} us   {nbytes / 1e6:8.1f} MB   {nbytes / us / 1e6:7.2f} TB/s", flush=True)


def main():
    dev = "cuda"
    bf = torch.bfloat16
    for M, H, tag in ((8192, 4096, "decoder"), (1028, 1024, "vit")):
        x = torch.randn(M, H, device=dev).to(bf)
        dy = torch.randn(M, H, device=dev).to(bf)
        w = torch.randn(H, device=dev).to(bf)
        b = torch.randn(H, device=dev).to(bf)
        if tag == "decoder":
            y, rstd = K.rmsnorm_fwd(x, w, 1e-5)
            timeit(f"rmsnorm_fwd {M}x{H}", lambda: K.rmsnorm_fwd(x, w, 1e-5), 2 * M * H * 2)
            timeit(f"rmsnorm_bwd {M}x{H} (+dres)", lambda: K.rmsnorm_bwd(dy, x, w, rstd, dy), 4 * M * H * 2)
            dx, dwp = K.rmsnorm_bwd(dy, x, w, rstd, dy)
        else:
            y, mean, rstd = K.layernorm_fwd(x, w, b, 1e-5)
            timeit(f"layernorm_fwd {M}x{H}", lambda: K.layernorm_fwd(x, w, b, 1e-5), 2 * M * H * 2)
            timeit(f"layernorm_bwd {M}x{H} (+dres)", lambda: K.layernorm_bwd(dy, x, w, mean, rstd, dy), 4 * M * H * 2)
            dx, dwp, dbp = K.layernorm_bwd(dy, x, w, mean, rstd, dy)
        out = torch.zeros(H, device=dev, dtype=bf)
        timeit(f"reduce_partials {dwp.shape[0]}x{H}", lambda: K.reduce_partials(dwp, out, True), dwp.numel() * 4)
    for M, N in ((1028, 1024), (1028, 4096), (1028, 3072)):
        x = torch.randn(M, N, device=dev).to(bf)
        out = torch.zeros(N, device=dev, dtype=bf)
        timeit(f"colsum {M}x{N}", lambda: K.colsum(x, out, False), M * N * 2)
    x = torch.randn(1028, 4096, device=dev).to(bf)
    timeit("gelu_fwd(quick) 1028x4096", lambda: K.gelu_fwd(x, 1), 2 * x.numel() * 2)
    timeit("gelu_bwd(quick) 1028x4096", lambda: K.gelu_bwd(x, x, 1), 3 * x.numel() * 2)
    x = torch.empty(1 << 28, device=dev, dtype=bf)
    timeit("torch zero_ 512 MB", lambda: x.zero_(), x.numel() * 2, it=10)


if __name__ == "__main__":
    main()

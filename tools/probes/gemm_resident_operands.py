import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimeditron_amd import kernels as K
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
for M, N, Kd in ((8192, 4096, 4096), (8192, 4096, 14336), (8192, 28672, 4096), (8192, 4096, 28672)):
    a, b = r(M, Kd), r(N, Kd)
    a0, b0 = r(256, Kd), r(256, Kd)
    # operands whose 256-row panels repeat: every tile reads the same 256 x K panel of A and of B (L2-resident after the first K pass)
    ar = a0.repeat(1, 1).unsqueeze(0).expand(M // 256, 256, Kd)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for name, (x, y) in {"streamed": (a, b)}.items():
        for _ in range(3): K.gemm(0, x, y, M, N, Kd, out=c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): K.gemm(0, x, y, M, N, Kd, out=c)
        e1.record(); torch.cuda.synchronize()
        res[name] = 2.0 * M * N * Kd / (e0.elapsed_time(e1) / 5) / 1e9
    # row stride 0: all rows of A (and of B) are one row of K elements: 28 KB + 28 KB of operand data in all
    x, y = a[:1].expand(M, Kd), b[:1].expand(N, Kd)
    for _ in range(3): K.gemm(0, x, y, M, N, Kd, out=c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): K.gemm(0, x, y, M, N, Kd, out=c)
    e1.record(); torch.cuda.synchronize()
    res["row stride 0 (L2-resident)"] = 2.0 * M * N * Kd / (e0.elapsed_time(e1) / 5) / 1e9
    print(f"NT M={M} N={N} K={Kd}: " + "  ".join(f"{k}: {v:6.0f} TF/s" for k, v in res.items()), flush=True)

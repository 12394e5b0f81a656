#!/usr/bin/env python3
"""The step's GEMM shapes on libmmhip's 256x256 LDS-DMA kernel against the vendor library behind torch.matmul (hipBLASLt / rocBLAS),
same process, interleaved, uniform random [-1, 1) operands.  A comparison only: the product never calls the library.
   python tools/gemm_vs_library.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd import kernels as K

T = 8192
SHAPES = {"NT": [(T, 6144, 4096), (T, 4096, 4096), (T, 28672, 4096), (T, 4096, 14336)],
          "NN": [(T, 4096, 6144), (T, 4096, 4096), (T, 4096, 28672), (T, 14336, 4096)],
          "TN": [(6144, 4096, T), (4096, 4096, T), (28672, 4096, T), (4096, 14336, T)]}
LAY = {"NT": 0, "NN": 1, "TN": 2}
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
tot = {"mm": 0.0, "lib": 0.0}
for lay, shapes in SHAPES.items():
    for (M, N, Kd) in shapes:
        if lay == "NT":
            a, b = r(M, Kd), r(N, Kd)
            lib = lambda: torch.matmul(a, b.t(), out=c2)
        elif lay == "NN":
            a, b = r(M, Kd), r(Kd, N)
            lib = lambda: torch.matmul(a, b, out=c2)
        else:
            a, b = r(Kd, M), r(Kd, N)
            lib = lambda: torch.matmul(a.t(), b, out=c2)
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        c2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        mm = lambda: K.gemm(LAY[lay], a, b, M, N, Kd, out=c)
        res = {"mm": [], "lib": []}
        for rnd in range(5):
            for name, fn in (("mm", mm), ("lib", lib)):
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[name].append(e0.elapsed_time(e1) / 3)
        fl = 2.0 * M * N * Kd
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        same = float((c.float() - c2.float()).abs().max())
        for k in med:
            tot[k] += med[k]
        print(f"{lay} M={M:6d} N={N:6d} K={Kd:6d}   libmmhip {fl / med['mm'] / 1e9:7.1f} TF/s   library {fl / med['lib'] / 1e9:7.1f} TF/s   max |diff| {same:.3g}", flush=True)
print(f"sum over the 12 shapes: libmmhip {tot['mm']:.2f} ms, library {tot['lib']:.2f} ms")

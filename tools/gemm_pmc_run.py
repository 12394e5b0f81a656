#!/usr/bin/env python3
"""Workload for tools/gemm_pmc.sh: a few launches of the 8-wave and the 4-wave 256x256 GEMM kernels on step shapes (NT, NN)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd import kernels as K
from multimeditron_amd._lib import lib

g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
modes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1".split(","))]
for lay, M, N, Kd in ((0, 8192, 4096, 4096), (0, 8192, 4096, 14336)):
    a = r(M, Kd)
    b = r(N, Kd) if lay == 0 else r(Kd, N)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for mode in modes:
        assert lib().mm_set_option(b"gemm_w4", mode) == 0
        for _ in range(6):
            K.gemm(lay, a, b, M, N, Kd, out=c)
torch.cuda.synchronize()

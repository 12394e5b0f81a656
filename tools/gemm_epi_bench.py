#!/usr/bin/env python3
"""Epilogue A/B of the 256x256 LDS-DMA GEMM on the step's shapes that carry one (run on the GPU box):
   python tools/gemm_epi_bench.py
Interleaved in one process: mm_set_option("gemm_epi_pipe", 0 | 1) = the serial per-element epilogue vs the pipelined, branch-free
one (csrc/mm_gemm.hip gemm_epilogue_plain_pipe / gemm_epilogue_swiglu_bwd_pipe).  Also checks the two give identical bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd import kernels as K
from multimeditron_amd._lib import lib

T, H, I = 8192, 4096, 14336
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)


def opt(v):
    assert lib().mm_set_option(b"gemm_epi_pipe", v) == 0


def timed(fn, it=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


cases = []
x, w_o, res = r(T, H), r(H, H), r(T, H)
cases.append(("NT o_proj + residual      M=8192 N=4096  K=4096 ", 2.0 * T * H * H, lambda: K.linear_fwd(x, w_o, residual=res)))
cases.append(("NT o_proj plain           M=8192 N=4096  K=4096 ", 2.0 * T * H * H, lambda: K.linear_fwd(x, w_o)))
act, w_d = r(T, I), r(H, I)
cases.append(("NT down_proj + residual   M=8192 N=4096  K=14336", 2.0 * T * H * I, lambda: K.linear_fwd(act, w_d, residual=res)))
dy, gu = r(T, H), r(T, 2 * I)
cases.append(("NN down dgrad + SwiGLU'   M=8192 N=14336 K=4096 ", 2.0 * T * H * I, lambda: K.gemm_swiglu_bwd(dy, w_d, gu, I)))
cases.append(("NN down dgrad plain       M=8192 N=14336 K=4096 ", 2.0 * T * H * I, lambda: K.linear_dgrad(dy, w_d)))
dw = torch.zeros(H, H, device="cuda", dtype=torch.bfloat16)
cases.append(("TN o_proj wgrad accumulate M=4096 N=4096 K=8192 ", 2.0 * T * H * H, lambda: K.linear_wgrad(dy, x, dw, True)))
cases.append(("TN o_proj wgrad overwrite  M=4096 N=4096 K=8192 ", 2.0 * T * H * H, lambda: K.linear_wgrad(dy, x, dw, False)))

# bit-identity of the two epilogues
for name, fl, fn in cases[:5]:
    opt(0); a = fn(); a = a.clone() if torch.is_tensor(a) else a
    opt(1); b = fn()
    same = torch.equal(a, b)
    print(f"identical bits: {same}  {name}", flush=True)
    assert same, name

tot = {0: 0.0, 1: 0.0}
for name, fl, fn in cases:
    res_ms = {0: [], 1: []}
    for rnd in range(6):
        for mode in (0, 1):
            opt(mode)
            res_ms[mode].append(timed(fn))
    med = {m: sorted(v)[len(v) // 2] for m, v in res_ms.items()}
    for m in med:
        tot[m] += med[m]
    print(f"{name}  serial {med[0] * 1e3:7.1f} us {fl / med[0] / 1e9:7.1f} TF/s   pipelined {med[1] * 1e3:7.1f} us {fl / med[1] / 1e9:7.1f} TF/s", flush=True)
opt(1)
print("TOTAL ms", tot)

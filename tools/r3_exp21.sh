#!/bin/bash
# round-3 experiment 21: the plain 256x256 GEMM on 4 waves of 128x128 (one wave per SIMD) vs 8 waves of 128x64
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp21
mkdir -p $O
cd $R
timeout -k 10 300 python3 - > $O/check.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from multimeditron_amd import kernels as K
from multimeditron_amd._lib import lib
torch.manual_seed(0)
for lay, (M, N, Kd) in ((0, (8192, 4096, 4096)), (1, (8192, 4096, 6144)), (2, (6144, 4096, 8192)), (0, (1000, 520, 328)), (1, (777, 264, 200))):
    a = (torch.rand((M, Kd) if lay != 2 else (Kd, M), device="cuda") * 2 - 1).to(torch.bfloat16)
    b = (torch.rand((N, Kd) if lay == 0 else (Kd, N), device="cuda") * 2 - 1).to(torch.bfloat16)
    outs = []
    for wv in (8, 4):
        lib().mm_set_option(b"gemm_waves", wv)
        lib().mm_set_option(b"gemm_kernel", 3)
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        K.gemm(lay, a, b, M, N, Kd, out=c)
        torch.cuda.synchronize()
        outs.append(c)
    lib().mm_set_option(b"gemm_waves", 8)
    lib().mm_set_option(b"gemm_kernel", 0)
    print(lay, M, N, Kd, "bit-identical" if torch.equal(outs[0], outs[1]) else f"DIFF {float((outs[0].float() - outs[1].float()).abs().max())}", flush=True)
PY
cat $O/check.txt | grep -v amdgpu
timeout -k 10 600 python3 tools/gemm_bench.py --ab=gemm_waves:8:4 > $O/ab.txt 2>&1; grep -v amdgpu $O/ab.txt | tail -18

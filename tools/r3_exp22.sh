#!/bin/bash
# round-3 experiment 22: plain NT / NN GEMMs of the decoder through the vendor library (MM_GEMM_LIB=1)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp22
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm_lib" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -12 $O/pytest.txt | cut -c1-250
timeout -k 10 900 python3 tools/step_ab.py --rounds 4 --steps 5 --warmup 2 "MM_GEMM_LIB=0" "MM_GEMM_LIB=1" > $O/step_ab.txt 2>&1
tail -3 $O/step_ab.txt

#!/bin/bash
# round-3 experiment 2: pipelined GEMM epilogues (correctness, micro A/B, step A/B) + the new cross-attention kernels' tests
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp2
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_xattn_gpu.py tests/test_moe_modality_gpu.py -q -m gpu -x > $O/pytest_xattn.txt 2>&1; echo "xattn tests rc=$?"; tail -15 $O/pytest_xattn.txt
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_kernels_random_gpu.py -q -m gpu > $O/pytest_kernels.txt 2>&1; echo "kernel tests rc=$?"; tail -5 $O/pytest_kernels.txt
timeout -k 10 300 python3 tools/gemm_epi_bench.py > $O/epi_bench.txt 2>&1; echo "epi bench rc=$?"; tail -12 $O/epi_bench.txt
timeout -k 10 500 python3 tools/step_ab.py --rounds 3 --steps 8 "opt:gemm_epi_pipe=0" "opt:gemm_epi_pipe=1" > $O/step_ab.txt 2>&1
tail -4 $O/step_ab.txt

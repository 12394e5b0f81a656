#!/bin/bash
# round-3 experiment 9: gradient-norm sum of squares inside the wgrad GEMM epilogue (EK = 5) vs the sweep
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp9
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_trainer_gpu.py -q -m gpu -k "sumsq or gradnorm or gemm" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -6 $O/pytest.txt
timeout -k 10 900 python3 tools/step_ab.py --rounds 3 --steps 8 "MM_FUSED_NORM=0" "MM_FUSED_NORM=1" "MM_FUSED_NORM=1,MM_SUMSQ_EPILOGUE=0" > $O/step_ab.txt 2>&1
tail -4 $O/step_ab.txt

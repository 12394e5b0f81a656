#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp14
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_pipeline_gpu.py tests/test_kernels_random_gpu.py tests/test_abi.py -q -m gpu > $O/pytest2.txt 2>&1; echo "tests rc=$?"; tail -12 $O/pytest2.txt | cut -c1-250

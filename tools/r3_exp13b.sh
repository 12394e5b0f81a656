#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp13
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_moe_modality_gpu.py tests/test_trainer_gpu.py tests/test_dp_gpu.py tests/test_fullsize_gpu.py -q -m gpu > $O/pytest2.txt 2>&1; echo "tests rc=$?"; tail -8 $O/pytest2.txt | cut -c1-250

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp19
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py -q -m gpu -k "fused_decode_chain" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -30 $O/pytest.txt | cut -c1-250

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp13
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_moe_modality_gpu.py -q -m gpu -k "full_recipe or trainable" > $O/pytest3.txt 2>&1; echo "tests rc=$?"; tail -30 $O/pytest3.txt | cut -c1-250

#!/bin/bash
# round-3 experiment 13: trainable MoE towers replayed from captured forward / backward graphs
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp13
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_moe_modality_gpu.py -q -m gpu > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -15 $O/pytest.txt | cut -c1-250
timeout -k 10 300 python3 tools/moe_bench.py 4 4 > $O/moe_bench.txt 2>&1; tail -30 $O/moe_bench.txt | cut -c1-200

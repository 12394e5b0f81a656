#!/bin/bash
# round-3 experiment 8: frozen MoE towers as a captured graph; other workloads of the final code
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp8
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_moe_modality_gpu.py tests/test_xattn_gpu.py -q -m gpu > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -12 $O/pytest.txt
timeout -k 10 400 python3 tools/moe_bench.py 4 4 > $O/moe_bench.txt 2>&1; cat $O/moe_bench.txt | grep -v amdgpu.ids

#!/bin/bash
# round-3 experiment 7: AdamW with the master held as parameter + remainder; fused RoPE (fixed) re-measured
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp7
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_trainer_gpu.py tests/test_model_gpu.py -q -m gpu > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -6 $O/pytest.txt
timeout -k 10 900 python3 tools/step_ab.py --rounds 3 --steps 8 "MM_ADAMW_SPLIT=0,MM_FUSED_ROPE=0" "MM_ADAMW_SPLIT=1,MM_FUSED_ROPE=0" "MM_ADAMW_SPLIT=1,MM_FUSED_ROPE=1" > $O/step_ab.txt 2>&1
tail -4 $O/step_ab.txt

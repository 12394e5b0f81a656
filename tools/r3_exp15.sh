#!/bin/bash
# round-3 experiment 15: last decoder layer's o_proj + MLP on the labelled rows only
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp15
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_trainer_gpu.py tests/test_dp_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -q -m gpu > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -6 $O/pytest.txt | cut -c1-250
timeout -k 10 900 python3 tools/step_ab.py --rounds 5 --steps 5 --warmup 2 "MM_LOSS_ROWS=0" "MM_LOSS_ROWS=1" > $O/step_ab.txt 2>&1
tail -3 $O/step_ab.txt

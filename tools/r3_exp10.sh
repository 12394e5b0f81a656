#!/bin/bash
# round-3 experiment 10: final norm + lm_head + loss on the labelled rows only (MM_LOSS_ROWS) vs every row
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp10
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_trainer_gpu.py tests/test_dp_gpu.py tests/test_prefetch_gpu.py -q -m gpu -k "rows or sumsq or training_steps or dp or prefetch or resume" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -6 $O/pytest.txt
timeout -k 10 900 python3 tools/step_ab.py --rounds 3 --steps 8 "MM_LOSS_ROWS=0" "MM_LOSS_ROWS=1" > $O/step_ab.txt 2>&1
tail -4 $O/step_ab.txt

#!/bin/bash
# round-3 experiment 23: LayerNorm dw / db partial reductions in one launch (49 fewer launches in the ViT backward chain)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp23
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_trainer_gpu.py tests/test_kernels_random_gpu.py -q -m gpu -k "norm or vit or f32_grads or bf16_grads or training_steps or layernorm" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.txt | cut -c1-200
bash tools/r3_trace.sh r3_exp23/trace > $O/trace.log 2>&1; grep -n "window of stream" -A2 $O/trace/stream_time.txt | cut -c1-260; grep -n "AFTER the window" $O/trace/stream_time.txt | cut -c1-200; head -1 $O/trace/stream_time.txt

#!/bin/bash
# round-3 experiment 4: fused RoPE GEMM + fused decode-step kernels: correctness, step A/B, decode A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp4
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_headline_geometry_gpu.py tests/test_trainer_gpu.py -q -m gpu > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -12 $O/pytest.txt
MM_DECODE_FUSED=0 timeout -k 10 300 python3 tools/decode_bench.py > $O/decode_unfused.txt 2>&1; tail -1 $O/decode_unfused.txt
MM_DECODE_FUSED=1 timeout -k 10 300 python3 tools/decode_bench.py > $O/decode_fused.txt 2>&1; tail -1 $O/decode_fused.txt
timeout -k 10 500 python3 tools/step_ab.py --rounds 3 --steps 8 "MM_FUSED_ROPE=0" "MM_FUSED_ROPE=1" > $O/step_ab.txt 2>&1
tail -3 $O/step_ab.txt

#!/bin/bash
# round-3 experiment 14: ring-buffered weight-streaming kernel with the RMSNorm as its prologue (gemv_stream_kernel)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp14
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_pipeline_gpu.py -q -m gpu -k "decode or skinny or generate or greedy or cache or attn" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -12 $O/pytest.txt | cut -c1-250
timeout -k 10 300 python3 tools/gemv_bench.py 4 16 > $O/gemv.txt 2>&1; tail -20 $O/gemv.txt | cut -c1-200
timeout -k 10 300 python3 tools/decode_bench.py > $O/decode.txt 2>&1; tail -3 $O/decode.txt | cut -c1-250

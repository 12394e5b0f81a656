#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp20
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -m gpu -k "decode or skinny or fused_decode_chain or generate" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.txt | cut -c1-250
timeout -k 10 300 python3 tools/gemv_bench.py 4 16 > $O/gemv.txt 2>&1; sed -n 2,8p $O/gemv.txt | cut -c1-200
timeout -k 10 300 python3 tools/decode_bench.py > $O/decode.txt 2>&1; tail -1 $O/decode.txt | cut -c1-250

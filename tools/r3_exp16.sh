#!/bin/bash
# round-3 experiment 16: decode attention slices with the scores on MFMA
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp16
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_pipeline_gpu.py -q -m gpu -k "decode or generate or greedy or cache or attention" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -12 $O/pytest.txt | cut -c1-250
bash tools/r3_gemv_prof.sh r3_exp16/prof > $O/prof.log 2>&1; grep -E "attention|attn_decode" $O/prof.log | cut -c1-200
timeout -k 10 300 python3 tools/decode_bench.py > $O/decode.txt 2>&1; tail -2 $O/decode.txt | cut -c1-250

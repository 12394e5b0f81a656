#!/bin/bash
# round-3 experiment 17: argmax(softmax) of the decode step over many workgroups
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp17
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_kernels_random_gpu.py tests/test_model_gpu.py tests/test_abi.py -q -m gpu -k "argmax or cross_entropy or generate or greedy or decode or abi or ids" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -8 $O/pytest.txt | cut -c1-250
timeout -k 10 300 python3 tools/decode_bench.py > $O/decode.txt 2>&1; tail -2 $O/decode.txt | cut -c1-250
MM_ARGMAX_SPLIT=0 timeout -k 10 300 python3 tools/decode_bench.py > $O/decode_one.txt 2>&1; tail -1 $O/decode_one.txt | cut -c1-250

#!/bin/bash
# round-3 experiment 3: per-shape GEMM rates of the current build; deferred-wgrad knobs at step level
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp3
mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/gemm_bench.py > $O/gemm_bench.txt 2>&1; tail -17 $O/gemm_bench.txt
timeout -k 10 700 python3 tools/step_ab.py --rounds 3 --steps 8 "MM_DEFER_PERSIST=0" "MM_DEFER_PERSIST=1" "MM_DEFER_WGRAD_LAYERS=20" "MM_DEFER_WGRAD_LAYERS=24" > $O/step_ab.txt 2>&1
tail -5 $O/step_ab.txt

#!/bin/bash
# round-3 experiment 12: side bursts (AdamW, deferred wgrads) on CU-masked streams
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp12
mkdir -p $O
cd $R
timeout -k 10 1000 python3 tools/step_ab.py --rounds 3 --steps 5 --warmup 2 "MM_ADAMW_CUS=0" "MM_ADAMW_CUS=224" "MM_ADAMW_CUS=192" "MM_DEFER_CUS=240" "MM_DEFER_CUS=224" "MM_ADAMW_CUS=224,MM_DEFER_CUS=240" > $O/step_ab.txt 2>&1
tail -7 $O/step_ab.txt

#!/bin/bash
# round-3 experiment 1: wgrad GEMMs on a side stream beside the dgrad chain (tools/step_ab.py), plus baseline micro-benchmarks
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp1
mkdir -p $O
cd $R
python3 -c "import torch; print('prio range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" > $O/prio.txt 2>&1
timeout -k 10 200 python3 tools/attn_bench.py > $O/attn_bench.txt 2>&1
echo "attn done"; cat $O/attn_bench.txt | tail -5
timeout -k 10 700 python3 tools/step_ab.py --rounds 3 --steps 8 "MM_WGRAD_SIDE=0" "MM_WGRAD_SIDE=1" "MM_WGRAD_SIDE=1,MM_DEFER_WGRAD_LAYERS=0" "MM_WGRAD_SIDE=1,opt:gemm_persist=0" "MM_WGRAD_SIDE=1,MM_WGRAD_SIDE_PRIO=-1" > $O/step_ab.txt 2>&1
tail -8 $O/step_ab.txt

#!/bin/bash
# round-3 experiment 18: the side bursts on LOW-priority HIP streams
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp18
mkdir -p $O
cd $R
python3 -c "
import torch; torch.cuda.init()
from multimeditron_amd import kernels as K
print('priority range (least, greatest):', K.stream_priority_range())" > $O/range.txt 2>&1; cat $O/range.txt | tail -1
timeout -k 10 1000 python3 tools/step_ab.py --rounds 4 --steps 5 --warmup 2 "MM_X=0" "MM_DEFER_PRIO=1" "MM_ADAMW_PRIO=1" "MM_DEFER_PRIO=1,MM_ADAMW_PRIO=1" > $O/step_ab.txt 2>&1
tail -5 $O/step_ab.txt

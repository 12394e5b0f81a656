#!/bin/bash
# round-3 experiment 6: decode hand-offs with write-through stores (no release fence): fused skinny epilogues / norm tail, merge in the partial kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp6
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -m gpu -k "decode or skinny or generate or greedy" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.txt
for cfg in "0 launch" "1 launch" "1 fused" "0 fused" "1 launch" "0 launch"; do
  set -- $cfg
  MM_DECODE_FUSED=$1 MM_DECODE_MERGE=$2 timeout -k 10 300 python3 tools/decode_bench.py > $O/decode_$1_$2.txt 2>&1; echo "fused=$1 merge=$2: $(tail -1 $O/decode_$1_$2.txt)"
done
cd /tmp && export TMPDIR=/tmp
MM_DECODE_FUSED=1 MM_DECODE_MERGE=fused timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python3 $R/tools/decode_bench.py > $O/decode_prof.txt 2>&1
csv=$(find $O/prof -name "*kernel_stats.csv" | head -1); python3 $R/tools/prof_summary.py $csv 33 12 > $O/decode_kernels.txt; cat $O/decode_kernels.txt
rm -rf $O/prof

#!/bin/bash
# On the GPU box (through gpurun):  tools/gemm_pmc.sh TAG [modes]   -> gpurun_out/gemm_pmc_TAG/summary.txt
# One --pmc pass (8 SQ counters + GRBM_GUI_ACTIVE) over tools/gemm_pmc_run.py, summarised per kernel by tools/pmc_kernels.py.
set -e -o pipefail
tag=${1:-run}
modes=${2:-0,1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/gemm_pmc_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/p1 -o r -- python3 $R/tools/gemm_pmc_run.py $modes > $O/run1.txt 2>&1
python3 $R/tools/pmc_kernels.py $O/p1 gemm_bf16 > $O/summary.txt
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt

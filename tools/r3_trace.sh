#!/bin/bash
# per-stream composition of one training step (tools/stream_time.py over a rocprofv3 kernel trace of bench.py)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-trace}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o r -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-all-rows > $O/bench.json 2> $O/bench.err
csv=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/stream_time.py $csv > $O/stream_time.txt 2>&1
python3 $R/tools/gap_analysis.py $csv 30 > $O/gaps.txt 2>&1
rm -rf $O/kt
tail -70 $O/stream_time.txt | cut -c1-200

#!/bin/bash
# full GPU test suite + smoke + driver-style bench (what the driver runs at round end)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-suite}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -25 $O/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.txt
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json

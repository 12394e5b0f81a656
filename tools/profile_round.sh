#!/bin/bash
# Run on the GPU box (through gpurun):  tools/profile_round.sh TAG
# Writes gpurun_out/prof_TAG/: unprofiled bench line, rocprofv3 --kernel-trace --stats of the same command, three
# separate --pmc passes (MFMA busy + clock, FETCH_SIZE, WRITE_SIZE), and the two summaries (+ pmc_traffic.json, stamped with
# the kernel sources' hash: copy it to profiles/rNN_pmc_traffic.json and bench.py reports it as roofline.traffic).
set -e -o pipefail
tag=${1:-run}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py --steps 10 --warmup 3 > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
echo "unprofiled: $(cut -c1-160 $O/bench_unprofiled.json)"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-all-rows > $O/bench_profiled.json 2> $O/bench_profiled.err
csv=$(find $O/stats -name "*kernel_stats.csv" | head -1)
cp $csv $O/kernel_stats.csv
python3 $R/tools/prof_summary.py $O/kernel_stats.csv 5 48 > $O/kernel_summary.txt
echo "stats done"
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $O/pmc/$n -o r -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-all-rows > $O/pmc_$n.json 2> $O/pmc_$n.err
  echo "pmc $n done"
done
python3 $R/tools/pmc_summary.py $O/pmc $O/pmc_traffic.json > $O/pmc_summary.md
find $O -name "*.csv" -size +2M -delete
rm -rf $O/pmc/*/*/*kernel_trace* 2>/dev/null || true
echo done

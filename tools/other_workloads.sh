#!/bin/bash
# Run on the GPU box (through gpurun):  tools/other_workloads.sh TAG
# The bench lines of the workloads that are NOT the headline (BASELINE configs 4 and 5, the 1B model, ALIGNMENT mode), the
# attention and decode micro-benchmarks and the small-kernel table -> gpurun_out/other_TAG/.
set -o pipefail
tag=${1:-run}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/other_$tag
mkdir -p $O
cd $R
for wl in llama31_8b_vitl14_s4096_b2_4img qwen2_7b_siglip_so400m_s2048_b4 llama32_1b_vitb32_s2048_b4; do
  timeout -k 10 300 python3 bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err || echo "FAILED $wl"
  echo "$wl: $(cut -c1-220 $O/bench_$wl.json)"
done
timeout -k 10 300 python3 bench.py --mode ALIGNMENT --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_alignment.json 2> $O/bench_alignment.err || echo "FAILED alignment"
echo "alignment: $(cut -c1-220 $O/bench_alignment.json)"
timeout -k 10 200 python3 tools/attn_bench.py --quick > $O/attn_quick.log 2>&1; cat $O/attn_quick.log
timeout -k 10 200 python3 tools/attn_bench.py --ab-q > $O/attn_ab_q.log 2>&1; tail -16 $O/attn_ab_q.log
timeout -k 10 300 python3 tools/decode_bench.py > $O/decode.log 2>&1; tail -5 $O/decode.log
timeout -k 10 120 python3 tools/rowwise_bench.py > $O/rowwise.log 2>&1; cat $O/rowwise.log
timeout -k 10 300 python3 tools/gemv_bench.py 4 16 > $O/gemv.txt 2>&1; tail -12 $O/gemv.txt
timeout -k 10 300 python3 tools/moe_bench.py 4 4 > $O/moe_bench.txt 2>&1; tail -6 $O/moe_bench.txt
echo done

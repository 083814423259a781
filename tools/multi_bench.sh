#!/bin/bash
# bf16 4-step inference: layer-synchronous multi-branch block launches / time paths computed ahead
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $R/gpurun_out
cd $R
run() { echo "# $*"; env "$@" python3 bench.py $BA --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"; }
{
for BA in "--workload infer4 --gemm bf16" "--workload infer4 --gemm bf16 --no-graph" "--workload infer4"; do
  echo "## bench.py $BA"
  run F2G_FUSED_MULTI=1 F2G_TIME_AHEAD=1
  run F2G_FUSED_MULTI=0 F2G_TIME_AHEAD=1
  run F2G_FUSED_MULTI=1 F2G_TIME_AHEAD=0
  run F2G_FUSED_MULTI=0 F2G_TIME_AHEAD=0
done
} > $O/multi_bench.txt 2>&1

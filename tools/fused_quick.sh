#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
python3 tools/fused_multi_bench.py 2>&1 | grep -v amdgpu.ids
python3 tools/fused_mlp_bench.py 2>&1 | grep -v amdgpu.ids | head -8
for a in "" "" "" "--no-graph"; do python3 bench.py --workload infer4 --gemm bf16 $a --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('infer4 bf16 $a', d['ms_per_step'], d['value'])"; done

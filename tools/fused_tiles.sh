#!/bin/bash
# tile heights of the fused block kernel: single launches (F2G_MLP_RT) and the multi-branch launch
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
{
for rt in 0 1 2 3 4; do echo "## F2G_MLP_RT=$rt"; F2G_MLP_RT=$rt python3 tools/fused_multi_bench.py 2>&1 | grep -v amdgpu.ids | grep "alone\|serial"; done
for cfg in "3 4" "2 4" "3 2" "2 2"; do set -- $cfg; echo "## multi: 512-channel tiles $((32*$1)) rows, 384-channel tiles $((32*$2)) rows"; MODE=multi F2G_MULTI_RT512=$1 F2G_MULTI_RT384=$2 python3 tools/fused_multi_bench.py 2>&1 | grep multi; done
} > gpurun_out/fused_tiles.txt 2>&1
cat gpurun_out/fused_tiles.txt

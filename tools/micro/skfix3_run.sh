#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r4_skfix3.txt
for m in 0 2 3; do
  echo "## F2G_SKFIX=$m" >> gpurun_out/r4_skfix3.txt
  F2G_SKFIX=$m timeout 300 python3 tools/micro/lean_epi_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4_skfix3.txt
done
cat gpurun_out/r4_skfix3.txt

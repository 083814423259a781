#!/bin/bash
mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_hip_ops.py -q -x -k "streamk_with_seam" 2>&1 | tail -15 > gpurun_out/r4_skfix.txt
for m in 0 1 2; do
  echo "## F2G_SKFIX=$m" >> gpurun_out/r4_skfix.txt
  F2G_SKFIX=$m timeout 300 python3 tools/micro/lean_epi_bench.py >> gpurun_out/r4_skfix.txt 2>&1
done
cat gpurun_out/r4_skfix.txt

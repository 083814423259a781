#!/bin/bash
# on the GPU box: the epilogue / prologue ablations of the exact-fp32 lean GEMM
mkdir -p gpurun_out
: > gpurun_out/r4_lean_epi.txt
for v in 0 32 64 128 224; do
  echo "## F2G_LABVAR=$v" >> gpurun_out/r4_lean_epi.txt
  F2G_LIB_PATH=$PWD/tools/micro/liblev$v.so timeout 300 python3 tools/micro/lean_epi_bench.py >> gpurun_out/r4_lean_epi.txt 2>&1
done
cat gpurun_out/r4_lean_epi.txt

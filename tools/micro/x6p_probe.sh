#!/bin/bash
# round 5: gemm_x6p_kernel against gemm_x6t8_kernel in the bf16x6 step (per-shape HIP-event table of the
# serialised pass + the laned step), and lab builds of the old kernels without epilogue / without image read-back
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "halo_windows" 2>&1 | tail -4
i=0
for v in "F2G_X6P=0" "F2G_X6P=1" "F2G_X6P=0 F2G_LIB_PATH=$R/tools/micro/libx6lab1.so" "F2G_X6P=0 F2G_LIB_PATH=$R/tools/micro/libx6lab2.so"; do
  i=$((i+1))
  echo "## $v" > gpurun_out/x6p_$i.txt
  env $v F2G_GEMM_REPORT=45 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast-mode 2>> gpurun_out/x6p_$i.txt | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$v', j['ms_per_step'], r['achieved'], r['mfma_class']['ms_per_step_serialised'], {k:(v['ms'],v['tflops']) for k,v in r['mfma_class']['by_family'].items()})"
done

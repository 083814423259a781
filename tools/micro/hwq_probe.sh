#!/bin/bash
# does the number of hardware queues behind the launch lanes' streams matter (ROCm default: 4)?  And the steps with
# the lanes off, classic tile grids against the fix-up stream-K rule
mkdir -p gpurun_out
O=gpurun_out/r4_hwq_probe.txt
: > $O
run() { echo "# $*" >> $O; env "$@" python3 bench.py $BA --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $O; }
for BA in "--workload gan_stage2" "--workload stage1" "--workload gan_stage2 --gemm bf16x6"; do
  echo "## bench.py $BA" >> $O
  run F2G_SKFIX=0
  for q in 1 2 3 5 6 8 16; do run F2G_SKFIX=0 GPU_MAX_HW_QUEUES=$q; done
  run F2G_SKFIX=0 F2G_STREAMS=0
  run F2G_SKFIX=1 F2G_STREAMS=0
  run F2G_SKFIX=2 F2G_STREAMS=0
done
cat $O

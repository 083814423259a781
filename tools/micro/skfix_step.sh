#!/bin/bash
# A/B of the fix-up stream-K rule on the whole steps (same box)
mkdir -p gpurun_out
O=gpurun_out/r4_skfix_step.txt
: > $O
run() { echo "# $*" >> $O; env "$@" python3 bench.py $BA --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $O; }
for BA in "--workload gan_stage2" "--workload stage1" "--workload infer4"; do
  echo "## bench.py $BA" >> $O
  run F2G_SKFIX=0
  run F2G_SKFIX=1
  run F2G_SKFIX=1 F2G_SKFIX_MIN_SLABS=12
  run F2G_SKFIX=1 F2G_SKFIX_EFF=0.95
  run F2G_SKFIX=0
  run F2G_SKFIX=1
done
cat $O

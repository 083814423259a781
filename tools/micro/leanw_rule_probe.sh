#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r4_leanw_rule.txt
: > $O
run() { echo "# $*" >> $O; env "$@" python3 bench.py $BA --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $O; }
for BA in "--workload gan_stage2" "--workload stage1"; do
  echo "## bench.py $BA" >> $O
  run F2G_LEAN_WGRAD=1
  run F2G_LEAN_WGRAD=2
  run F2G_LEAN_WGRAD=0
  run F2G_LEAN_WGRAD=1
  run F2G_LEAN_WGRAD=2
done
cat $O

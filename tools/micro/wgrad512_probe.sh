#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r4_wgrad512.txt
: > $O
python3 -m pytest tests/test_hip_ops.py -q -x -k "wgrad" 2>&1 | tail -2 >> $O
run() { echo "# $*" >> $O; env "$@" python3 bench.py $BA --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $O; }
for BA in "--workload gan_stage2" "--workload gan_stage2 --n-timesteps 4"; do
  echo "## bench.py $BA" >> $O
  run F2G_WGRAD_SPLIT512=0
  run F2G_WGRAD_SPLIT512=1
  run F2G_WGRAD_SPLIT512=0
  run F2G_WGRAD_SPLIT512=1
done
cat $O

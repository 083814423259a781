#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r4_fuse_lrelu.txt
: > $O
run() { echo "# $*" >> $O; env "$@" python3 bench.py $BA --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $O; }
for BA in "--workload gan_stage2" "--workload gan_stage2 --gemm bf16x6"; do
  echo "## bench.py $BA" >> $O
  run F2G_FUSE_LRELU=0
  run F2G_FUSE_LRELU=1
  run F2G_FUSE_LRELU=2
  run F2G_FUSE_LRELU=3
  run F2G_FUSE_LRELU=0
done
cat $O

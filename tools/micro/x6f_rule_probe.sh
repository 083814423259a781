#!/bin/bash
# the in-kernel-split six-product kernel on more of the generator's GEMMs, judged in the laned step (same box)
mkdir -p gpurun_out
O=gpurun_out/r4_x6f_rule.txt
: > $O
BA="--workload gan_stage2 --gemm bf16x6"
run() { echo "# $*" >> $O; env "$@" python3 bench.py $BA --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $O; }
run F2G_X6F=2
run F2G_X6F=2 F2G_X6F_MIN_N=384
run F2G_X6F=2 F2G_X6F_MIN_N=384 F2G_X6F_MIN_K=512
run F2G_X6F=2 F2G_X6F_MIN_N=384 F2G_X6F_MIN_K=384
run F2G_X6F=2 F2G_X6F_MIN_K=512
run F2G_X6F=2 F2G_X6F_MIN_K=384
run F2G_X6F=2 F2G_X6_MIN_K=1024
run F2G_X6F=2 F2G_X6_MIN_K=1536
run F2G_X6F=2
cat $O

#!/bin/bash
# host cost of issuing a step: the same launches with almost no GPU work behind them (tiny batch)
mkdir -p gpurun_out
O=gpurun_out/r4_host_probe.txt
: > $O
run() { echo "# $*" >> $O; python3 bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d.get('host_issue_ms_per_step'))" >> $O; }
run --workload stage1 --batch 1
run --workload stage1 --batch 2
run --workload stage1 --batch 8
run --workload stage1
run --workload gan_stage2 --batch 1
run --workload gan_stage2 --batch 2
run --workload gan_stage2 --batch 8
run --workload gan_stage2
cat $O

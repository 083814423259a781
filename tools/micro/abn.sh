#!/bin/bash
# several environment settings on the laned step, interleaved over rounds: abn.sh <rounds> "<env 1>" "<env 2>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
N=$1; shift
for i in $(seq 1 $N); do
  for v in "$@"; do
    echo -n "# [$v] "
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])"
  done
done

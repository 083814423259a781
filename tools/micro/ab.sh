#!/bin/bash
# A/B of environment switches on the laned step, interleaved (boxes drift): ab.sh "<env A>" "<env B>" [rounds] [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
A="$1"; B="$2"; N=${3:-2}; shift 3
for i in $(seq 1 $N); do
  for v in "$A" "$B"; do
    echo -n "# [$v] "
    env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])"
  done
done

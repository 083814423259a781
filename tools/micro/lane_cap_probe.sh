#!/bin/bash
# how many streams behind each lane pool does the stage-2 step want? (F2G_LANE_CAP, same box)
mkdir -p gpurun_out
O=gpurun_out/r4_lane_cap.txt
: > $O
run() { echo "# $*" >> $O; env "$@" python3 bench.py $BA --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $O; }
for BA in "--workload gan_stage2" "--workload gan_stage2 --gemm bf16x6"; do
  echo "## bench.py $BA" >> $O
  run F2G_LANE_CAP=
  run F2G_LANE_CAP=mpd=1
  run F2G_LANE_CAP=mpd=2
  run F2G_LANE_CAP=mpd=3
  run F2G_LANE_CAP=mrd=1
  run F2G_LANE_CAP=mrd=2
  run F2G_LANE_CAP=mel=1
  run F2G_LANE_CAP=mel=2
  run F2G_LANE_CAP=branch=1
  run F2G_LANE_CAP=branch=2
  run F2G_LANE_CAP=disc=1
  run F2G_LANE_CAP=disc=2
  run F2G_LANE_CAP=condpath=1,timepath=1
  run F2G_LANE_CAP=
done
cat $O

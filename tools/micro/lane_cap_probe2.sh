#!/bin/bash
# lane-pool caps with MORE hardware queues: is the loss at 8 queues too much concurrency (then explicit caps win) or the queues themselves?
mkdir -p gpurun_out
O=gpurun_out/r4_lane_cap2.txt
: > $O
BA="--workload gan_stage2"
run() { echo "# $*" >> $O; env "$@" python3 bench.py $BA --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $O; }
run F2G_LANE_CAP=
run GPU_MAX_HW_QUEUES=8 F2G_LANE_CAP=mpd=3,mrd=1,mel=1
run GPU_MAX_HW_QUEUES=8 F2G_LANE_CAP=mpd=2,mrd=2,mel=1
run GPU_MAX_HW_QUEUES=8 F2G_LANE_CAP=mpd=2,mrd=1,mel=1
run GPU_MAX_HW_QUEUES=8 F2G_LANE_CAP=mpd=4,mrd=1,mel=1
run GPU_MAX_HW_QUEUES=8 F2G_LANE_CAP=mpd=3,mrd=2,mel=1,branch=2
run GPU_MAX_HW_QUEUES=8 F2G_LANE_CAP=mpd=1,mrd=1,mel=1
run GPU_MAX_HW_QUEUES=6 F2G_LANE_CAP=mpd=3,mrd=1,mel=1
run F2G_LANE_CAP=mpd=3,mrd=2
run F2G_LANE_CAP=mpd=4,mrd=2,mel=2
run F2G_LANE_CAP=mpd=3,mrd=2,mel=1
run F2G_LANE_CAP=
cat $O

#!/bin/bash
# kernel time of the weight-image rebuild launches (multi_kernel) in a steady-state step, launch lanes off:
# multi_time.sh [F2G_OPTS value]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/multi_time
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export F2G_STREAMS=0
[ -n "$1" ] && export F2G_OPTS="$1"
rocprofv3 --kernel-trace --output-format csv -d $O -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode > /dev/null 2>&1
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$O/**/p_kernel_trace.csv", recursive=True)[0])))
m = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))) for r in rows if "multi_kernel" in r["Kernel_Name"])
print("F2G_OPTS=$1: multi_kernel launches", len(m), "total %.2f ms over 3 steps" % (sum(x[0] for x in m) / 1e3), "largest (us, grid):", [(round(a), g) for a, g in m[-8:]])
import collections
by = collections.Counter(); cnt = collections.Counter()
for a, g in m: by[g] += a; cnt[g] += 1
print("by grid size (threads): total us, launches:", [(g, round(by[g]), cnt[g]) for g in sorted(by, key=lambda g: -by[g])[:12]])
PY
rm -rf $O

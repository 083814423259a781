R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
rm -rf $O/prof_serial
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o p -- $B > /dev/null 2>&1
ls $O/prof_serial
python3 - <<PY
import csv, collections, re
rows = list(csv.DictReader(open("$O/prof_serial/p_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms (3 steps)", tot/1e6)
for r in rows[:45]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:70]
    print(f"{n:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
PY

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for m in fp32 bf16x6; do
  rm -rf $O/prof_ko_$m
  MODE=$m KO=mpd F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ko_$m -o p -- python3 $R/tools/knockout.py > $O/ko_$m.log 2>&1
  tail -1 $O/ko_$m.log
  python3 - $O/prof_ko_$m/p_kernel_stats.csv <<'P'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over 9 steps = {tot/9e6:.2f} ms per step (lanes off)")
for r in rows[:32]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n)[:60]
    print(f"{n:60s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/9e6:8.3f} ms/step {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
P
  rm -rf $O/prof_ko_$m
done

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_generator.py tests/test_hip_gan.py -x -q -k "fp32 and not full_width" 2>&1 | tail -2
for v in 1; do MODE=fp32 python tools/knockout.py 2>&1 | tail -1; MODE=bf16x6 python tools/knockout.py 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_lc
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lc -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode > /dev/null 2>&1
python3 - $O/prof_lc/p_kernel_stats.csv <<'P'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["Calls"]) for r in rows)
print("launches per step:", tot / 3)
small = [r for r in rows if float(r["AverageNs"]) < 12000]
print("launches per step under 12 us:", sum(int(r["Calls"]) for r in small) / 3)
for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:14]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n)[:50]
    print(f"{n:50s} {int(r['Calls'])/3:8.1f} per step  {float(r['TotalDurationNs'])/3e6:7.3f} ms/step")
P
rm -rf $O/prof_lc

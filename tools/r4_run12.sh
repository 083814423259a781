cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py -x -q -k "conv2ch or convpost" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for v in 0 1; do
  rm -rf $O/prof_c2
  F2G_CONV2CH_V2=$v MODE=fp32 KO=mpd F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o p -- python3 $R/tools/knockout.py > $O/c2_$v.log 2>&1
  echo "## F2G_CONV2CH_V2=$v"; tail -1 $O/c2_$v.log
  python3 - $O/prof_c2/p_kernel_stats.csv <<'P'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "conv2ch" in r["Name"] or "convpost" in r["Name"]:
        n = re.sub(r"\(.*", "", r["Name"])[:40]
        print(f"{n:40s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/9e6:8.3f} ms/step {float(r['AverageNs'])/1e3:9.1f} us")
P
done
rm -rf $O/prof_c2
cd $R
for v in 0 1; do F2G_CONV2CH_V2=$v MODE=fp32 python tools/knockout.py 2>&1 | tail -1; done

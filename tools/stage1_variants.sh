# stage-1 / fp32 inference under the library's scheduling switches (is the laned run sensitive to
# stream-K / K-major weight gradients when only three lanes are live?)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
run() { echo "# $*"; env "$@" python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"; }
for W in stage1 infer4; do
  echo "## workload $W"
  run F2G_NOP=1
  run F2G_STREAMK=2
  run F2G_STREAMK=0
  run F2G_LEAN_WGRAD=2
  run F2G_STREAMS=0
  run F2G_STREAMK=2 F2G_STREAMS=0
done > $O/stage1_variants.txt 2>&1

R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_b3_lanes
rocprofv3 --kernel-trace --output-format csv -d $O/prof_b3_lanes -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode --gemm bf16x3 > /dev/null 2>&1
ls $O/prof_b3_lanes

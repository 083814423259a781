R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_infer
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o p -- python3 $R/bench.py --workload infer4 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode --no-graph --gemm ${MODE:-bf16} > /dev/null 2>&1
ls $O/prof_infer
head -40 $O/prof_infer/p_kernel_stats.csv | cut -c1-200 > $O/prof_infer_stats.txt

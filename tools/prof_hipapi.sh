R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_api
F2G_STREAMS=0 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/prof_api -o p -- python3 $R/bench.py --workload infer4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode --no-graph --gemm bf16 > /dev/null 2>&1
python3 - <<'PY' > $O/prof_api_seq.txt
import csv, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/prof_api/"
api=list(csv.DictReader(open(O+"p_hip_api_trace.csv")))
ker=list(csv.DictReader(open(O+"p_kernel_trace.csv")))
print(list(api[0].keys())); print(list(ker[0].keys()))
kc={r["Correlation_Id"]:r["Kernel_Name"][:60] for r in ker}
rows=[r for r in api if r["Function"] in ("hipMemcpyWithStream","hipLaunchKernel","hipMemcpyAsync","hipMemsetAsync")]
n=len(rows)
for r in rows[n-260:n-60]:
    print(r["Function"], kc.get(r["Correlation_Id"],""))
PY

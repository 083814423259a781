#!/bin/bash
# HBM traffic of the dominant kernel (exact-fp32 gemm_lean_kernel) from PMC passes over the bench command:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), --kernel-trace only.  Run on the GPU box AFTER the
# last change to the library: the json carries f2g_version() (a digest of the sources) and bench.py reports
# `roofline.traffic` only from a file recorded with the library it runs.  -> gpurun_out/pmcb_*/ + the json
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B1="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmcb_$set
  F2G_STREAMS=0 timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcb_$set -o p -- $B1 > /dev/null 2>&1
  find $O/pmcb_$set -type f ! -name "p_counter_collection.csv" -delete
done
cd $R && python3 tools/pmc_traffic_json.py $O/r04_pmc_gemm_traffic.json && cat $O/r04_pmc_gemm_traffic.json
# the same for the bf16x6 mode's dominant family (the six-product GEMM kernels) -> r04_pmc_x6_traffic.json
cd /tmp
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmcb_x6_$set
  F2G_STREAMS=0 timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcb_x6_$set -o p -- $B1 --gemm bf16x6 > /dev/null 2>&1
  find $O/pmcb_x6_$set -type f ! -name "p_counter_collection.csv" -delete
done
cd $R && MODE=bf16x6 python3 tools/pmc_traffic_json.py $O/r04_pmc_x6_traffic.json && cat $O/r04_pmc_x6_traffic.json

#!/bin/bash
# Run ON the GPU box AFTER the last change to the library: the whole GPU suite, the PMC traffic of the dominant
# kernel families (the json carries f2g_version(); bench.py reports roofline.traffic only for a matching library),
# then the default bench line.  -> gpurun_out/final_r06/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final_r06
rm -rf $O; mkdir -p $O
cd $R
( time python -m pytest tests -x -q -m gpu ) > $O/gputest.txt 2>&1
tail -5 $O/gputest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
B1="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
for set in FETCH_SIZE WRITE_SIZE; do
  F2G_STREAMS=0 timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcb_x6_$set -o p -- $B1 > /dev/null 2>&1
  F2G_STREAMS=0 timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcb_$set -o p -- $B1 --gemm fp32 > /dev/null 2>&1
done
for d in pmcb_FETCH_SIZE pmcb_WRITE_SIZE pmcb_x6_FETCH_SIZE pmcb_x6_WRITE_SIZE; do find $O/$d -type f ! -name "p_counter_collection.csv" -delete; done
cd $R
G=gpurun_out/final_r06 MODE=bf16x6 python3 tools/pmc_traffic_json.py $O/pmc_x6_traffic.json
G=gpurun_out/final_r06 MODE=fp32 python3 tools/pmc_traffic_json.py $O/pmc_gemm_traffic.json
cp $O/pmc_x6_traffic.json profiles/r06_pmc_x6_traffic.json; cp $O/pmc_gemm_traffic.json profiles/r06_pmc_gemm_traffic.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode > /dev/null 2>&1
find $O/prof_serial -type f ! -name "p_kernel_stats.csv" -delete
F2G_GEMM_REPORT=90 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode 2> $O/shapes_x6.txt > /dev/null
bash tools/pmc_busy.sh r06_x6 > /dev/null 2>&1; cp gpurun_out/pmc_busy_r06_x6.txt $O/pmc_busy_x6.txt
tail -c 300 $O/bench_default.json

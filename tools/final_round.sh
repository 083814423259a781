#!/bin/bash
# Run ON the GPU box AFTER the last change to the library: the whole GPU suite (three GEMM modes), the PMC
# traffic of the dominant kernel (carries f2g_version()), the default bench line.  -> gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( time python -m pytest tests -x -q -m gpu --durations=15 ) > gpurun_out/gputest_final.txt 2>&1
tail -4 gpurun_out/gputest_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
cp gpurun_out/r04_pmc_gemm_traffic.json gpurun_out/r04_pmc_x6_traffic.json profiles/   # (bench.py reads profiles/)
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --gemm bf16x6 --no-fast-mode > gpurun_out/bench_x6_candidate.json 2>/dev/null
tail -c 400 gpurun_out/bench_default.json

#!/bin/bash
# Run ON the GPU box (through gpurun): every measurement profiles/<round>_* is made from.
# Outputs go to gpurun_out/; tools/make_profiles.py turns them into the committed summaries.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
F2G_GEMM_REPORT=80 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fast-mode 2> $O/shapes.txt > /dev/null
[ -x tools/micro/gemm_lab ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/gemm_lab.hip -o tools/micro/gemm_lab > /dev/null 2>&1
./tools/micro/gemm_lab > $O/gemm_lab.txt 2>&1
for w in stage1 infer4; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$w.json
done
python bench.py --model mel_44k_128band_512x_base --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_44k.json
python bench.py --n-timesteps 4 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_n4.json
python bench.py --optimizer --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | tail -1 > $O/bench_opt.json
python tools/hbm_kernel_bench.py > $O/hbm_kernels.txt 2>/dev/null
python tools/lean3_bench.py > $O/lean3_bench.txt 2>/dev/null
python tools/conv32_b3_bench.py > $O/conv32_b3_bench.txt 2>/dev/null
python tools/streaming_latency.py 100 > $O/streaming.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lanes -o p -- $B > /dev/null 2>&1
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b3 -o p -- $B --gemm bf16x3 > /dev/null 2>&1
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o p -- python3 $R/bench.py --workload infer4 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode --gemm bf16 > /dev/null 2>&1
B1="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
for set in FETCH_SIZE WRITE_SIZE; do
  F2G_STREAMS=0 timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcb_$set -o p -- $B1 > /dev/null 2>&1
done
tail -1 $O/bench_default.json | cut -c1-200

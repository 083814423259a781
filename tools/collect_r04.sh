#!/bin/bash
# Run ON the GPU box (through gpurun): the measurements profiles/r04_* are made from (the round-4 subset of
# tools/collect_profiles.sh + this round's probes).  Outputs go to gpurun_out/; ROUND=r04 python
# tools/make_profiles.py turns them into the committed summaries.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
F2G_GEMM_REPORT=80 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fast-mode 2> $O/shapes.txt > /dev/null
for w in stage1 infer4; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$w.json
done
python bench.py --model mel_44k_128band_512x_base --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_44k.json
python bench.py --n-timesteps 4 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_n4.json
python bench.py --optimizer --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | tail -1 > $O/bench_opt.json
python bench.py --workload infer4 --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --gemm bf16 2>/dev/null | tail -1 > $O/bench_infer4_bf16.json
python tools/hbm_kernel_bench.py > $O/hbm_kernels.txt 2>/dev/null
python tools/streaming_latency.py 100 > $O/streaming.txt 2>/dev/null
# bf16x6 mode: per-shape table, variants
F2G_GEMM_REPORT=60 python bench.py --gemm bf16x6 --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode 2> $O/x6_shapes.txt > $O/x6_bench_roofline.json
F2G_X6_MIN_K=32 F2G_GEMM_REPORT=60 python bench.py --gemm bf16x6 --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode 2> $O/x6_shapes_all.txt > /dev/null
( for env in "" "F2G_CONV32_X6=0" "F2G_X6_TAP8=0" "F2G_X6_WGRAD=0"; do
    echo "# $env python bench.py --gemm bf16x6 --steps 8 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline"
    env $env python bench.py --gemm bf16x6 --steps 8 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline 2>/dev/null | tail -1 | cut -c1-260
  done
  for env in "" "F2G_CONV32_WGRAD_V2=0" "F2G_CONV2CH_V2=0" "F2G_BAND_LANES=1"; do
    echo "# $env python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline   (exact fp32, same box)"
    env $env python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline 2>/dev/null | tail -1 | cut -c1-260
  done ) > $O/x6_step_variants.txt
python tools/x6_gemm_bench.py > $O/x6_gemm_bench.txt 2>/dev/null
# direct MRD convs: exact fp32 and fp32 class
( for m in fp32 bf16x6; do echo "## MODE=$m"; MODE=$m python tools/conv32_probe.py 2>&1 | grep -v amdgpu.ids; done ) > $O/conv32_probe.txt
# marginal cost of the step's components in the laned schedule
( for m in fp32 bf16x6; do for k in "" mrd mpd mel "mpd,mrd" "mpd,mrd,mel"; do MODE=$m KO=$k python tools/knockout.py 2>&1 | tail -1; done; done ) > $O/knockout.txt
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lanes -o p -- $B > /dev/null 2>&1
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b3 -o p -- $B --gemm bf16x3 > /dev/null 2>&1
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x6 -o p -- $B --gemm bf16x6 > /dev/null 2>&1
B1="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
for set in FETCH_SIZE WRITE_SIZE; do
  F2G_STREAMS=0 timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcb_$set -o p -- $B1 > /dev/null 2>&1
done
# (keep only what make_profiles.py reads: the csv summaries)
for d in prof_serial prof_lanes prof_b3 prof_x6; do find $O/$d -type f ! -name "p_kernel_stats.csv" -delete; done
for d in pmcb_FETCH_SIZE pmcb_WRITE_SIZE; do find $O/$d -type f ! -name "p_counter_collection.csv" -delete; done
bash $R/tools/pmc_lean_fp32.sh > $O/pmc_lean_fp32.txt 2>&1
tail -1 $O/bench_default.json | cut -c1-200

#!/bin/bash
# Run ON the GPU box (through gpurun): the measurements profiles/r06_* are made from.  Outputs go to gpurun_out/;
# python tools/make_profiles_r06.py turns them into the committed summaries.  bench.py's default GEMM arithmetic
# is bf16x6 (fp32 class) since round 5; every exact-fp32 run says --gemm fp32.  Round 6: the default line invalidates the
# derived weight images after every sub-step (--frozen-weights: the rounds 1-5 condition).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
rm -rf $O; mkdir -p $O
cd $R
python tools/hbm_kernel_bench.py $O/hbm_class_isolated.json > $O/hbm_kernels.txt 2>/dev/null
cp $O/hbm_class_isolated.json profiles/hbm_class_isolated.json      # (so that the default line below carries it)
# HBM traffic of the dominant kernel families (PMC passes) FIRST: the default line below reports it as
# roofline.traffic only when profiles/ holds a json of THIS library version
cd /tmp && export TMPDIR=/tmp
B1="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
for set in FETCH_SIZE WRITE_SIZE; do
  F2G_STREAMS=0 timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcb_x6_$set -o p -- $B1 > /dev/null 2>&1
  F2G_STREAMS=0 timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcb_$set -o p -- $B1 --gemm fp32 > /dev/null 2>&1
done
for d in pmcb_FETCH_SIZE pmcb_WRITE_SIZE pmcb_x6_FETCH_SIZE pmcb_x6_WRITE_SIZE; do find $O/$d -type f ! -name "p_counter_collection.csv" -delete; done
cd $R
G=gpurun_out/r06 MODE=bf16x6 python3 tools/pmc_traffic_json.py $O/pmc_x6_traffic.json
G=gpurun_out/r06 MODE=fp32 python3 tools/pmc_traffic_json.py $O/pmc_gemm_traffic.json
cp $O/pmc_x6_traffic.json profiles/r06_pmc_x6_traffic.json; cp $O/pmc_gemm_traffic.json profiles/r06_pmc_gemm_traffic.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
F2G_GEMM_REPORT=90 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode 2> $O/shapes_x6.txt > /dev/null
F2G_GEMM_REPORT=90 python bench.py --gemm fp32 --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode 2> $O/shapes_fp32.txt > /dev/null
for w in stage1 infer4; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$w.json
done
python bench.py --model mel_44k_128band_512x_base --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_44k.json
python bench.py --n-timesteps 4 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_n4.json
python bench.py --optimizer --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | tail -1 > $O/bench_opt.json
python bench.py --frozen-weights --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | tail -1 > $O/bench_frozen.json
python bench.py --workload stage1 --frozen-weights --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-fast-mode 2>/dev/null | tail -1 > $O/bench_stage1_frozen.json
F2G_GEMM_REPORT=40 F2G_GEMM_REPORT_EPI=1 python bench.py --workload stage1 --steps 4 --warmup 2 --no-cpu-baseline --no-fast-mode 2> $O/shapes_stage1_epi.txt > /dev/null
bash tools/dbg/copybuffer_context.sh > /dev/null 2>&1; cp gpurun_out/copybuffer_context.txt $O/steady_state_kernels.txt
./tools/micro/x6_stream_peak > $O/x6_stream_peak.txt 2>&1
python bench.py --workload infer4 --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --gemm bf16 2>/dev/null | tail -1 > $O/bench_infer4_bf16.json
( for m in bf16x6 fp32; do for k in "" mrd mpd mel "mpd,mrd" "mpd,mrd,mel"; do MODE=$m KO=$k python tools/knockout.py 2>&1 | tail -1; done; done ) > $O/knockout.txt
( for m in fp32 bf16x6; do echo "## MODE=$m"; MODE=$m python tools/conv32_probe.py 2>&1 | grep -v amdgpu.ids; done ) > $O/conv32_probe.txt
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lanes -o p -- $B > /dev/null 2>&1
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32 -o p -- $B --gemm fp32 > /dev/null 2>&1
for d in prof_serial prof_lanes prof_fp32; do find $O/$d -type f ! -name "p_kernel_stats.csv" -delete; done
bash $R/tools/pmc_busy.sh r06_x6 > /dev/null 2>&1; cp $R/gpurun_out/pmc_busy_r06_x6.txt $O/pmc_busy_x6.txt
bash $R/tools/pmc_busy.sh r06_fp32 --gemm fp32 > /dev/null 2>&1; cp $R/gpurun_out/pmc_busy_r06_fp32.txt $O/pmc_busy_fp32.txt
tail -1 $O/bench_default.json | cut -c1-200

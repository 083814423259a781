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
# round 3: fused block kernel, persistent MRD convs, weight-gradient splits, bf16 inference
python tools/fused_mlp_bench.py > $O/fused_mlp_bench.txt 2>/dev/null
SPECS="0_16 1_16 2_16 4_16 8_16 15_16 0_8 0_24" bash tools/micro/fusedmlp_lab.sh > $O/fused_mlp_lab.txt 2>/dev/null
( python tools/fused_multi_bench.py 2>&1 | grep -v amdgpu.ids
  for rt in 1 2 3 4; do echo "## F2G_MLP_RT=$rt (rows per tile = 32 x $rt where the shape has the instance)"; F2G_MLP_RT=$rt python3 tools/fused_multi_bench.py 2>&1 | grep "alone"; done ) > $O/fused_multi.txt
bash tools/pmc_multi.sh > $O/pmc_multi.txt 2>&1
# round 3: fp32-class products on the bf16 pipe (bf16x6): per-shape tables in the step, every eligible GEMM
# (thresholds off) and the default rule; the laned step with and without the weight-gradient kernel /
# the producer-written images; a kernel trace of the mode
F2G_X6_MIN_K=32 F2G_GEMM_REPORT=60 python bench.py --gemm bf16x6 --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode 2> $O/x6_shapes_all.txt > /dev/null
F2G_GEMM_REPORT=60 python bench.py --gemm bf16x6 --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode 2> $O/x6_shapes.txt > $O/x6_bench_roofline.json
( for env in "" "F2G_X6_WGRAD=0" "F2G_X6_WGRAD=0 F2G_X3_PRODUCERS=0" "F2G_X6_MIN_K=32"; do
    echo "# $env python bench.py --gemm bf16x6 --steps 8 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline"
    env $env python bench.py --gemm bf16x6 --steps 8 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline 2>/dev/null | tail -1 | cut -c1-260
  done
  echo "# python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline   (exact fp32, same box)"
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline 2>/dev/null | tail -1 | cut -c1-260 ) > $O/x6_step_variants.txt
python tools/x6_gemm_bench.py > $O/x6_gemm_bench.txt 2>/dev/null
bash tools/pmc_x6_step.sh > $O/pmc_x6_step.txt 2>&1
python tools/conv32_probe.py > $O/conv32_probe.txt 2>/dev/null
F2G_CONV32_V2=0 python tools/conv32_probe.py 2>/dev/null | grep "all 45" > $O/conv32_probe_round2_kernels.txt
SWEEP=2,4,8,16 python tools/wgrad_probe.py > $O/wgrad_probe.txt 2>/dev/null
BI="python bench.py --workload infer4 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fast-mode --gemm bf16"
( echo "# $BI   [default: fused block kernel, HIP-graph replay]"; $BI 2>/dev/null | tail -1
  echo "# ... --no-graph"; $BI --no-graph 2>/dev/null | tail -1
  echo "# F2G_FUSED_MULTI=0 F2G_TIME_AHEAD=0 (one block launch per branch and lane, time paths per step)"; F2G_FUSED_MULTI=0 F2G_TIME_AHEAD=0 $BI 2>/dev/null | tail -1
  echo "# F2G_FUSED_BLOCK=0 (dwnorm + fused MLP as two launches)"; F2G_FUSED_BLOCK=0 $BI 2>/dev/null | tail -1
  echo "# F2G_FUSED_BLOCK=0 F2G_FUSED_MLP=0 (round 2: dwnorm + two lean GEMMs)"; F2G_FUSED_BLOCK=0 F2G_FUSED_MLP=0 $BI 2>/dev/null | tail -1
  echo "# F2G_FUSED_BLOCK=0 F2G_FUSED_MLP=0 --no-graph (round 2 as it was launched)"; F2G_FUSED_BLOCK=0 F2G_FUSED_MLP=0 $BI --no-graph 2>/dev/null | tail -1 ) > $O/infer4_bf16_variants.txt
python bench.py --workload infer4 --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --gemm bf16 2>/dev/null | tail -1 > $O/bench_infer4_bf16.json
BARGS="--workload infer4 --gemm bf16" bash tools/prof_timeline.sh > /dev/null 2>&1; cp $O/prof_tl.txt $O/infer4_bf16_timeline.txt
python3 tools/timeline_last.py $O/prof_tl/p_kernel_trace.csv bct_to_rows 400 > $O/infer4_bf16_timeline_kernels.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lanes -o p -- $B > /dev/null 2>&1
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b3 -o p -- $B --gemm bf16x3 > /dev/null 2>&1
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o p -- python3 $R/bench.py --workload infer4 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode --gemm bf16 > /dev/null 2>&1
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x6 -o p -- $B --gemm bf16x6 > /dev/null 2>&1
B1="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode"
for set in FETCH_SIZE WRITE_SIZE; do
  F2G_STREAMS=0 timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcb_$set -o p -- $B1 > /dev/null 2>&1
done
tail -1 $O/bench_default.json | cut -c1-200

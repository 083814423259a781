#!/bin/bash
# the bf16-inference part of tools/collect_profiles.sh alone (after a change to the fused block kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
python tools/fused_mlp_bench.py > $O/fused_mlp_bench.txt 2>/dev/null
SPECS="0_16 1_16 2_16 4_16 8_16 15_16 0_8 0_24" bash tools/micro/fusedmlp_lab.sh > $O/fused_mlp_lab.txt 2>/dev/null
( python tools/fused_multi_bench.py 2>&1 | grep -v amdgpu.ids
  for rt in 1 2 3 4; do echo "## F2G_MLP_RT=$rt (rows per tile = 32 x $rt where the shape has the instance)"; F2G_MLP_RT=$rt python3 tools/fused_multi_bench.py 2>&1 | grep "alone"; done ) > $O/fused_multi.txt
bash tools/pmc_multi.sh > $O/pmc_multi.txt 2>&1
cd $R
BI="python bench.py --workload infer4 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-fast-mode --gemm bf16"
( echo "# $BI   [default: one fused launch per layer for all branches, time paths ahead, HIP-graph replay]"; $BI 2>/dev/null | tail -1
  echo "# ... --no-graph"; $BI --no-graph 2>/dev/null | tail -1
  echo "# F2G_FUSED_MULTI=0 F2G_TIME_AHEAD=0 (one block launch per branch and lane, time paths per step)"; F2G_FUSED_MULTI=0 F2G_TIME_AHEAD=0 $BI 2>/dev/null | tail -1
  echo "# F2G_FUSED_BLOCK=0 (dwnorm + fused MLP as two launches)"; F2G_FUSED_BLOCK=0 $BI 2>/dev/null | tail -1
  echo "# F2G_FUSED_BLOCK=0 F2G_FUSED_MLP=0 (round 2: dwnorm + two lean GEMMs)"; F2G_FUSED_BLOCK=0 F2G_FUSED_MLP=0 $BI 2>/dev/null | tail -1
  echo "# F2G_FUSED_BLOCK=0 F2G_FUSED_MLP=0 --no-graph (round 2 as it was launched)"; F2G_FUSED_BLOCK=0 F2G_FUSED_MLP=0 $BI --no-graph 2>/dev/null | tail -1 ) > $O/infer4_bf16_variants.txt
python bench.py --workload infer4 --steps 20 --warmup 5 --no-cpu-baseline --no-fast-mode --gemm bf16 2>/dev/null | tail -1 > $O/bench_infer4_bf16.json
BARGS="--workload infer4 --gemm bf16" bash tools/prof_timeline.sh > /dev/null 2>&1; cp $O/prof_tl.txt $O/infer4_bf16_timeline.txt
python3 tools/timeline_last.py $O/prof_tl/p_kernel_trace.csv bct_to_rows 400 > $O/infer4_bf16_timeline_kernels.txt 2>/dev/null
python tools/streaming_latency.py 100 > $O/streaming.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
F2G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o p -- python3 $R/bench.py --workload infer4 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode --gemm bf16 > /dev/null 2>&1
grep -o '"ms_per_step": [0-9.]*' $O/infer4_bf16_variants.txt; cat $O/streaming.txt

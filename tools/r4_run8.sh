cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -x -q -k "fp32_class" > gpurun_out/r4_t8.txt 2>&1; tail -2 gpurun_out/r4_t8.txt
B="python bench.py --gemm bf16x6 --steps 6 --warmup 2 --no-cpu-baseline --no-fast-mode"
for e in "F2G_X6_TAP8=0" "F2G_X6_TAP8=1"; do
  echo "## $e"
  env $e F2G_GEMM_REPORT=14 $B 2> gpurun_out/r4_x6_shapes_$e.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['mfma_class']['by_family']['x6'])"
  grep -A14 "^form" gpurun_out/r4_x6_shapes_$e.txt | head -15
done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -x -q -k "conv32" > gpurun_out/r4_conv32_tests.txt 2>&1; tail -3 gpurun_out/r4_conv32_tests.txt
MODE=bf16x6 python tools/conv32_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_conv32_probe_x6.txt
grep "all 45\|Win=256\|H= 94 Win= 39\|H=188 Win= 20" gpurun_out/r4_conv32_probe_x6.txt
{
for v in 1 2 4 16 7; do
  echo "## F2G_LABVAR=$v"
  F2G_LIB_PATH=$GRAFT_REPO_ROOT/tools/micro/libc6v$v.so MODE=bf16x6 ONLY=fwd python tools/conv32_probe.py 2>&1 | grep "all 45\|H= 47 Win=256\|H= 94 Win= 39\|H=188 Win= 20"
done
} > gpurun_out/r4_conv32x6_ablate2.txt 2>&1
cat gpurun_out/r4_conv32x6_ablate2.txt
for m in bf16x6; do python bench.py --gemm $m --steps 6 --warmup 2 --no-cpu-baseline --no-fast-mode 2>/dev/null | tail -1 > gpurun_out/r4_bench_x6_a.json; done
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4_bench_x6_a.json').read().strip().split('\n')[-1])
print('x6 step', d['ms_per_step'], d['value']); print(json.dumps(d['roofline']['mfma_class']['by_family']))
P
F2G_CONV32_X6=0 python bench.py --gemm bf16x6 --steps 6 --warmup 2 --no-cpu-baseline --no-fast-mode --no-roofline 2>/dev/null | tail -1 | cut -c1-200
